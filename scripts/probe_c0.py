"""Dev probe: distribution of c0 = log rho - s (custom-HMM E-step, scripts/experiments/README.md) over the first
Baum-Welch iterations on the benchmark's data: python scripts/probe_c0.py"""
import contextlib, io, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd.custom_hmm import HMM, pack_features
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.trellis import FeatureBatch
N = 20000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, N, seed=1234, device=dev)
feats, _ = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)(pcm, np.full(N, bench.N_SAMP, dtype=np.int64))
pk = pack_features(FeatureBatch.from_packed(feats, np.full(N, bench.T_FRAMES)))
host = feats.view(N, bench.T_FRAMES, bench.D)[:400].cpu().numpy()
sample = [np.ascontiguousarray(x.T) for x in host]
with contextlib.redirect_stdout(io.StringIO()):
    h = HMM(bench.N_STATES, bench.D, feature_set=pk, model_name="b")
for it in range(4):
    c0s = []
    with np.errstate(all="ignore"):
        for x in sample:
            e = h.compute_emission_matrix(x)
            a, s = h.forward(e)
            ll = np.logaddexp.reduce(a[-1])
            c0s.append(a[-1, -1] - ll - s)
    c0 = np.array(c0s)
    fin = np.isfinite(c0)
    print(f"before iteration {it + 1}: NaN {np.isnan(c0).mean():.3f}  -inf {np.isneginf(c0).mean():.3f}  "
          f">= -678: {(c0 >= -678).mean():.3f}  band: {((c0 < -678) & (c0 >= -750)).mean():.3f}  < -750: {(c0 < -750).mean():.3f}  "
          f"median {np.median(c0[fin]) if fin.any() else float('nan'):.1f} min {c0[fin].min() if fin.any() else 0:.1f}")
    with contextlib.redirect_stdout(io.StringIO()):
        h.baum_welch(pk, max_iter=1)
