"""Dev tool: host-side profile (cProfile) of custom_hmm.HMM.baum_welch at bench scale."""
import contextlib, cProfile, io, pstats, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd.custom_hmm import HMM, pack_features
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.trellis import FeatureBatch
N = 100000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, N, seed=1234, device=dev)
feats, _ = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)(pcm, np.full(N, bench.N_SAMP, dtype=np.int64))
pk = pack_features(FeatureBatch.from_packed(feats, np.full(N, bench.T_FRAMES)))
with contextlib.redirect_stdout(io.StringIO()):
    h = HMM(8, 13, feature_set=pk, model_name="b")
    h.baum_welch(pk, max_iter=1)
    h = HMM(8, 13, feature_set=pk, model_name="b")
    pr = cProfile.Profile()
    pr.enable()
    h.baum_welch(pk, max_iter=3)
    torch.cuda.synchronize()
    pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22)
print(st.getvalue()[:4500])
