"""Dev tool: where one custom-HMM Baum-Welch iteration spends its wall time at bench scale (phases separated by
device synchronisation): python scripts/phase_custom.py [N]"""
import contextlib, io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd import _lib, custom_hmm as ch
from sapr_amd.custom_hmm import HMM, pack_features
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.trellis import FeatureBatch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, N, seed=1234, device=dev)
lens = np.full(N, bench.N_SAMP, dtype=np.int64)
feats, _ = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)(pcm, lens)
pk = pack_features(FeatureBatch.from_packed(feats, np.full(N, bench.T_FRAMES)))
with contextlib.redirect_stdout(io.StringIO()):
    h = HMM(bench.N_STATES, bench.D, feature_set=pk, model_name="bench")
    h.baum_welch(pk, max_iter=1)
    h = HMM(bench.N_STATES, bench.D, feature_set=pk, model_name="bench")
T = {}
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r
# wrap the pieces baum_welch calls
orig_ma, orig_ub, orig_ua = ch.model_arrays, HMM._update_b_device, HMM.update_A
ch.model_arrays = lambda m: timed("model_arrays (inv, slogdet, log A)", lambda: orig_ma(m))
HMM._update_b_device = lambda self, *a, **k: timed("update_B (device + D2H + floor)", lambda: orig_ub(self, *a, **k))
HMM.update_A = lambda self, *a, **k: timed("update_A (host)", lambda: orig_ua(self, *a, **k))
lib = _lib.load()
class Wrap:
    def __init__(self, lib): self._l = lib
    def __getattr__(self, n):
        f = getattr(self._l, n)
        if n in ("sapr_custom_estep", "sapr_custom_estep_staged", "sapr_custom_fold_rows"):
            return lambda *a: timed(n, lambda: f(*a))
        return f
_lib_load = _lib.load
_lib.load = lambda: Wrap(_lib_load())
with contextlib.redirect_stdout(io.StringIO()):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist = h.baum_welch(pk, max_iter=3)
    torch.cuda.synchronize(); tot = (time.perf_counter() - t0) * 1e3
print(f"3 iterations {tot:.2f} ms = {tot / 3:.2f} per iteration (phases synchronised, so slower than the bench figure)")
for k, v in T.items():
    print(f"  {k:40s} {v / 3:.3f} ms per iteration")
print(f"  {'unaccounted (allocation, upload, D2H, python)':40s} {(tot - sum(T.values())) / 3:.3f}")
