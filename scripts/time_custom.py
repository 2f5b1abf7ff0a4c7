"""Ad-hoc timing of the custom-HMM Baum-Welch at scale (dev tool): python scripts/time_custom.py [N] [iters]"""
import contextlib, io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.custom_hmm import HMM
from tests._synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = synth_batch(N, T=101, D=13, seed=1)            # (N, T, D)
feats = [np.ascontiguousarray(u.T) for u in x]      # reference layout (D, T)
t0 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    h = HMM(8, 13, feature_set=feats, model_name="w")
torch.cuda.synchronize()
t1 = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    hist = h.baum_welch(feats, max_iter=iters)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"custom HMM N={N}: flat start {t1-t0:.2f} s, baum_welch {iters} iterations {(t2-t1):.2f} s "
      f"({N*101*len(hist)/(t2-t1):.3e} frames/s incl. host work)  LL {hist}")
