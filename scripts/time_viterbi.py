"""Ad-hoc timing of the Viterbi kernels (dev tool): python scripts/time_viterbi.py N D n_states W"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.trellis import DiagModelPack, FeatureBatch, viterbi_decode
from tests._synth import trained_like_models
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 13
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 8
W = int(sys.argv[4]) if len(sys.argv) > 4 else 11
T = 101
sp, A, mu, cv = trained_like_models(W, ns, D, seed=3)
feats = torch.randn(N * T, D, device="cuda") * 20
feats[:, 0] -= 300
batch = FeatureBatch.from_packed(feats.contiguous(), np.full(N, T))
pack = DiagModelPack.from_params(sp, A, mu, cv)
for _ in range(2):
    viterbi_decode(batch, pack)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 3
for _ in range(K):
    viterbi_decode(batch, pack)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
ops = N * T * W * (ns + 2) * (D * 7 + 6)
print(f"N={N} D={D} S={ns+2} W={W}: {dt*1e3:.3f} ms  {N*T/dt:.3e} frames/s  fp64-instr-lanes/s={ops/dt:.3e} ({ops/dt/39.3e12*100:.0f}% of peak)")
