"""Ad-hoc timing of the Viterbi kernels at BASELINE config 3 (dev tool, not the bench)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd import _lib
from sapr_amd.trellis import DiagModelPack, FeatureBatch, viterbi_decode
from tests._synth import trained_like_models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T, D, W = 101, 13, 11
sp, A, mu, cv = trained_like_models(W, 8, D, seed=3)
g = torch.Generator(device="cuda").manual_seed(0)
feats = torch.randn(N * T, D, device="cuda", generator=g) * 20
feats[:, 0] -= 300
batch = FeatureBatch.from_packed(feats.contiguous(), np.full(N, T))
pack = DiagModelPack.from_params(sp, A, mu, cv)
for _ in range(2):
    viterbi_decode(batch, pack)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    viterbi_decode(batch, pack)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"N={N} frames={N*T} time/step={dt*1e3:.3f} ms  frames/s={N*T/dt:.3e}  frame-models/s={N*T*W/dt:.3e}")
