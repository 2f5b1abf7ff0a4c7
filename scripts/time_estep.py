"""Ad-hoc timing of one Baum-Welch E-step at BASELINE config 4 scale (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
from tests._synth import trained_like_models
N, T, D, W = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, 101, 13, 10
sp, A, mu, cv = trained_like_models(W, 8, D, seed=3)
feats = torch.randn(N * T, D, device="cuda") * 20
feats[:, 0] -= 300
batch = FeatureBatch.from_packed(feats.contiguous(), np.full(N, T))
es = EStep(batch, np.arange(N) % W, W, 10)
pack = DiagModelPack.from_params(sp, A, mu, cv)
for _ in range(2):
    es.run(pack)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    es.run(pack)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"E-step N={N} W={W}: {dt*1e3:.2f} ms/iteration  {N*T/dt:.3e} frames/s  workspace {es.ws_bytes/1e9:.2f} GB")
