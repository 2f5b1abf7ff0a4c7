"""Ad-hoc timing of one Baum-Welch E-step at BASELINE config 4 scale (dev tool):
    python scripts/time_estep.py [N] [D] [emitting states]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
from tests._synth import trained_like_models
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 13
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
T, W = 101, 10
sp, A, mu, cv = trained_like_models(W, NS, D, seed=3)
feats = torch.randn(N * T, D, device="cuda") * 20
feats[:, 0] -= 300
batch = FeatureBatch.from_packed(feats.contiguous(), np.full(N, T))
es = EStep(batch, np.arange(N) % W, W, NS + 2)
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
print(f"E-step N={N} D={D} S={NS + 2} W={W}: {dt*1e3:.2f} ms/iteration  {N*T/dt:.3e} frames/s  workspace {es.ws_bytes/1e9:.2f} GB")
