"""Ad-hoc timing of the custom-HMM batched decode (dev tool): python scripts/time_custom_decode.py [N] [W] [D]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.custom_hmm import HMM, decode_batch, pack_features
from sapr_amd.trellis import FeatureBatch
from tests._synth import synth_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 11
D = int(sys.argv[3]) if len(sys.argv) > 3 else 13
x = synth_batch(min(N, 4096), T=101, D=D, seed=1)
reps = -(-N // x.shape[0])
x = np.tile(x, (reps, 1, 1))[:N]
fb = FeatureBatch.from_packed(torch.from_numpy(x.reshape(-1, D)).cuda(), np.full(N, 101))
rng = np.random.default_rng(0)
models = []
for w in range(W):
    h = HMM(8, D)
    A = np.zeros((10, 10)); A[0, 1] = 1
    for i in range(1, 9):
        A[i, i], A[i, i + 1] = 0.84, 0.16
    A[9, 9] = 1
    h.A = A
    mu = rng.normal(0, 20, (10, D)); mu[:, 0] -= 300
    cov = np.stack([np.cov(rng.normal(0, 10, (D, 60))) + 5 * np.eye(D) for _ in range(10)])
    h.B = {"mean": mu, "covariance": cov}
    models.append(h)
pk = pack_features(fb)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = decode_batch(models, pk, with_best=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"custom decode N={N} W={W}: {1e3*(t1-t0):.1f} ms wall ({N*101/(t1-t0):.3e} frames/s)", flush=True)
