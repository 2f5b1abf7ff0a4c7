"""Dev timing of the bench-preset MFCC launch only: python scripts/time_mfcc_q.py [N]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.frontend import BENCH, MfccPlan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
plan = MfccPlan(**BENCH, max_frames=101)
g = torch.Generator(device="cuda").manual_seed(0)
pcm = (torch.rand(N * 16000, device="cuda", generator=g) - 0.5)
lens = np.full(N, 16000)
for _ in range(2):
    plan(pcm, lens)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out, fr = plan(pcm, lens)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
import os
print(f"ablate={os.environ.get('SAPR_Q_ABLATE','0'):>3} core={os.environ.get('SAPR_MFCC_CORE','q'):>3}: {dt*1e3:.3f} ms")
