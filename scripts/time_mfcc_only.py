"""Kernel time of the bench-preset MFCC launch (HIP events), whatever library SAPR_LIB selects (dev tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, MfccPlan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
plan = MfccPlan(**BENCH, max_frames=101)
pcm = torch.rand(N * 16000, device="cuda") - 0.5
lens = np.full(N, 16000)
for _ in range(3):
    plan(pcm, lens)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(10):
    plan(pcm, lens)
b.record()
torch.cuda.synchronize()
print(f"{os.path.basename(_lib.LIB_PATH)}: {a.elapsed_time(b) / 10:.3f} ms")
