"""MFCC kernel time vs number of resident workgroups (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.frontend import BENCH, MfccPlan
N = 100000
plan = MfccPlan(**BENCH, max_frames=101)
pcm = torch.rand(N * 16000, device="cuda") - 0.5
lens = np.full(N, 16000)
for grid in (128, 256, 384, 512, 768, 1024):
    for _ in range(2):
        plan(pcm, lens, grid_blocks=grid)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        plan(pcm, lens, grid_blocks=grid)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"grid={grid:5d}  {dt*1e3:7.3f} ms   per-utterance-per-block {dt/(N/grid)*1e6:7.2f} us")
