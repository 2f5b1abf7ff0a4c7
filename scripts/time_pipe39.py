"""39-dim / 18-state pipeline on one 100 000-utterance chunk (BASELINE configs[4]) for profiling (dev tool)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
r = bench.extra_pipeline39(torch, torch.device("cuda", 0), bench.synth_pcm(torch, 100000, seed=1, device=torch.device("cuda", 0)), 100000)
print(r)
