import sys; sys.path.insert(0,'.')
import numpy as np, torch, bench
from oracle import mfcc_oracle as mo
from sapr_amd.frontend import REFERENCE, MfccPlan
dev=torch.device('cuda',0)
r=bench.extra_mfcc_reference_preset(torch, dev, 2000)
print(r)
pcm = bench.synth_pcm(torch, 4, seed=99, device=dev, SR=22050, N_SAMP=22050)
plan = MfccPlan(**REFERENCE, max_frames=101)
f,fr = plan(pcm, np.full(4,22050))
h = pcm.cpu().numpy().reshape(4,22050)
w = np.concatenate([mo.mfcc(y, **mo.REFERENCE).T for y in h])
g = f.cpu().numpy()
print(g[:2,:4], w[:2,:4], np.abs(g-w).max(), fr)
