"""Ad-hoc timing + error report of the MFCC kernel (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd.frontend import BENCH, BENCH39, REFERENCE, MfccPlan, mfcc_batch
from oracle import mfcc_oracle as mo

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sig = mo.synth_utterances(8, 16000, 16000, seed=0)
plan = MfccPlan(**BENCH, max_frames=101)
got = mfcc_batch([s for s in sig], plan)
err = [np.abs(g - mo.mfcc(s, **mo.BENCH)).max() for g, s in zip(got, sig)]
print("bench preset max abs err per utt:", np.round(err, 5))
for name, cfg, n_samp, mf in (("bench13", BENCH, 16000, 101), ("bench13-2pass", BENCH, 16000, 0),
                              ("bench39", BENCH39, 16000, 101), ("bench39-2pass", BENCH39, 16000, 0),
                              ("reference", REFERENCE, 22050, 101), ("reference-2pass", REFERENCE, 22050, 0)):
    plan = MfccPlan(**cfg, max_frames=mf)
    n = N if not name.startswith("reference") else max(N // 4, 1)
    g = torch.Generator(device="cuda").manual_seed(0)
    pcm = (torch.rand(n * n_samp, device="cuda", generator=g) - 0.5)
    lens = np.full(n, n_samp)
    for _ in range(2):
        plan(pcm, lens)
    torch.cuda.synchronize()
    K = 5
    t0 = time.perf_counter()
    for _ in range(K):
        out, fr = plan(pcm, lens)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    frames = int(fr.sum())
    bpf = 4 * cfg["hop_length"] + 4 * plan.d_out
    print(f"{name}: N={n} frames={frames} {dt*1e3:.3f} ms  {frames/dt:.3e} frames/s  "
          f"{frames*bpf/dt/1e9:.1f} GB/s algorithmic  lds={plan.lds_bytes}")
