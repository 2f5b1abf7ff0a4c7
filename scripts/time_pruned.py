"""Ad-hoc timing of the pruned vs the all-vocabulary decoder on bench-like data (dev tool):
python scripts/time_pruned.py [N]   — features from the MFCC kernel, models from bench.build_models"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.pipeline import RecognizerPipeline
from sapr_amd.trellis import DiagModelPack
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, N, seed=1234, device=dev)
lens = np.full(N, bench.N_SAMP, dtype=np.int64)
plan = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)
f_all, _ = plan(pcm, lens)
models = bench.build_models(f_all[: 2200 * bench.T_FRAMES].cpu().numpy().reshape(2200, bench.T_FRAMES, bench.D))
pack = DiagModelPack.from_params(*models, device=dev)
st = _lib.current_stream()


def ev_time(fn, k=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k


full = RecognizerPipeline(plan, pack, lens, mode="full")
fast = RecognizerPipeline(plan, pack, lens)
full.run(pcm); fast.run(pcm)
torch.cuda.synchronize()
print("identical:", torch.equal(full.best_word, fast.best_word), torch.equal(full.best_score, fast.best_score),
      torch.equal(full.path, fast.path))
asc, aeps, exs, cslot, ccnt = fast.pruned_views()
print("candidates per word:", ccnt.cpu().numpy().tolist(), "total", int(ccnt.sum()), "of", N * pack.W,
      " eps median", float(aeps.median()), "max", float(aeps.max()),
      " |approx-exact| max", float((asc - full.scores).abs().max()))
print(f"mfcc            {ev_time(lambda: full.launch_mfcc(pcm, st)):.3f} ms")
print(f"full decode     {ev_time(lambda: full.launch_decode(st)):.3f} ms")
print(f"pruned decode   {ev_time(lambda: fast.launch_decode(st)):.3f} ms  ({len(fast._pieces)} pieces)")
for n_p in (1, 2, 3, 6, 8):
    pp = RecognizerPipeline(plan, pack, lens, pieces=n_p)
    pp.feats.copy_(full.feats)
    pp.launch_decode(st); torch.cuda.synchronize()
    same = torch.equal(pp.best_word, full.best_word) and torch.equal(pp.best_score, full.best_score) and torch.equal(pp.path, full.path)
    print(f"pruned decode, {n_p} piece(s): {ev_time(lambda: pp.launch_decode(st)):.3f} ms  identical={same}")
    del pp
