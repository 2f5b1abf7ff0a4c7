"""Dev tool: pruned decoder on bench-like data for the bounding-pass variants selected by the environment
(SAPR_BOUND_WC, SAPR_APPROX): python scripts/time_bound.py [N] [13|39] [emitting states 8|16]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.pipeline import RecognizerPipeline
from sapr_amd.trellis import DiagModelPack
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
wide = len(sys.argv) > 2 and sys.argv[2] == "39"
NS = int(sys.argv[3]) if len(sys.argv) > 3 else (16 if wide else 8)
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, N, seed=1234, device=dev)
lens = np.full(N, bench.N_SAMP, dtype=np.int64)
if wide:
    preset = dict(BENCH, deltas=True, preemph=0.97)
    plan = MfccPlan(**preset, max_frames=bench.T_FRAMES)
    D, S = 39, NS
else:
    plan = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)
    D, S = bench.D, NS
f_all, _ = plan(pcm, lens)
sample = f_all[: 2200 * bench.T_FRAMES].cpu().numpy().reshape(2200, bench.T_FRAMES, D)
models = bench.build_models(sample, n_states=S)
pack = DiagModelPack.from_params(*models, device=dev)
st = _lib.current_stream()
def ev_time(fn, k=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(k):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
full = RecognizerPipeline(plan, pack, lens, mode="full")
fast = RecognizerPipeline(plan, pack, lens)
full.run(pcm); fast.run(pcm)
torch.cuda.synchronize()
same = torch.equal(full.best_word, fast.best_word) and torch.equal(full.best_score, fast.best_score) and torch.equal(full.path, fast.path)
asc, aeps, exs, cslot, ccnt = fast.pruned_views()
inside = bool(((asc - full.scores).abs() <= aeps).all())
print(f"identical={same} inside={inside} survivors/utt={int(ccnt.sum()) / N:.4f} eps median {float(aeps.median()):.3f} max {float(aeps.max()):.3f}"
      f" |approx-exact| max {float((asc - full.scores).abs().max()):.2e}  pruned decode {ev_time(lambda: fast.launch_decode(st)):.3f} ms")
