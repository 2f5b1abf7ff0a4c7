"""Dev probe: K steps of the headline pipeline (MFCC -> pruned decode, 100 000 utterances each) back to back on one
stream against the same K steps software-pipelined over two streams (the decode of step k under the MFCC of step k + 1,
two sets of buffers).   python scripts/overlap_steps.py [utts] [steps]"""
import sys, time
import ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.pipeline import RecognizerPipeline
from sapr_amd.trellis import DiagModelPack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
plan = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)
lens = np.full(n, bench.N_SAMP, dtype=np.int64)
pcm = bench.synth_pcm(torch, n, seed=1, device=dev)
f_all, _ = plan(pcm, lens)
models = bench.build_models(f_all[: 2200 * bench.T_FRAMES].cpu().numpy().reshape(2200, bench.T_FRAMES, bench.D))
del f_all
pack = DiagModelPack.from_params(*models, device=dev)
pipes = [RecognizerPipeline(plan, pack, lens, mode="pruned") for _ in range(2)]
cur = _lib.current_stream()
for p in pipes:
    p.run(pcm)
torch.cuda.synchronize()
ref = [pipes[0].best_word.clone(), pipes[0].best_score.clone(), pipes[0].path.clone()]

t0 = time.perf_counter()
for k in range(K):
    p = pipes[k % 2]
    p.launch_mfcc(pcm, cur)
    p.launch_decode(cur)
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / K * 1e3

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h1, h2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)
ev_m = [torch.cuda.Event() for _ in range(2)]
ev_d = [torch.cuda.Event() for _ in range(2)]
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    i = k % 2
    p = pipes[i]
    if k >= 2:
        s1.wait_event(ev_d[i])       # the features of step k - 2 have been decoded
    with torch.cuda.stream(s1):
        p.launch_mfcc(pcm, h1)
        ev_m[i].record(s1)
    s2.wait_event(ev_m[i])
    with torch.cuda.stream(s2):
        p.launch_decode(h2)
        ev_d[i].record(s2)
torch.cuda.synchronize()
ovl = (time.perf_counter() - t0) / K * 1e3
same = all(bool(torch.equal(a, b)) for a, b in zip(ref, [pipes[(K - 1) % 2].best_word, pipes[(K - 1) % 2].best_score, pipes[(K - 1) % 2].path]))
print(f"{n} utterances x {K} steps: back to back {seq:.3f} ms/step, two streams {ovl:.3f} ms/step, results identical {same}")
