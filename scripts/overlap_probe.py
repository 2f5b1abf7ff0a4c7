"""Dev probe: does running the MFCC kernel of one half-batch concurrently with the Viterbi kernel of
the other (two HIP streams) beat running them back to back?   python scripts/overlap_probe.py [utts] [grid_blocks_per_cu]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.pipeline import RecognizerPipeline
from sapr_amd.trellis import DiagModelPack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dev = torch.device("cuda", 0)
plan = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)
lens = np.full(n, bench.N_SAMP, dtype=np.int64)
pcm = [bench.synth_pcm(torch, n, seed=s, device=dev) for s in (1, 2)]
f_all, _ = plan(pcm[0], lens)
models = bench.build_models(f_all[: 2200 * bench.T_FRAMES].cpu().numpy().reshape(2200, bench.T_FRAMES, bench.D))
pack = DiagModelPack.from_params(*models, device=dev)
pipes = [RecognizerPipeline(plan, pack, lens) for _ in range(2)]
lib = _lib.load()


def mfcc(p, x, stream, grid):
    _lib.check(lib.sapr_mfcc_batch(p.plan._h, _lib.ptr(x), _lib.ptr(p.sample_offsets), _lib.ptr(p.frame_offsets),
                                   p.n_utts, p.total_frames, _lib.ptr(p.feats), grid, _lib.ptr(p.mfcc_ws),
                                   p.mfcc_ws_bytes, stream), "mfcc")


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h1, h2 = s1.cuda_stream, s2.cuda_stream
for p, x in zip(pipes, pcm):
    p.run(x)
torch.cuda.synchronize()
K = 10


def serial():
    for k in range(K):
        for p, x in zip(pipes, pcm):
            mfcc(p, x, h1, 0)
            p.launch_viterbi(h1)
            p.launch_backtrace(h1)


def overlapped(grid):
    # steady state: MFCC of one half on s1 while the other half's Viterbi runs on s2
    ev_f = [torch.cuda.Event(), torch.cuda.Event()]   # features of half i ready
    ev_v = [torch.cuda.Event(), torch.cuda.Event()]   # Viterbi of half i done reading the features
    for k in range(K):
        for i in (0, 1):
            if k > 0:
                s1.wait_event(ev_v[i])
            mfcc(pipes[i], pcm[i], h1, grid)
            ev_f[i].record(s1)
            s2.wait_event(ev_f[i])
            pipes[i].launch_viterbi(h2)
            pipes[i].launch_backtrace(h2)
            ev_v[i].record(s2)


for name, fn in (("serial", serial), ("overlap grid=default", lambda: overlapped(0)),
                 ("overlap grid=512", lambda: overlapped(512)), ("overlap grid=256", lambda: overlapped(256)),
                 ("overlap grid=640", lambda: overlapped(640))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (2 * K)
    print(f"{name:24s} {dt*1e3:7.3f} ms per half-batch of {n}  -> {n*bench.T_FRAMES/dt/1e6:8.1f} M frames/s", flush=True)
