"""BASELINE configs[4] end to end on one GPU: 1 M synthetic utterances (16 kHz, 1 s, int16 PCM in pinned host
memory) -> 39-dim MFCC+d+dd -> pruned Viterbi vs 11 word models x 18 states, in chunks of 100 000 utterances with
the upload of chunk k+1 overlapped with the kernels of chunk k.  Prints PCIe-inclusive and kernel-only frames/s.
    python scripts/stream_1m.py [n_chunks=10] [chunk_utts=100000]
The ten chunks are ONE synthetic chunk re-used (host memory: 3.2 GB instead of 32 GB); every chunk is uploaded,
converted, analysed and decoded in full."""
import json, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd.frontend import BENCH39, MfccPlan
from sapr_amd.stream import StreamingRecognizer
from sapr_amd.trellis import DiagModelPack
n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, n, seed=1, device=dev)
lens = np.full(n, bench.N_SAMP, dtype=np.int64)
plan = MfccPlan(**BENCH39, max_frames=bench.T_FRAMES)
f, _ = plan(pcm[: 2200 * bench.N_SAMP], lens[:2200])
models = bench.build_models(f.cpu().numpy().reshape(2200, bench.T_FRAMES, 39), n_states=16)
pack = DiagModelPack.from_params(*models, device=dev)
pcm16 = torch.clamp((pcm * 32768.0).round(), -32768, 32767).to(torch.int16).cpu().pin_memory()
del pcm, f
rec = StreamingRecognizer(plan, pack, device=dev)
rec.run([(pcm16, lens)] * 2, keep_results=False)   # warm-up: allocations, pipelines
hist = np.zeros(bench.W + 1, dtype=np.int64)
def on_result(k, bw, bs, path):
    hist[:] += np.bincount(bw + 1, minlength=bench.W + 1)
_, rep = rec.run([(pcm16, lens)] * n_chunks, on_result=on_result, keep_results=False)
print(json.dumps({"workload": f"{n_chunks} chunks x {n} utterances x 1 s, int16 PCM from pinned host memory, 39-dim / 18 states",
                  "utterances": rep.n_utts, "frames": rep.frames, "wall_s": rep.wall_s,
                  "frames_per_s_pcie_inclusive": rep.frames_per_s_pcie_inclusive,
                  "frames_per_s_kernels_only": rep.frames_per_s_kernels_only,
                  "kernel_ms_per_chunk": float(np.mean(rep.chunk_kernel_ms)),
                  "h2d_GBps_sample": 2 * 2 * n * bench.N_SAMP / rep.h2d_s / 1e9 if rep.h2d_s else None,
                  "words_histogram": hist.tolist()}))
