"""Pruned decode only, a few launches, for rocprofv3 --kernel-trace (dev tool): python scripts/time_decode_only.py [N]"""
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
fast = RecognizerPipeline(plan, pack, lens)
fast.run(pcm)
for _ in range(6):
    fast.launch_decode(st)
torch.cuda.synchronize()
asc, aeps, exs, cslot, ccnt = fast.pruned_views()
print("candidates", int(ccnt.sum()), "eps median", float(aeps.median()))
