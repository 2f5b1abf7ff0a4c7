"""Kernel time of the bench-preset MFCC launch sequence with pre-allocated buffers (HIP events on the launch
stream), whatever library SAPR_LIB / core SAPR_MFCC_CORE select (dev tool).  usage: time_mfcc_launch.py [N] [39]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, BENCH39, MfccPlan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = BENCH39 if len(sys.argv) > 2 and sys.argv[2] == "39" else BENCH
plan = MfccPlan(**cfg, max_frames=101)
lib = _lib.load()
pcm = torch.rand(N * 16000, device="cuda") - 0.5
so = torch.arange(N + 1, dtype=torch.int64, device="cuda") * 16000
fo = torch.arange(N + 1, dtype=torch.int64, device="cuda") * 101
out = torch.empty((N * 101, plan.d_out), dtype=torch.float32, device="cuda")
ws, wsb = plan.workspace(N * 101, N, out.device)
st = _lib.current_stream()
def run():
    _lib.check(lib.sapr_mfcc_batch(plan._h, _lib.ptr(pcm), _lib.ptr(so), _lib.ptr(fo), N, N * 101, _lib.ptr(out), 0,
                                   _lib.ptr(ws), wsb, st), "mfcc")
for _ in range(3):
    run()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(10):
    run()
b.record()
torch.cuda.synchronize()
print(f"{os.path.basename(_lib.LIB_PATH)} core={os.environ.get('SAPR_MFCC_CORE', 'default')} d_out={plan.d_out} "
      f"two_pass={plan.two_pass} lds={plan.lds_bytes}: {a.elapsed_time(b) / 10:.3f} ms per launch sequence", flush=True)
