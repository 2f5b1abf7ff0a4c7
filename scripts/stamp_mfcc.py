"""Per-phase s_memtime shares of the MFCC kernel (diagnostic build; dev tool)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, MfccPlan
N = 100000
plan = MfccPlan(**BENCH, max_frames=101)
pcm = torch.rand(N * 16000, device="cuda") - 0.5
lens = np.full(N, 16000)
out, fr = plan(pcm, lens)
so = torch.from_numpy(np.arange(N + 1, dtype=np.int64) * 16000).cuda()
fo = torch.from_numpy(np.arange(N + 1, dtype=np.int64) * 101).cuda()
grid = 512
st = torch.zeros(grid * 4 * 12, dtype=torch.int64, device="cuda")
lib = _lib.load()
_lib.check(lib.sapr_mfcc_batch_stamped(plan._h, _lib.ptr(pcm), _lib.ptr(so), _lib.ptr(fo), N, _lib.ptr(out), grid,
                                       _lib.ptr(st), _lib.current_stream()), "stamped")
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(grid, 4, 12).astype(np.float64)
names = ["loop/prev", "wait samples+window", "FFT A+transpose+FFT B", "barrier1", "untangle+power", "issue loads",
         "barrier2", "mel MFMA+log", "barrier3", "utt max", "DCT", "deltas+store"]
tot = s.sum(axis=2).mean()
print(f"mean cycles per wave total: {tot:.3e}  (per utterance {tot/ (N/grid):.0f}; s_memtime ticks)")
for w in range(4):
    print("wave", w, " ".join(f"{v/ s[:, w].sum(axis=1).mean()*100:5.1f}%" for v in s[:, w].mean(axis=0)))
for i, n in enumerate(names):
    print(f"{n:26s} {s[:, :, i].mean()/tot*100:6.2f}%   per-utt {s[:, :, i].mean()/(N/grid):9.0f}")
