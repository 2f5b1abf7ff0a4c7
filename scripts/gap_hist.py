"""How far apart are the best and second-best words' exact Viterbi scores on the bench data (dev tool)?  Decides how
wide a bounding pass's intervals may be before the pruned decoder keeps noticeably more than one word."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd.frontend import BENCH, MfccPlan
from sapr_amd.pipeline import RecognizerPipeline
from sapr_amd.trellis import DiagModelPack
N = 100000
dev = torch.device("cuda", 0)
pcm = bench.synth_pcm(torch, N, seed=1234, device=dev)
lens = np.full(N, bench.N_SAMP, dtype=np.int64)
plan = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)
f_all, _ = plan(pcm, lens)
models = bench.build_models(f_all[: 2200 * bench.T_FRAMES].cpu().numpy().reshape(2200, bench.T_FRAMES, bench.D))
pack = DiagModelPack.from_params(*models, device=dev)
full = RecognizerPipeline(plan, pack, lens, mode="full")
full.run(pcm)
sc = full.scores.sort(dim=1, descending=True).values
for k in (1, 2, 3):
    gap = (sc[:, 0] - sc[:, k]).cpu().numpy()
    print(f"gap best - #{k+1}: " + "  ".join(f"P(<{e})={np.mean(gap < e):.4f}" for e in (0.06, 0.2, 0.5, 1, 2, 4, 8)))
