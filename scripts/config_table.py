"""Kernel time of every BASELINE.json config on one GPU (dev tool; prints a markdown table).
Config 1 is plumbing (one wav through the reference preset), configs 2-5 are timed with HIP events
around pre-allocated launches, inputs resident in HBM."""
import contextlib, io, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from sapr_amd import _lib
from sapr_amd.frontend import BENCH, BENCH39, REFERENCE, MfccPlan
from sapr_amd.pipeline import RecognizerPipeline
from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
from tests._synth import trained_like_models

dev = torch.device("cuda", 0)


def ev_time(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rows = []
# config 1
plan_ref = MfccPlan(**REFERENCE, max_frames=112)
pcm1 = (torch.rand(22050, device=dev) - 0.5)
ms = ev_time(lambda: plan_ref(pcm1, np.array([22050])))
rows.append(("1", "one 1 s file, reference preset (22.05 kHz, n_fft 2048, 128 mels)", f"{ms:.3f} ms per call (launch-bound)", "—"))
# config 2
N = 10000
pcm = bench.synth_pcm(torch, N, seed=0, device=dev)
lens = np.full(N, bench.N_SAMP)
plan = MfccPlan(**BENCH, max_frames=101)
pipe_tmp = None
ms = ev_time(lambda: plan(pcm, lens))
rows.append(("2", "MFCC, 10 000 x 1 s @16 kHz, 13 coefficients", f"{ms:.3f} ms", f"{N*101/ms/1e3:.0f} M"))
del pcm
# config 3 (+ the headline pipeline at 100k)
N = 100000
pcm = bench.synth_pcm(torch, N, seed=1, device=dev)
lens = np.full(N, bench.N_SAMP)
f_all, _ = plan(pcm, lens)
models = bench.build_models(f_all[: 2200 * 101].cpu().numpy().reshape(2200, 101, 13))
pack = DiagModelPack.from_params(*models, device=dev)
pipe = RecognizerPipeline(plan, pack, lens)
st = _lib.current_stream()
pipe.run(pcm)
ms_v = ev_time(lambda: (pipe.launch_viterbi(st), pipe.launch_backtrace(st)))
rows.append(("3", "Viterbi + back-trace, 100 000 x 101 x 13, 11 words x 8 states", f"{ms_v:.3f} ms", f"{N*101/ms_v/1e3:.0f} M"))
ms_p = ev_time(lambda: pipe.run(pcm))
rows.append(("2+3", "headline pipeline (bench.py), 100 000 utterances", f"{ms_p:.3f} ms", f"{N*101/ms_p/1e3:.0f} M"))
# config 4: hmmlearn-compatible E-step, 10 words
feats = pipe.feats.clone()
batch = FeatureBatch.from_packed(feats, np.full(N, 101))
sp, A, mu, cv = trained_like_models(10, 8, 13, seed=3)
mu[:, :, 0] += 0.0
es = EStep(batch, np.arange(N) % 10, 10, 10)
pk = DiagModelPack.from_params(sp, A, mu, cv)
ms_e = ev_time(lambda: es.run(pk))
rows.append(("4", "Baum-Welch E-step (hmmlearn-compatible), 100 000 utterances, 10 words", f"{ms_e:.3f} ms per iteration", f"{N*101/ms_e/1e3:.0f} M"))
del pipe, es, batch, feats
torch.cuda.empty_cache()
# config 5: 39-dim features, 16-state models, one 100 000-utterance chunk of the 1 M
plan39 = MfccPlan(**BENCH39, max_frames=101)
sp, A, mu, cv = trained_like_models(11, 16, 39, seed=5)
pack39 = DiagModelPack.from_params(sp, A, mu, cv)
pipe39 = RecognizerPipeline(plan39, pack39, lens)
pipe39.run(pcm)
ms_m = ev_time(lambda: pipe39.launch_mfcc(pcm, st))
ms_5 = ev_time(lambda: pipe39.run(pcm))
rows.append(("5", "full pipeline, 39-dim MFCC+Δ+ΔΔ (pre-emphasis), 11 words x 16 states, per 100 000-utterance chunk",
             f"{ms_5:.3f} ms (MFCC {ms_m:.2f})", f"{N*101/ms_5/1e3:.0f} M"))
print("| config | workload | kernel time | frames/s |")
print("|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
