#!/usr/bin/env python3
"""Headline benchmark: frames/sec through MFCC + all-vocabulary Viterbi (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic audio already resident in HBM:
  PCM (16 kHz, 1 s / utterance) --sapr_mfcc_batch--> 13 MFCC x 101 frames
      --sapr_viterbi_decode_pruned (W=11 word models, 8 emitting states + entry/exit)-->
      best word, its Viterbi score and its state path per utterance (decoder.py:35-49)
i.e. BASELINE configs[1] (batched MFCC) feeding configs[2] (decoder.py-API Viterbi), which is
the combination the metric "frames/sec MFCC+Viterbi (16kHz, 13-MFCC, 8-state HMM)" is quoted on.
A frame is counted once however many word models score it (decoder.py:42 semantics).  The pruned
decoder returns the same bits as scoring every word exactly (tests/test_viterbi_gpu.py, test_fullsize_gpu.py);
`--decode full` times the all-vocabulary evaluation instead.

Multi-GPU: utterances shard across ranks with NO data-path collective (weak scaling: every rank
owns --utts utterances); value = frames of all ranks / max-over-ranks time.

`--mode em` times BASELINE configs[3] instead: one Baum-Welch iteration = E-step over this rank's shard of
utterances (hmmlearn-compatible kernels, 10 word models) + ONE all-reduce of the sufficient statistics
(RCCL) + the M-step on every rank; the all-reduce time is reported separately.

The JSON line also carries `roofline` (dominant kernel: algorithmic bytes / HIP-event time vs the 8 TB/s HBM
peak, plus its flop rate vs the float32 vector peak, and what actually limits it), `cpu_baseline` (the oracle
on one host core), `cpu_baseline_all_cores` (the same sample over every core this process may use) and `extra`
(the other BASELINE configs on one GPU, each with a parity flag), rank 0 / N=1 only.
"""
import os

os.environ.setdefault("OMP_NUM_THREADS", "1")        # cpu_baseline is a single-thread port
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import argparse  # noqa: E402
import json  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, N_SAMP, HOP, T_FRAMES, D, W, N_STATES = 16000, 16000, 160, 101, 13, 11, 8
HBM_PEAK_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md, chip table (spec)
FP32_VALU_PEAK_TFLOPS = 157.3  # same table: vector FP32 = matrix FP32 (f32-input MFMA)
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X datasheet vector FP64 (not in the local guide)
N_SIMDS, CLOCK_HZ = 1024, 2.4e9  # 256 CUs x 4 SIMDs, nominal clock (same chip table)
VALU_SAT_PER_CYCLE = 0.238    # VALU wave-instructions per NOMINAL cycle and SIMD that a loop of nothing but the spectral
                              # kernel's own vector-instruction mix sustains at the kernel's occupancy (4 wavefronts per
                              # SIMD): scripts/ubench/mix_rate, test mix_set13_valu_only — 16.2-17.3 nominal cycles per
                              # instruction and wavefront (0.33 per real cycle: the chip clocks down to ~1.75 GHz under
                              # this load).  Plain v_add/mul/fma_f32 alone: 0.51 per real cycle; DPP and packed
                              # forms 0.32; DESIGN 6 has the table.  (Rounds 1-3 divided by 0.40, v_fma_f32 only.)
MFCC_FLOP_PER_FRAME = 35.0e3  # rFFT-512 11.5 k + window/power 1.5 k + 40x257 mel 20.6 k + log/DCT 1.1 k
BYTES_PER_FRAME = {"mfcc": 4 * HOP + 4 * D,   # fp32 PCM hop in + 13 fp32 out        
                   "decode": 4 * D + 4}       # fp32 features in + int32 state out   


def synth_pcm(torch, n_utts, seed, device, SR=SR, N_SAMP=N_SAMP):
    """SURVEY §8(d) config-2 generator on the device: three sinusoids (100-4000 Hz, random phase,
    amplitude U(0.05,0.3)) + N(0,0.01^2) noise; first/last 100 ms zeroed in 10 % of utterances."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty(n_utts * N_SAMP, dtype=torch.float32, device=device)
    t = torch.arange(N_SAMP, device=device, dtype=torch.float32) / SR
    chunk = 8192
    for u0 in range(0, n_utts, chunk):
        n = min(chunk, n_utts - u0)
        f = 100.0 + 3900.0 * torch.rand(n, 3, 1, device=device, generator=g)
        ph = 6.2831853 * torch.rand(n, 3, 1, device=device, generator=g)
        a = 0.05 + 0.25 * torch.rand(n, 3, 1, device=device, generator=g)
        y = (a * torch.sin(6.2831853 * f * t + ph)).sum(1)
        y += 0.01 * torch.randn(n, N_SAMP, device=device, generator=g)
        sil = torch.rand(n, 1, device=device, generator=g) < 0.1
        edge = torch.zeros(N_SAMP, dtype=torch.bool, device=device)
        edge[: SR // 10] = True
        edge[-SR // 10:] = True
        y = torch.where(sil & edge, torch.zeros_like(y), y)
        out[u0 * N_SAMP:(u0 + n) * N_SAMP] = y.reshape(-1)
    return out


def build_models(feats_3d, n_words=W, n_states=N_STATES):
    """Word models by uniform segmentation of a few hundred utterances (plumbing, numpy):
    state s of word w = mean / variance of the frames of segment s over utterances u % W == w;
    bidiagonal transitions with a_ii = exp(-1/(T/N_s - 1)) (hmmlearn_hmm.py:45-78)."""
    n, T, d = feats_3d.shape
    S = n_states + 2
    seg = np.minimum((np.arange(T) * S) // T, S - 1)
    means = np.empty((n_words, S, d))
    covars = np.empty((n_words, S, d))
    for w in range(n_words):
        x = feats_3d[w::n_words].astype(np.float64)
        for s in range(S):
            fr = x[:, seg == s].reshape(-1, d)
            means[w, s] = fr.mean(0)
            covars[w, s] = fr.var(0) + 1.0
    aii = np.exp(-1.0 / (T / n_states - 1.0))
    A = np.zeros((S, S))
    A[0, 1] = 1.0
    for i in range(1, n_states + 1):
        A[i, i], A[i, i + 1] = aii, 1 - aii
    A[S - 1, S - 1] = 1.0
    sp = np.zeros(S)
    sp[0] = 1.0
    return np.tile(sp, (n_words, 1)), np.tile(A, (n_words, 1, 1)), means, covars


# ------------------------------------------------------------------------------------ CPU baseline
def _cpu_chunk(args):
    """Oracle over one chunk of utterances (runs in a worker process or in the caller): numpy MFCC
    restatement (librosa chain) + C Viterbi restatement (hmmlearn) for all W models."""
    pcm_host, models = args
    from oracle import c_oracle, mfcc_oracle as mo
    sp, A, mu, cv = models
    c_oracle.load()
    t0 = time.perf_counter()
    feats = [mo.mfcc(y, **mo.BENCH).T for y in pcm_host]
    t_mfcc = time.perf_counter() - t0
    packed = np.ascontiguousarray(np.concatenate(feats, axis=0), dtype=np.float32)
    offs = np.r_[0, np.cumsum([f.shape[0] for f in feats])].astype(np.int64)
    t0 = time.perf_counter()
    sc, bw, path = c_oracle.decode_batch(packed, offs, sp, A, mu, cv, tie=1, sum_order=1)
    t_vit = time.perf_counter() - t0
    return t_mfcc, t_vit, packed, offs, bw, path, sc


def cpu_baseline(pcm_host, models):
    t_mfcc, t_vit, packed, offs, bw, path, sc = _cpu_chunk((pcm_host, models))
    frames = int(offs[-1])
    n = len(pcm_host)
    return {"value": frames / (t_mfcc + t_vit), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} utterances x 1 s ({frames} frames): numpy MFCC restatement "
                      f"{t_mfcc:.2f} s + C Viterbi restatement x{W} models {t_vit:.2f} s, 1 thread "
                      f"of {os.cpu_count()} host cpus",
            "note": "the Viterbi half is a C port of the hmmlearn recursion, far faster than the reference's "
                    "Python loops (BASELINE.md section 2 implies ~4.6 k frames/s for them): the GPU/CPU ratio taken "
                    "from this line is conservative"}, (packed, offs, bw, path, sc)


def end_to_end_agreement(o_word, o_score, o_path, offs, g_word, g_score, g_path):
    """Oracle front-end + oracle decoder vs HIP front-end + HIP decoder on the same utterances: fraction of equal
    best words, of utterances whose whole state path is equal, of equal path frames, and the largest score gap."""
    n = len(o_word)
    same_path = np.array([np.array_equal(o_path[offs[i]:offs[i + 1]], g_path[offs[i]:offs[i + 1]]) for i in range(n)])
    with np.errstate(invalid="ignore"):
        gap = np.abs(np.asarray(o_score, dtype=np.float64) - np.asarray(g_score, dtype=np.float64))
    rel = gap / np.maximum(np.abs(o_score), 1e-300)
    return {"utterances": int(n), "best_word_equal_frac": float(np.mean(o_word == g_word)),
            "path_equal_frac": float(np.mean(same_path)),
            "path_frames_equal_frac": float(np.mean(o_path[: offs[n]] == g_path[: offs[n]])),
            "max_abs_score_diff": float(np.nanmax(gap)) if n else 0.0,
            "max_rel_score_diff": float(np.nanmax(rel)) if n else 0.0,
            "what": "oracle MFCC -> oracle Viterbi vs HIP MFCC -> HIP Viterbi (the features differ in the last "
                    "float32 bits: an agreement rate, not a bit-for-bit gate)"}


def usable_cores():
    """Worker processes for the all-cores baseline: the CPUs this process may run on, capped by the cgroup CPU
    quota when one is visible and otherwise by 16 per GPU (the CPU share of a one-GPU box in this pool: 64 workers
    measured only 12x one worker there)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = int(txt[0]) / int(txt[1])
            else:
                q = int(txt[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError, IndexError):
            continue
        break
    cap = int(quota) if quota and quota >= 1 else 16
    return max(1, min(n, cap))


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_chunk_shared(args):
    """Worker of the all-cores baseline: its utterances come from a memory-mapped file (no pickling of audio)."""
    path, lo, hi, models = args
    pcm = np.load(path, mmap_mode="r")
    t_mfcc, t_vit, _, offs, *_ = _cpu_chunk((pcm[lo:hi], models))
    return t_mfcc, t_vit, int(offs[-1])


def cpu_baseline_all_cores(pcm_host, models, cores):
    """The same oracle, one worker process per usable core (utterances are independent): wall time of the
    whole pool over `len(pcm_host)` utterances; pool start-up is excluded by a warm-up task per worker and the
    audio is shared through a memory-mapped file."""
    import multiprocessing as mp
    import tempfile
    ctx = mp.get_context("spawn")   # never fork a process that has initialised the GPU
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    fd, path = tempfile.mkstemp(suffix=".npy", prefix="sapr_bench_pcm_", dir=shm)
    os.close(fd)
    try:
        np.save(path, pcm_host)
        n = len(pcm_host)
        cuts = np.linspace(0, n, cores + 1).astype(int)
        # every wait is bounded: a pool that cannot start its workers must cost the bench line one field, not hang it
        with ctx.Pool(cores) as pool:
            pool.map_async(_cpu_chunk_shared, [(path, 0, 2, models)] * cores).get(timeout=180)   # imports, warm-up
            t0 = time.perf_counter()
            res = pool.map_async(_cpu_chunk_shared,
                                 [(path, int(a), int(b), models) for a, b in zip(cuts[:-1], cuts[1:]) if b > a],
                                 chunksize=1).get(timeout=300)
            wall = time.perf_counter() - t0
    finally:
        os.unlink(path)
    frames = int(sum(r[2] for r in res))
    return {"value": frames / wall, "unit": "frames/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "host_cpus": os.cpu_count(), "numpy": np.__version__,
            "sample": f"{len(pcm_host)} utterances x 1 s ({frames} frames) over {cores} worker processes "
                      f"(1 thread each) in {wall:.2f} s wall; busiest worker: MFCC "
                      f"{max(r[0] for r in res):.2f} s + Viterbi {max(r[1] for r in res):.2f} s"}


# ------------------------------------------------------------------------------------------- extras
def _ev_ms(torch, fn, k):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k


def extra_em_hmmlearn(torch, dev, feats, n_utts, n_words=10):
    """configs[3] on one GPU, hmmlearn-compatible path: E-step kernels over n_utts utterances / 10 words +
    D2H of the statistics + host M-step, per EM iteration; parity = statistics of a 60-utterance sample
    against the numpy restatement (rtol 1e-9)."""
    from oracle import hmmlearn_oracle as ho
    from sapr_amd.hmmlearn_hmm import m_step_batch
    from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
    S = N_STATES + 2
    f3 = feats.view(n_utts, T_FRAMES, D)
    models = build_models(f3[:2000].cpu().numpy(), n_words=n_words)
    sp, A, mu, cv = models
    batch = FeatureBatch.from_packed(feats, np.full(n_utts, T_FRAMES))
    utt_model = np.arange(n_utts) % n_words
    es = EStep(batch, utt_model, n_words, S)
    pack = DiagModelPack.from_params(sp, A, mu, cv, device=dev)
    kernel_ms = _ev_ms(torch, lambda: es.run(pack), 5)

    def iteration():
        p = DiagModelPack.from_params(sp, A, mu, cv, device=dev, exact_only=True)  # as fit_models packs per iteration
        host = es.run(p).cpu().numpy()
        return m_step_batch(host, es.S, es.D, sp, A, mu, cv)                        # as fit_models updates the vocabulary
    iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / 3 * 1e3
    # parity on a sample: the first 60 utterances as their own batch
    n_s = 60
    sb = FeatureBatch.from_packed(feats[: n_s * T_FRAMES].contiguous(), np.full(n_s, T_FRAMES))
    ss = EStep(sb, utt_model[:n_s], n_words, S)
    got = ss.run(pack).cpu().numpy()
    host_f = f3[:n_s].cpu().numpy()
    ok = True
    for w in range(n_words):
        ref = ho.new_stats(S, D)
        lp = sum(ho.accumulate(ref, host_f[u], sp[w], A[w], mu[w], cv[w]) for u in range(n_s) if utt_model[u] == w)
        st = ss.split(got[w])
        ok &= bool(abs(st["logprob"] - lp) <= 1e-9 * abs(lp))
        for k, ko in (("start", "start"), ("trans", "trans"), ("post", "post"), ("obs", "obs"), ("obs**2", "obs2")):
            ok &= bool(np.allclose(st[k], ref[ko], rtol=1e-9, atol=1e-9))
    frames = n_utts * T_FRAMES
    return {"workload": f"configs[3], 1 GPU: Baum-Welch iteration, {n_utts} utterances x {T_FRAMES} frames, {n_words} "
                        f"words x {S} states, hmmlearn-compatible (GaussianHMM.fit semantics, hmmlearn_hmm.py:103)",
            "estep_kernels_ms": kernel_ms, "iteration_wall_ms": wall_ms, "frames_per_s": frames / (wall_ms * 1e-3),
            "parity_vs_oracle_sample": ok}


def extra_em_custom(torch, dev, feats, n_utts):
    """configs[3], the reference's from-scratch algorithm: HMM.baum_welch (custom_hmm.py:402-460) wall time per
    iteration on n_utts utterances of one word, host work included; parity = 2-iteration log-likelihood
    history of a 24-utterance sample against the pinned numpy restatement (rtol 1e-8)."""
    import contextlib
    import io
    from oracle import custom_hmm_oracle as co
    from sapr_amd.custom_hmm import HMM, pack_features
    from sapr_amd.trellis import FeatureBatch
    pk = pack_features(FeatureBatch.from_packed(feats, np.full(n_utts, T_FRAMES)))
    with contextlib.redirect_stdout(io.StringIO()):
        HMM(N_STATES, D, feature_set=pk, model_name="bench")   # warm-up: the first call pays the allocator's hipMallocs
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        h = HMM(N_STATES, D, feature_set=pk, model_name="bench")
        torch.cuda.synchronize()
        flat_ms = (time.perf_counter() - t0) * 1e3
        h.baum_welch(pk, max_iter=1)                      # warm-up (allocations)
        h = HMM(N_STATES, D, feature_set=pk, model_name="bench")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = h.baum_welch(pk, max_iter=3)
        torch.cuda.synchronize()
        it_ms = (time.perf_counter() - t0) / max(len(hist), 1) * 1e3
        # parity sample
        n_s = 24
        host = feats.view(n_utts, T_FRAMES, D)[:n_s].cpu().numpy()
        lst = [np.ascontiguousarray(x.T) for x in host]
        hs = HMM(N_STATES, D, feature_set=lst, model_name="sample")
        fs = co.flat_start(lst, N_STATES)
        with np.errstate(all="ignore"):
            got = hs.baum_welch(lst, max_iter=2)
            want, *_ = co.baum_welch(lst, fs["A"], fs["mean"], fs["covariance"], fs["global_covariance"], 0.001,
                                     max_iter=2)
    ok = bool(np.allclose(got, want, rtol=1e-8, equal_nan=True)) and bool(np.array_equal(hs.global_mean, fs["global_mean"]))
    frames = n_utts * T_FRAMES
    return {"workload": f"configs[3], 1 GPU: custom_hmm.HMM flat start + baum_welch, {n_utts} utterances x {T_FRAMES} "
                        f"frames, 1 word x {N_STATES + 2} states, full covariances (custom_hmm.py:35-116,402-460)",
            "flat_start_wall_ms": flat_ms, "iteration_wall_ms": it_ms, "frames_per_s": frames / (it_ms * 1e-3),
            "parity_vs_oracle_sample": ok}


def extra_decode_custom(torch, dev, feats, n_utts, n_words=W):
    """configs[2] with the reference's from-scratch models: custom_hmm.HMM.decode (custom_hmm.py:462-514: full-covariance
    emission in the reference's evaluation order, trellis over the first D frames — its quirk) for n_utts utterances x
    n_words models + Decoder.decode_sequence's arg-max (decoder.py:35-49), one launch sequence; parity = scores and
    paths of a 6-utterance sample against the pinned numpy restatement in the same evaluation order, bit for bit."""
    import contextlib
    import io
    from oracle import custom_hmm_oracle as co
    from sapr_amd.custom_hmm import HMM, decode_batch, pack_features
    from sapr_amd.trellis import FeatureBatch
    pk = pack_features(FeatureBatch.from_packed(feats, np.full(n_utts, T_FRAMES)))
    f3 = feats.view(n_utts, T_FRAMES, D)
    models = []
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        for w in range(n_words):   # small per-word training sets: distinct, trained models
            lst = [np.ascontiguousarray(x.T) for x in f3[w:2000:n_words][:40].cpu().numpy()]
            h = HMM(N_STATES, D, feature_set=lst, model_name=f"w{w}")
            h.baum_welch(lst, max_iter=2)
            models.append(h)
    decode_batch(models, pk, with_best=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sc, paths, bw, bs, bp = decode_batch(models, pk, with_best=True)
    ms = (time.perf_counter() - t0) * 1e3      # incl. the device-to-host copies of scores and paths
    n_s, ok = 6, True
    host = f3[:n_s].cpu().numpy()
    with np.errstate(all="ignore"):
        for u in range(n_s):
            for w, h in enumerate(models):
                lp, path = co.decode(np.ascontiguousarray(host[u].T), h.A, h.B["mean"], h.B["covariance"], N_STATES,
                                     gram="chain")
                ok &= bool(lp == sc[u, w] or (np.isnan(lp) and np.isnan(sc[u, w]))) and bool(np.array_equal(path, paths[u, w]))
    frames = n_utts * T_FRAMES
    return {"workload": f"{n_utts} utterances x {n_words} custom_hmm.HMM models ({N_STATES + 2} states, full covariances): "
                        "HMM.decode for every pair + Decoder arg-max (custom_hmm.py:462-514, decoder.py:35-49), "
                        "wall incl. D2H of all scores and paths",
            "ms": ms, "frames_per_s": frames / (ms * 1e-3), "best_word_histogram": np.bincount(bw[bw >= 0], minlength=n_words).tolist(),
            "parity_vs_oracle_sample": ok}


def extra_pipeline39(torch, dev, pcm, n_utts):
    """configs[4], one 100 k-utterance chunk on one GPU: pre-emphasis + 39-dim MFCC+delta+delta-delta ->
    pruned Viterbi vs 11 word models x 16 emitting states; parity = a 48-utterance sample against the oracle
    (features to 1e-3, words / scores / paths of the oracle's own decode bit for bit)."""
    from oracle import c_oracle, mfcc_oracle as mo
    from sapr_amd import _lib
    from sapr_amd.frontend import BENCH39, MfccPlan
    from sapr_amd.pipeline import RecognizerPipeline
    from sapr_amd.trellis import DiagModelPack
    lens = np.full(n_utts, N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH39, max_frames=T_FRAMES)
    f_all, _ = plan(pcm, lens)
    models = build_models(f_all[: 2200 * T_FRAMES].cpu().numpy().reshape(2200, T_FRAMES, 39), n_states=16)
    del f_all
    pack = DiagModelPack.from_params(*models, device=dev)
    pipe = RecognizerPipeline(plan, pack, lens)
    st = _lib.current_stream()
    ms_mfcc = _ev_ms(torch, lambda: pipe.launch_mfcc(pcm, st), 5)
    ms_dec = _ev_ms(torch, lambda: pipe.launch_decode(st), 3)
    n_s = 512
    host = pcm[: n_s * N_SAMP].cpu().numpy().reshape(n_s, N_SAMP)
    cfg = dict(mo.BENCH, preemph=0.97, deltas=True)
    o_feats = np.concatenate([mo.mfcc(y, **cfg).T for y in host], axis=0)
    g_feats = pipe.feats[: n_s * T_FRAMES].cpu().numpy()
    offs = (np.arange(n_s + 1) * T_FRAMES).astype(np.int64)
    osc, obw, opath = c_oracle.decode_batch(g_feats, offs, *models, tie=1, sum_order=1)
    ok = bool(np.abs(g_feats - o_feats).max() < 1e-3)
    ok &= bool(np.array_equal(pipe.best_word[:n_s].cpu().numpy(), obw))
    ok &= bool(np.array_equal(pipe.best_score[:n_s].cpu().numpy(), osc[np.arange(n_s), obw]))
    ok &= bool(np.array_equal(pipe.path[: n_s * T_FRAMES].cpu().numpy(), opath))
    o2sc, o2bw, o2path = c_oracle.decode_batch(np.ascontiguousarray(o_feats, dtype=np.float32), offs, *models, tie=1,
                                               sum_order=1)
    e2e = end_to_end_agreement(o2bw, o2sc[np.arange(n_s), o2bw], o2path, offs, pipe.best_word[:n_s].cpu().numpy(),
                               pipe.best_score[:n_s].cpu().numpy(), pipe.path[: n_s * T_FRAMES].cpu().numpy())
    frames = n_utts * T_FRAMES
    lattices = float(pipe.pruned_views()[4].sum().item()) / n_utts if pipe.mode == "pruned" else float(W)
    return {"workload": f"configs[4], one chunk: {n_utts} x 1 s utterances -> 39-dim MFCC+d+dd (pre-emphasis 0.97) -> "
                        f"Viterbi vs {W} word models x 18 states ({pipe.mode} decoder)",
            "mfcc_ms": ms_mfcc, "decode_ms": ms_dec, "frames_per_s": frames / ((ms_mfcc + ms_dec) * 1e-3),
            "exact_lattices_per_utterance": lattices,
            "hbm_frac_mfcc": (4 * HOP + 4 * 39) * frames / (ms_mfcc * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "parity_vs_oracle_sample": ok, "end_to_end": e2e}


def extra_mfcc_reference_preset(torch, dev, n_utts=10000):
    """The reference's OWN front-end configuration in batch (mfcc_extract.py:12-23 with librosa's defaults: 22 050 Hz,
    n_fft 2048, Hamming 661 / hop 220, 128 Slaney mels, 13 coefficients — what train.py / eval.py users hit): n_utts
    x 1 s through sapr_mfcc_batch; 4 * 220 + 4 * 13 = 932 algorithmic bytes per frame; parity = a 16-utterance sample
    against the oracle (1e-3)."""
    from oracle import mfcc_oracle as mo
    from sapr_amd.frontend import REFERENCE, MfccPlan
    sr, n_samp = 22050, 22050
    pcm = synth_pcm(torch, n_utts, seed=99, device=dev, SR=sr, N_SAMP=n_samp)
    lens = np.full(n_utts, n_samp, dtype=np.int64)
    plan = MfccPlan(**REFERENCE, max_frames=1 + n_samp // 220)
    feats, frames = plan(pcm, lens)
    ms = _ev_ms(torch, lambda: plan(pcm, lens), 5)
    n_s, T = 16, int(frames[0])
    host = pcm[: n_s * n_samp].cpu().numpy().reshape(n_s, n_samp)
    want = np.concatenate([mo.mfcc(y, **mo.REFERENCE).T for y in host], axis=0)
    err = float(np.abs(feats[: n_s * T].cpu().numpy() - want).max())
    total = int(frames.sum())
    return {"workload": f"{n_utts} x 1 s @22 050 Hz -> 13 MFCC, n_fft 2048 / win 661 / hop 220 / 128 mels (the reference's "
                        f"preset, mfcc_extract.py:12-23), {T} frames per utterance, "
                        f"{'two-pass' if plan.two_pass else 'fused'} workgroup-tile core",
            "ms": ms, "frames_per_s": total / (ms * 1e-3), "bytes_per_frame": 4 * 220 + 4 * 13,
            "hbm_frac": (4 * 220 + 4 * 13) * total / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "max_abs_diff_vs_oracle_sample": err, "parity_vs_oracle_sample": bool(err < 1e-3)}


def extra_decode_sensitivity(torch, dev, feats, n_utts, models):
    """How much of the headline's decode time is the data: the pruned decoder evaluates the exact lattice only for
    words whose bounding interval overlaps the best one, so its time follows the vocabulary.  Three measurements on
    the benchmark's own features: the pruned decoder as timed, the all-vocabulary evaluation (`--decode full`: every
    word's exact score + back-trace, what W calls of GaussianHMM.decode compute, decoder.py:42-47), and the pruned
    decoder's WORST case — eleven copies of one word model: every score ties, nothing may be dropped, all W exact
    lattices per utterance on top of the bounding pass."""
    from sapr_amd import _lib
    from sapr_amd.trellis import DiagModelPack, FeatureBatch, PrunedDecoder, viterbi_decode
    batch = FeatureBatch.from_packed(feats, np.full(n_utts, T_FRAMES))
    st = _lib.current_stream()
    out = {}

    def pruned(pack):
        dec = PrunedDecoder(batch.n_utts, batch.max_T, batch.total_frames, pack, dev)
        run = lambda: dec.launch(batch.feats, batch.offsets, batch.order, _lib.TIE_HIGH, _lib.SUM_TVIEW, st)  # noqa: E731
        ms = _ev_ms(torch, run, 5)
        return ms, float(dec.views()[4].sum().item()) / n_utts, dec
    pack = DiagModelPack.from_params(*models, device=dev)
    ms, lat, dec = pruned(pack)
    out["pruned_ms"], out["pruned_exact_lattices_per_utterance"] = ms, lat
    full_ms = _ev_ms(torch, lambda: viterbi_decode(batch, pack), 3)
    full = viterbi_decode(batch, pack)
    out["all_vocabulary_ms"] = full_ms
    out["pruned_equals_all_vocabulary"] = bool(torch.equal(dec.best_word, full.best_word)
                                               and torch.equal(dec.best_score, full.best_score)
                                               and torch.equal(dec.path, full.path))
    del dec, full
    same = tuple(np.repeat(m[:1], W, axis=0) for m in models)
    ms, lat, dec = pruned(DiagModelPack.from_params(*same, device=dev))
    out["worst_case_identical_models"] = {"pruned_ms": ms, "exact_lattices_per_utterance": lat,
                                          "best_word_is_first_model": bool((dec.best_word == 0).all().item())}
    out["workload"] = (f"{n_utts} x {T_FRAMES} frames x {D} MFCC, {W} word models x {N_STATES + 2} states; the benchmark's "
                       "models come from interleaved draws of ONE synthetic distribution (near-ties: close to the "
                       "worst case for pruning already)")
    return out


def extra_stream_1m(torch, dev, pcm, n_utts, n_chunks=10):
    """configs[4] end to end on one GPU: n_chunks x n_utts utterances as int16 PCM in pinned host memory,
    uploaded chunk by chunk on a second stream while the previous chunk computes (sapr_amd/stream.py).  The
    chunks re-use ONE synthetic chunk (3.2 GB of host memory instead of 32 GB); each is uploaded and processed in
    full.  parity = tests/test_stream_gpu.py (driver == direct pipeline, bit for bit)."""
    from sapr_amd.frontend import BENCH39, MfccPlan
    from sapr_amd.stream import StreamingRecognizer
    from sapr_amd.trellis import DiagModelPack
    lens = np.full(n_utts, N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH39, max_frames=T_FRAMES)
    f, _ = plan(pcm[: 2200 * N_SAMP], lens[:2200])
    pack = DiagModelPack.from_params(*build_models(f.cpu().numpy().reshape(2200, T_FRAMES, 39), n_states=16), device=dev)
    pcm16 = torch.clamp((pcm * 32768.0).round(), -32768, 32767).to(torch.int16).cpu().pin_memory()
    rec = StreamingRecognizer(plan, pack, device=dev)
    rec.run([(pcm16, lens)] * 2, keep_results=False)
    _, rep = rec.run([(pcm16, lens)] * n_chunks, keep_results=False)
    return {"workload": f"configs[4]: {n_chunks} chunks x {n_utts} utterances x 1 s, int16 PCM from pinned host memory "
                        f"(H2D overlapped with the previous chunk's kernels) -> 39-dim MFCC+d+dd -> Viterbi vs {W} word "
                        "models x 18 states",
            "utterances": rep.n_utts, "frames": rep.frames, "wall_s": rep.wall_s,
            "frames_per_s_pcie_inclusive": rep.frames_per_s_pcie_inclusive,
            "frames_per_s_kernels_only": rep.frames_per_s_kernels_only,
            "kernel_ms_per_chunk": float(np.mean(rep.chunk_kernel_ms))}


# ------------------------------------------------------------------------------------------- modes
def run_em_mode(args, torch, dist, dev, rank, world):
    """configs[3]: utterance-sharded Baum-Welch.  Per iteration: E-step kernels over this rank's shard,
    ONE all-reduce(SUM) of stats[W][width] float64, the same M-step on every rank."""
    from sapr_amd import dist as sdist
    from sapr_amd.frontend import BENCH, MfccPlan
    from sapr_amd.hmmlearn_hmm import m_step_batch
    from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
    n_words, S = 10, N_STATES + 2
    n_utts = args.utts
    pcm = synth_pcm(torch, n_utts, seed=4321 + rank, device=dev)
    lens = np.full(n_utts, N_SAMP, dtype=np.int64)
    feats, _ = MfccPlan(**BENCH, max_frames=T_FRAMES)(pcm, lens)
    del pcm
    models = build_models(feats[: 2000 * T_FRAMES].cpu().numpy().reshape(2000, T_FRAMES, D), n_words=n_words)
    if dist is not None:   # every rank starts from rank 0's models
        flat = torch.from_numpy(np.concatenate([m.reshape(-1) for m in models])).to(dev)
        dist.broadcast(flat, src=0)
        h, o, out = flat.cpu().numpy(), 0, []
        for m in models:
            out.append(h[o:o + m.size].reshape(m.shape).copy())
            o += m.size
        models = tuple(out)
    sp, A, mu, cv = (m.copy() for m in models)
    batch = FeatureBatch.from_packed(feats, np.full(n_utts, T_FRAMES))
    es = EStep(batch, np.arange(n_utts) % n_words, n_words, S)
    t_e, t_ar, t_m = [], [], []

    def iteration(timed):
        nonlocal sp, A, mu, cv
        t0 = time.perf_counter()
        pack = DiagModelPack.from_params(sp, A, mu, cv, device=dev, exact_only=True)
        stats = es.run(pack)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sdist.allreduce_sum_(stats)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host = stats.cpu().numpy()
        sp, A, mu, cv, lps = m_step_batch(host, es.S, es.D, sp, A, mu, cv)
        t3 = time.perf_counter()
        if timed:
            t_e.append(t1 - t0)
            t_ar.append(t2 - t1)
            t_m.append(t3 - t2)
        return float(lps.sum())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        iteration(False)
    barrier()
    t0 = time.perf_counter()
    lls = [iteration(True) for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    if rank == 0:
        frames = n_utts * T_FRAMES * world
        line = {"metric": "frames/sec Baum-Welch EM iteration (13-MFCC, 8-state HMM, 10 words; E-step + suff-stat "
                          "all-reduce + M-step)",
                "value": frames * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "configs[3]: Baum-Welch EM, hmmlearn-compatible E-step over an utterance shard "
                                       "per GPU + RCCL all-reduce of the sufficient statistics + M-step on every rank",
                           "utterances_per_gpu": n_utts, "frames_per_utterance": T_FRAMES, "word_models": n_words,
                           "states": S, "parallelism": f"utterance-shard x{world}",
                           "allreduce_doubles": int(es.stats.numel())},
                "phase_ms": {"estep_incl_model_upload": 1e3 * float(np.mean(t_e)),
                             "allreduce": 1e3 * float(np.mean(t_ar)), "mstep_incl_stats_d2h": 1e3 * float(np.mean(t_m))},
                "loglik_monotone": bool(all(b >= a - 1e-6 * abs(a) for a, b in zip(lls, lls[1:]))),
                # 52 algorithmic bytes per frame and iteration (13 float32 features read); the launch sequence
                # moves 2.55 GB per 100 000 utterances (profiles/r04_estep_rocprofv3_summary.txt): the share lattice once
                # each way, the slot-major features twice
                "roofline": {"bound": "fp64 valu (forward pass) / hbm (smoothing + sums): not the algorithmic bytes",
                             "kernel": "fb_forward_kernel<13,10,QEMIT> + fb_smooth_obs_kernel<13,10> + fb_reduce_kernel "
                                       "(incl. model upload and pack: phase_ms.estep_incl_model_upload)",
                             "achieved": 4 * D * n_utts * T_FRAMES / (float(np.mean(t_e))) / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": 4 * D * n_utts * T_FRAMES / float(np.mean(t_e)) / 1e9 / HBM_PEAK_GBS,
                             "traffic": 2.55e9 * n_utts / 100000 if D == 13 else None,
                             "traffic_source": "profiles/r04_estep_rocprofv3_summary.txt scaled to this batch"},
                "cpu_baseline": None}
        print(json.dumps(line), flush=True)


def run_stream_mode(args, torch, dist, dev, rank, world):
    """configs[4] as STRONG scaling: a corpus of --total-utts utterances (int16 PCM in pinned host memory, chunks of
    --utts utterances) is cut into contiguous runs of chunks, one run per rank (sapr_amd.stream.shard_chunks; the
    loops replaced are mfcc_extract.py:35-49 and decoder.py:58-70); every rank streams its run through
    StreamingRecognizer (H2D of chunk k+1 under the kernels of chunk k), no collective on the data path.  One step
    = one pass over the whole corpus; value = corpus frames x steps / max-over-ranks wall time (PCIe-inclusive, the
    honest figure for a corpus that does not fit in HBM), kernels-only rate reported next to it.  Every rank uploads
    and processes its chunks in full; the chunks re-use ONE synthetic chunk of host memory per rank."""
    from sapr_amd.frontend import BENCH39, MfccPlan
    from sapr_amd.stream import StreamingRecognizer, shard_chunks
    from sapr_amd.trellis import DiagModelPack
    chunk = args.utts
    n_chunks = max(1, (args.total_utts + chunk - 1) // chunk)
    mine = list(shard_chunks(n_chunks, rank, world))
    pcm = synth_pcm(torch, chunk, seed=777 + rank, device=dev)
    lens = np.full(chunk, N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH39, max_frames=T_FRAMES)
    n_model = min(chunk, 2200)
    f, _ = plan(pcm[: n_model * N_SAMP], lens[:n_model])
    models = build_models(f.cpu().numpy().reshape(n_model, T_FRAMES, 39), n_states=16)
    if dist is not None:   # every rank decodes against rank 0's models
        flat = torch.from_numpy(np.concatenate([m.reshape(-1) for m in models])).to(dev)
        dist.broadcast(flat, src=0)
        h, o, out = flat.cpu().numpy(), 0, []
        for m in models:
            out.append(h[o:o + m.size].reshape(m.shape).copy())
            o += m.size
        models = tuple(out)
    pack = DiagModelPack.from_params(*models, device=dev)
    pcm16 = torch.clamp((pcm * 32768.0).round(), -32768, 32767).to(torch.int16).cpu().pin_memory()
    del pcm, f
    rec = StreamingRecognizer(plan, pack, device=dev)
    work = [(pcm16, lens)] * len(mine)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        rec.run(work[:2] if len(work) > 2 else work, keep_results=False)
    barrier()
    t0 = time.perf_counter()
    kernel_s = 0.0
    words = None
    for _ in range(args.steps):
        if work:
            def keep(k, bw, bs, path):
                nonlocal words
                words = int(bw[0])
            _, rep = rec.run(work, on_result=keep, keep_results=False)
            kernel_s += rep.kernel_s
    barrier()
    elapsed = time.perf_counter() - t0
    red = torch.tensor([elapsed, kernel_s], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    elapsed, kernel_max = float(red[0].item()), float(red[1].item())
    if rank == 0:
        frames = n_chunks * chunk * T_FRAMES
        line = {"metric": "frames/sec full pipeline (39-dim MFCC+d+dd, 16-state HMMs, 1M utterances), strong scaling",
                "value": frames * args.steps / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f32 (MFCC) + f64 (trellis)", "data": "synthetic",
                "config": {"workload": "configs[4]: int16 PCM corpus in pinned host memory -> 39-dim MFCC+d+dd "
                                       "(pre-emphasis 0.97) -> pruned Viterbi vs 11 word models x 18 states, chunked, "
                                       "H2D overlapped with compute",
                           "total_utterances": n_chunks * chunk, "chunk_utterances": chunk, "chunks": n_chunks,
                           "chunks_per_rank": [len(list(shard_chunks(n_chunks, r, world))) for r in range(world)],
                           "frames_per_utterance": T_FRAMES, "word_models": W, "states": 18,
                           "parallelism": f"chunk-shard x{world} (no data-path collective)"},
                "pcie_inclusive": True,
                "frames_per_s_kernels_only": frames * args.steps / kernel_max if kernel_max else None,
                "roofline": {"bound": "pcie (host-resident corpus): 2 B/sample int16 upload",
                             "kernel": "StreamingRecognizer chunk (pcm16_to_f32 + mfcc_wave + finish + pruned decode)",
                             "achieved": 2 * N_SAMP * n_chunks * chunk * args.steps / elapsed / 1e9,
                             "peak": 63.0 * world, "unit": "GB/s (PCIe Gen5 x16 per GPU)",
                             "frac": 2 * N_SAMP * n_chunks * chunk * args.steps / elapsed / 1e9 / (63.0 * world),
                             "traffic": None},
                "cpu_baseline": None}
        print(json.dumps(line), flush=True)


def _launch_ranks(n):
    """Start `n` ranks of this very command under torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1 at a free port) as a child process that inherits stdout / stderr; returns its exit code."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this pool's host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # 0.26 s timed at 1 GPU: 20 steps (0.1 s) sat inside the box-to-box noise
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=100000, help="utterances per GPU per step")
    ap.add_argument("--cpu-utts", type=int, default=30000, help="utterances of the single-core CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the `extra` block (other BASELINE configs)")
    ap.add_argument("--mode", choices=["pipeline", "em", "stream"], default="pipeline")
    ap.add_argument("--total-utts", type=int, default=1000000, help="--mode stream: utterances of the whole corpus")
    ap.add_argument("--decode", choices=["pruned", "full"], default="pruned")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as a CHILD process (this one has
        # not touched the GPU and never will: no exec from a GPU process), forward its output and exit with its code
        raise SystemExit(_launch_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    # SAPR_BENCH_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks
    # (ranks then share cards); the driver's runs use the default: RCCL, one rank per GPU
    backend = os.environ.get("SAPR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("SAPR_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.mode in ("em", "stream"):
        (run_em_mode if args.mode == "em" else run_stream_mode)(args, torch, dist, dev, rank, world)
        if dist is not None:
            dist.destroy_process_group()
        return

    from sapr_amd import _lib
    from sapr_amd.frontend import BENCH, MfccPlan
    from sapr_amd.pipeline import RecognizerPipeline
    from sapr_amd.trellis import DiagModelPack

    n_utts = args.utts
    pcm = synth_pcm(torch, n_utts, seed=1234 + rank, device=dev)
    lens = np.full(n_utts, N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH, max_frames=T_FRAMES)

    # word models from the data itself: rank 0 segments the first utterances of its (full-size,
    # untimed) front-end pass and broadcasts the parameters — every launch the profiler sees has
    # the benchmark's size
    n_model = min(n_utts, 2200)
    f_all, _ = plan(pcm, lens)
    models = build_models(f_all[: n_model * T_FRAMES].cpu().numpy().reshape(n_model, T_FRAMES, D))
    del f_all
    if dist is not None:
        packed = torch.from_numpy(np.concatenate([m.reshape(-1) for m in models])).to(dev)
        dist.broadcast(packed, src=0)
        flat, o, shapes = packed.cpu().numpy(), 0, [m.shape for m in models]
        models = []
        for sh in shapes:
            n = int(np.prod(sh))
            models.append(flat[o:o + n].reshape(sh).copy())
            o += n
        models = tuple(models)
    pack = DiagModelPack.from_params(*models, device=dev)
    assert pack.topology == _lib.TOPO_BIDIAG
    pipe = RecognizerPipeline(plan, pack, lens, mode=args.decode)
    stream = _lib.current_stream()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.run(pcm)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        pipe.launch_mfcc(pcm, stream)
        ev[k][1].record()
        pipe.launch_decode(stream)
        ev[k][2].record()
    barrier()
    elapsed = time.perf_counter() - t0

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    pipe_frames = pipe.total_frames
    frames_per_step = pipe.total_frames * world
    value = frames_per_step * args.steps / elapsed

    if rank == 0:
        kt = {name: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(args.steps)]))
              for i, name in enumerate(("mfcc", "decode"))}  # ms per launch (sequence)
        dom = max(kt, key=kt.get)
        alg_bytes = BYTES_PER_FRAME[dom] * pipe.total_frames
        achieved = alg_bytes / (kt[dom] * 1e-3) / 1e9
        # HBM traffic and VALU instruction counts are properties of the kernels, not of this run: they come from the
        # committed rocprofv3 counter passes of THIS command (profiles/pmc_traffic.json names its source); the time is
        # measured here
        traffic = traffic_src = valu_insts = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                from sapr_amd.build import source_hash
                ent = json.load(open(tpath)).get(dom, {})
                if ent.get("utts") == n_utts and ent.get("source_hash") == source_hash(dom):
                    traffic, traffic_src = ent.get("hbm_bytes_per_launch"), ent.get("source")
                    valu_insts = ent.get("valu_insts_per_launch")
                elif ent:   # the kernels' sources changed after the counter passes were taken (or another batch size)
                    traffic_src = "stale: profiles/pmc_traffic.json was taken from other sources / sizes; re-profile"
            except Exception:
                traffic = None
        mfcc_tflops = MFCC_FLOP_PER_FRAME * pipe.total_frames / (kt["mfcc"] * 1e-3) / 1e12
        kernels = {"mfcc": "mfcc_wave_kernel<false,1,15,7> + mfcc_wave_finish_kernel (one launch sequence)"
                   if plan.two_pass else "mfcc_kernel<16,...>",
                   "decode": "viterbi_bound_lds_kernel<13,10,11,4,3> + viterbi_select_kernel + viterbi_bidiag_kernel<13,10,"
                             "...,CAND> + viterbi_backtrace_pruned_kernel (pruned decoder, one launch sequence)"
                   if pipe.mode == "pruned" else "viterbi_bidiag_kernel<13,10,...> + viterbi_backtrace_kernel"}
        roofline = {"bound": "valu issue (not hbm)" if dom == "mfcc" else "mfma + fp32/fp64 valu issue (not hbm)",
                    "kernel": kernels[dom],
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    # the MFCC kernels are neither HBM- nor MFMA-bound: the spectral kernel is limited by the rate at
                    # which a SIMD issues vector instructions; these fractions say how far that is
                    "flops_achieved_TFLOPs": mfcc_tflops, "flops_peak_TFLOPs": FP32_VALU_PEAK_TFLOPS,
                    "flops_frac": mfcc_tflops / FP32_VALU_PEAK_TFLOPS,
                    "flops_note": f"{MFCC_FLOP_PER_FRAME:.0f} flop/frame (SURVEY 8d) x frames / mfcc launch-sequence time "
                                  "vs 157.3 TFLOP/s vector/matrix float32",
                    "valu_issue_frac": (valu_insts / (kt[dom] * 1e-3) / (N_SIMDS * CLOCK_HZ * VALU_SAT_PER_CYCLE)
                                        if valu_insts else None),
                    "valu_issue_note": "SQ_INSTS_VALU per launch sequence (committed profile) / measured time / "
                                       f"({N_SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz x {VALU_SAT_PER_CYCLE} "
                                       "wave-instructions per nominal cycle that a pure-VALU loop of the kernel's own "
                                       "instruction mix sustains at 4 wavefronts per SIMD: scripts/ubench/mix_rate)",
                    "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
                    "all_kernels_GBps": {k: BYTES_PER_FRAME[k] * pipe.total_frames / (v * 1e-3) / 1e9
                                         for k, v in kt.items()}}
        if pipe.mode == "pruned":
            cc = pipe.pruned_views()[4]
            roofline["exact_lattices_per_utterance"] = float(cc.sum().item()) / n_utts
        cpu = cpu_all = extra = None
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = min(args.cpu_utts, n_utts)
            pcm_host = pcm[: n_cpu * N_SAMP].cpu().numpy().reshape(n_cpu, N_SAMP)
            cpu, (o_feats, o_offs, o_bw, o_path, o_sc) = cpu_baseline(pcm_host, models)
            # checker use of the oracle (never the thing measured): decode the ORACLE's features with
            # the HIP decoder the bench timed and demand identical words / scores / paths on the sample
            from sapr_amd.trellis import FeatureBatch, viterbi_decode, viterbi_decode_best
            fb = FeatureBatch.from_packed(torch.from_numpy(o_feats).to(dev), np.diff(o_offs))
            if pipe.mode == "pruned":
                g_bw, g_bs, g_path = viterbi_decode_best(fb, pack)
            else:
                r = viterbi_decode(fb, pack)
                g_bw, g_bs, g_path = r.best_word, r.best_score, r.path
            torch.cuda.synchronize()
            cpu["viterbi_paths_identical_on_sample"] = bool(
                np.array_equal(g_path.cpu().numpy(), o_path) and np.array_equal(g_bw.cpu().numpy(), o_bw)
                and np.array_equal(g_bs.cpu().numpy(), o_sc[np.arange(len(o_bw)), o_bw]))
            gpu_f = pipe.feats[: int(o_offs[-1])].cpu().numpy()
            cpu["mfcc_max_abs_diff_on_sample"] = float(np.abs(gpu_f - o_feats).max())
            # and END TO END: oracle MFCC -> oracle decode against HIP MFCC -> HIP decode (the timed pipeline's own
            # outputs) on the same utterances.  The two feature sets differ in the last float32 bits, so this is an
            # agreement rate, not a bit-for-bit gate (decoder.py:35-49 returns word, score and path)
            cpu["end_to_end"] = end_to_end_agreement(
                o_bw, o_sc[np.arange(len(o_bw)), o_bw], o_path, o_offs, pipe.best_word[:n_cpu].cpu().numpy(),
                pipe.best_score[:n_cpu].cpu().numpy(), pipe.path[: int(o_offs[-1])].cpu().numpy())
            cores = usable_cores()
            n_all = min(n_utts, 4000 * cores)
            host_all = pcm[: n_all * N_SAMP].cpu().numpy().reshape(n_all, N_SAMP)
            try:
                cpu_all = cpu_baseline_all_cores(host_all, models, cores)
            except Exception as e:  # a box that forbids worker processes must not lose the bench line
                cpu_all = {"value": None, "error": repr(e), "cores": cores}
        if world == 1 and not args.no_extras:
            extra = {}
            feats13 = pipe.feats.clone()
            del pipe
            torch.cuda.empty_cache()
            for name, fn in (("decode_sensitivity", lambda: extra_decode_sensitivity(torch, dev, feats13, n_utts, models)),
                             ("em_hmmlearn_compat", lambda: extra_em_hmmlearn(torch, dev, feats13, n_utts)),
                             ("em_custom_hmm", lambda: extra_em_custom(torch, dev, feats13, n_utts)),
                             ("decode_custom_hmm", lambda: extra_decode_custom(torch, dev, feats13, n_utts)),
                             ("mfcc_reference_preset", lambda: extra_mfcc_reference_preset(torch, dev)),
                             ("pipeline_39dim_18state", lambda: extra_pipeline39(torch, dev, pcm, n_utts)),
                             ("stream_1M_39dim_18state", lambda: extra_stream_1m(torch, dev, pcm, n_utts))):
                try:
                    extra[name] = fn()
                except Exception as e:
                    extra[name] = {"error": repr(e)}
                torch.cuda.empty_cache()
        both = None
        if extra and isinstance(extra.get("decode_sensitivity"), dict) and "all_vocabulary_ms" in extra["decode_sensitivity"]:
            # the pruned decoder's time follows the vocabulary (bit-identical outputs either way): the same step with
            # EVERY word scored exactly — what W calls of GaussianHMM.decode compute, decoder.py:42-47 — next to it
            ms_all = kt["mfcc"] + extra["decode_sensitivity"]["all_vocabulary_ms"]
            both = {"ms_per_step": ms_all, "value": pipe_frames / (ms_all * 1e-3), "unit": "frames/s",
                    "what": "MFCC launch sequence as timed + all-vocabulary Viterbi and back-trace (HIP events, 1 GPU)"}
        line = {"metric": "frames/sec MFCC+Viterbi (16kHz, 13-MFCC, 8-state HMM)", "value": value,
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32 (MFCC) + f64 (trellis)", "data": "synthetic",
                "config": {"workload": "configs[1]+[2]: 16 kHz 1 s utterances -> 13 MFCC (25 ms/10 ms, "
                                       "n_fft 512, 40 mels) -> Viterbi vs 11 word models x 8 emitting "
                                       "states (decoder.py API), features materialised in HBM",
                           "utterances_per_gpu": n_utts, "frames_per_utterance": T_FRAMES,
                           "word_models": W, "states": N_STATES + 2, "parallelism": f"utterance-shard x{world}",
                           "decoder": args.decode},
                "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
                "all_vocabulary_decode_equivalent": both, "extra": extra}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
