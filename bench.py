#!/usr/bin/env python3
"""Headline benchmark: frames/sec through MFCC + all-vocabulary Viterbi (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of synthetic audio already resident in HBM:
  PCM (16 kHz, 1 s / utterance) --sapr_mfcc_batch--> 13 MFCC x 101 frames
      --sapr_viterbi_diag_scores (W=11 word models, 8 emitting states + entry/exit)-->
      --sapr_viterbi_backtrace--> arg-max word + state path per utterance
i.e. BASELINE configs[1] (batched MFCC) feeding configs[2] (decoder.py-API Viterbi), which is
the combination the metric "frames/sec MFCC+Viterbi (16kHz, 13-MFCC, 8-state HMM)" is quoted on.
A frame is counted once however many word models score it (decoder.py:42 semantics).

Multi-GPU: utterances shard across ranks with NO data-path collective (weak scaling: every rank
owns --utts utterances); value = frames of all ranks / max-over-ranks time.

The JSON line also carries `roofline` (dominant kernel, algorithmic bytes / HIP-event time vs the
8 TB/s HBM peak) and `cpu_baseline` (the oracle timed on this host's cores, rank 0, N=1 only).
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
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md, chip table (spec)
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X datasheet vector FP64 (not in the local guide)
BYTES_PER_FRAME = {"mfcc": 4 * HOP + 4 * D,   # fp32 PCM hop in + 13 fp32 out        
                   "viterbi": 4 * D + 4,      # fp32 features in + int32 state out   
                   "backtrace": 4}


def synth_pcm(torch, n_utts, seed, device):
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


def build_models(feats_3d):
    """W word models by uniform segmentation of a few hundred utterances (plumbing, numpy):
    state s of word w = mean / variance of the frames of segment s over utterances u % W == w;
    bidiagonal transitions with a_ii = exp(-1/(T/N_s - 1)) (hmmlearn_hmm.py:45-78)."""
    n, T, d = feats_3d.shape
    S = N_STATES + 2
    seg = np.minimum((np.arange(T) * S) // T, S - 1)
    means = np.empty((W, S, d))
    covars = np.empty((W, S, d))
    for w in range(W):
        x = feats_3d[w::W].astype(np.float64)
        for s in range(S):
            fr = x[:, seg == s].reshape(-1, d)
            means[w, s] = fr.mean(0)
            covars[w, s] = fr.var(0) + 1.0
    aii = np.exp(-1.0 / (T / N_STATES - 1.0))
    A = np.zeros((S, S))
    A[0, 1] = 1.0
    for i in range(1, N_STATES + 1):
        A[i, i], A[i, i + 1] = aii, 1 - aii
    A[S - 1, S - 1] = 1.0
    sp = np.zeros(S)
    sp[0] = 1.0
    return np.tile(sp, (W, 1)), np.tile(A, (W, 1, 1)), means, covars


def cpu_baseline(pcm_host, models, n_utts):
    """Time the oracle (single thread) on a bounded sample of the same workload: numpy MFCC
    restatement (librosa chain) + C Viterbi restatement (hmmlearn) for all W models."""
    from oracle import c_oracle, mfcc_oracle as mo
    sp, A, mu, cv = models
    c_oracle.load()
    t0 = time.perf_counter()
    feats = [mo.mfcc(pcm_host[u], **mo.BENCH).T for u in range(n_utts)]
    t_mfcc = time.perf_counter() - t0
    packed = np.ascontiguousarray(np.concatenate(feats, axis=0), dtype=np.float32)
    offs = np.r_[0, np.cumsum([f.shape[0] for f in feats])].astype(np.int64)
    t0 = time.perf_counter()
    sc, bw, path = c_oracle.decode_batch(packed, offs, sp, A, mu, cv, tie=1, sum_order=1)
    t_vit = time.perf_counter() - t0
    frames = int(offs[-1])
    return {"value": frames / (t_mfcc + t_vit), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n_utts} utterances x 1 s ({frames} frames): numpy MFCC restatement "
                      f"{t_mfcc:.2f} s + C Viterbi restatement x{W} models {t_vit:.2f} s, 1 thread "
                      f"of {os.cpu_count()} host cpus"}, (packed, offs, bw, path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--utts", type=int, default=100000, help="utterances per GPU per step")
    ap.add_argument("--cpu-utts", type=int, default=30000, help="utterances of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # SAPR_BENCH_BACKEND=gloo rehearses the N>1 control flow on a box with fewer GPUs than ranks
    # (ranks then share cards); the driver's runs use the default: RCCL, one rank per GPU
    backend = os.environ.get("SAPR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

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
    pipe = RecognizerPipeline(plan, pack, lens)
    stream = _lib.current_stream()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pipe.run(pcm)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        pipe.launch_mfcc(pcm, stream)
        ev[k][1].record()
        pipe.launch_viterbi(stream)
        ev[k][2].record()
        pipe.launch_backtrace(stream)
        ev[k][3].record()
    barrier()
    elapsed = time.perf_counter() - t0

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    frames_per_step = pipe.total_frames * world
    value = frames_per_step * args.steps / elapsed

    if rank == 0:
        kt = {name: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(args.steps)]))
              for i, name in enumerate(("mfcc", "viterbi", "backtrace"))}  # ms per launch
        dom = max(kt, key=kt.get)
        alg_bytes = BYTES_PER_FRAME[dom] * pipe.total_frames
        achieved = alg_bytes / (kt[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(dom, {})
                if ent.get("utts") == n_utts:
                    traffic = ent.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": {"mfcc": "mfcc_kernel<16,false>",
                                               "viterbi": "viterbi_bidiag_kernel<13,10,true,true>",
                                               "backtrace": "viterbi_backtrace_kernel<true>"}[dom],
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": alg_bytes,
                    "kernel_ms": {k: round(v, 4) for k, v in kt.items()},
                    "all_kernels_GBps": {k: BYTES_PER_FRAME[k] * pipe.total_frames / (v * 1e-3) / 1e9
                                         for k, v in kt.items()}}
        # the Viterbi kernel is fp64-VALU bound, not HBM bound: per (frame, model, state, dim) it issues
        # 7 fp64 VALU instructions (sub, square, mul + 3 FMA of the exactly-rounded division, add) — report
        # the issue rate against the fp64 vector peak next to the (mandatory) HBM figure
        vit_instr = pipe.total_frames * W * (N_STATES + 2) * (D * 7 + 6)
        roofline["viterbi_fp64_valu"] = {"instr_lanes_per_s": vit_instr / (kt["viterbi"] * 1e-3),
                                         "peak_instr_lanes_per_s": FP64_VALU_PEAK_TFLOPS * 1e12 / 2,
                                         "frac": vit_instr / (kt["viterbi"] * 1e-3) / (FP64_VALU_PEAK_TFLOPS * 1e12 / 2),
                                         "note": "fp64 VALU instructions x lanes per second vs 78.6 TFLOP/s / 2 "
                                                 "(datasheet vector FP64, an FMA counted as 2 flops)"}
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = min(args.cpu_utts, n_utts)
            pcm_host = pcm[: n_cpu * N_SAMP].cpu().numpy().reshape(n_cpu, N_SAMP)
            cpu, (o_feats, o_offs, o_bw, o_path) = cpu_baseline(pcm_host, models, n_cpu)
            # checker use of the oracle (never the thing measured): decode the ORACLE's features with
            # the HIP Viterbi and demand identical words / paths on the sample
            from sapr_amd.trellis import FeatureBatch, viterbi_decode
            fb = FeatureBatch.from_packed(torch.from_numpy(o_feats).to(dev), np.diff(o_offs))
            res = viterbi_decode(fb, pack)
            torch.cuda.synchronize()
            cpu["viterbi_paths_identical_on_sample"] = bool(
                np.array_equal(res.path.cpu().numpy(), o_path) and np.array_equal(res.best_word.cpu().numpy(), o_bw))
            gpu_f = pipe.feats[: int(o_offs[-1])].cpu().numpy()
            cpu["mfcc_max_abs_diff_on_sample"] = float(np.abs(gpu_f - o_feats).max())
        line = {"metric": "frames/sec MFCC+Viterbi (16kHz, 13-MFCC, 8-state HMM)", "value": value,
                "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32 (MFCC) + f64 (trellis)", "data": "synthetic",
                "config": {"workload": "configs[1]+[2]: 16 kHz 1 s utterances -> 13 MFCC (25 ms/10 ms, "
                                       "n_fft 512, 40 mels) -> Viterbi vs 11 word models x 8 emitting "
                                       "states (decoder.py API), features materialised in HBM",
                           "utterances_per_gpu": n_utts, "frames_per_utterance": T_FRAMES,
                           "word_models": W, "states": N_STATES + 2, "parallelism": f"utterance-shard x{world}"},
                "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
