"""BASELINE configs[4] (39-dim MFCC+delta+delta-delta, 16 emitting states): the chunked streaming driver and a
full-size 100 000-utterance chunk.  The 1 M-utterance corpus is ten such chunks; the oracle cannot walk it, so the
full size is checked through size-independent properties plus an oracle sample (like tests/test_fullsize_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _models39(torch, plan, pcm, lens, dev, n_model):
    import bench
    from sapr_amd.trellis import DiagModelPack
    T = bench.T_FRAMES
    f_all, _ = plan(pcm[: n_model * bench.N_SAMP], lens[:n_model])
    models = bench.build_models(f_all.cpu().numpy().reshape(n_model, T, 39), n_states=16)
    return models, DiagModelPack.from_params(*models, device=dev)


def test_streaming_driver_equals_direct_pipeline():
    """Three ragged chunks of int16 PCM through StreamingRecognizer (H2D on a second stream, double-buffered)
    give exactly what one RecognizerPipeline per chunk gives on the float conversion of the same samples."""
    import torch
    import bench
    from sapr_amd.frontend import BENCH39, MfccPlan
    from sapr_amd.pipeline import RecognizerPipeline
    from sapr_amd.stream import StreamingRecognizer, shard_chunks
    dev = torch.device("cuda", 0)
    plan = MfccPlan(**BENCH39, max_frames=bench.T_FRAMES)
    rng = np.random.default_rng(0)
    n_per = [700, 512, 700]
    pcm_f = bench.synth_pcm(torch, 2200, seed=3, device=dev)
    lens_all = np.full(2200, bench.N_SAMP, dtype=np.int64)
    models, pack = _models39(torch, plan, pcm_f, lens_all, dev, 2200)
    chunks, direct = [], []
    o = 0
    for c, n in enumerate(n_per):
        lens = rng.integers(4000, bench.N_SAMP + 1, n).astype(np.int64) if c != 1 else np.full(n, bench.N_SAMP)
        parts = []
        for u in range(n):
            parts.append(pcm_f[(o + u) * bench.N_SAMP:(o + u) * bench.N_SAMP + int(lens[u])])
        o += n
        x = torch.cat(parts)
        x16 = torch.clamp((x * 32768.0).round(), -32768, 32767).to(torch.int16)
        chunks.append((x16.cpu().pin_memory(), lens))
        pipe = RecognizerPipeline(plan, pack, lens, device=dev)
        bw, bs, path = pipe.run(x16.float() / 32768.0)
        direct.append((bw.cpu().numpy().copy(), bs.cpu().numpy().copy(), path.cpu().numpy().copy()))
    rec = StreamingRecognizer(plan, pack, device=dev)
    seen = []
    results, rep = rec.run(chunks, on_result=lambda k, bw, bs, path: seen.append(k))
    assert seen == [0, 1, 2] and len(results) == 3
    for (bw, bs, path), (dbw, dbs, dpath) in zip(results, direct):
        np.testing.assert_array_equal(bw, dbw)
        np.testing.assert_array_equal(bs, dbs)
        np.testing.assert_array_equal(path, dpath)
    assert rep.n_utts == sum(n_per) and rep.frames == sum(len(r[2]) for r in results)
    assert rep.wall_s > 0 and rep.kernel_s > 0 and len(rep.chunk_kernel_ms) == 3
    # a second pass over the same chunks reuses the pipelines and returns the same bits
    again, _ = rec.run(chunks)
    for a, b in zip(again, results):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    assert list(shard_chunks(10, 1, 4)) == [3, 4, 5] and list(shard_chunks(10, 0, 1)) == list(range(10))
    # a rank may own ONE chunk (or none): the second upload slot is never used
    one, rep1 = StreamingRecognizer(plan, pack, device=dev).run(chunks[:1])
    assert len(one) == 1 and rep1.n_utts == n_per[0] and rep1.h2d_s > 0
    for x, y in zip(one[0], results[0]):
        np.testing.assert_array_equal(x, y)
    none, rep0 = StreamingRecognizer(plan, pack, device=dev).run([])
    assert none == [] and rep0.n_utts == 0 and rep0.wall_s == 0.0


def test_full_size_39dim_18state_chunk():
    import torch
    import bench
    from oracle import c_oracle, mfcc_oracle as mo
    from sapr_amd.frontend import BENCH39, MfccPlan
    from sapr_amd.pipeline import RecognizerPipeline
    dev = torch.device("cuda", 0)
    N, T, D, W = 100_000, bench.T_FRAMES, 39, bench.W
    pcm = bench.synth_pcm(torch, N, seed=77, device=dev)
    pcm.view(N, bench.N_SAMP)[N - 500:] = pcm.view(N, bench.N_SAMP)[:500]
    lens = np.full(N, bench.N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH39, max_frames=T)
    models, pack = _models39(torch, plan, pcm, lens, dev, 2200)
    assert pack.S == 18 and pack.D == 39 and pack.prunable
    full = RecognizerPipeline(plan, pack, lens, mode="full")
    bw, bs, path = (x.clone() for x in full.run(pcm))
    scores, feats = full.scores.clone(), full.feats.clone()
    del full
    fast = RecognizerPipeline(plan, pack, lens)
    assert fast.mode == "pruned"
    pbw, pbs, ppath = fast.run(pcm)
    torch.cuda.synchronize()
    # pruned == all-vocabulary, bit for bit, at full size; intervals contain the exact scores
    assert torch.equal(pbw, bw) and torch.equal(pbs, bs) and torch.equal(ppath, path)
    asc, aeps, exs, cslot, ccnt = fast.pruned_views()
    assert bool(((asc - scores).abs() <= aeps).all())
    # determinism and no cross-utterance leakage
    f3 = feats.view(N, T, D)
    assert torch.equal(f3[:500], f3[N - 500:]) and torch.equal(bw[:500], bw[N - 500:])
    assert torch.equal(path.view(N, T)[:500], path.view(N, T)[N - 500:])
    # structural invariants
    assert bool(torch.isfinite(feats).all()) and bool(torch.isfinite(scores).all())
    assert torch.equal(bs, scores.max(dim=1).values) and torch.equal(bw.long(), scores.argmax(dim=1))
    p = path.view(N, T)
    step = p[:, 1:] - p[:, :-1]
    assert int(p.min()) >= 0 and int(p.max()) < 18 and bool((p[:, 0] == 0).all())
    assert int(step.min()) >= 0 and int(step.max()) <= 1
    # oracle on a random sample: features to the front-end's tolerance, decode bit for bit
    rng = np.random.default_rng(1)
    idx = np.sort(rng.choice(N, 64, replace=False))
    sel = torch.from_numpy(idx).to(dev)
    host = pcm.view(N, bench.N_SAMP)[sel].cpu().numpy()
    cfg = dict(mo.BENCH, preemph=0.97, deltas=True)
    o_feats = np.stack([mo.mfcc(y, **cfg).T for y in host])
    g_feats = f3[sel].cpu().numpy()
    assert np.abs(g_feats - o_feats).max() < 1e-3
    offs = (np.arange(len(idx) + 1) * T).astype(np.int64)
    sc, obw, opath = c_oracle.decode_batch(np.ascontiguousarray(g_feats.reshape(-1, D)), offs, *models, tie=1,
                                           sum_order=1)
    np.testing.assert_array_equal(scores[sel].cpu().numpy(), sc)
    np.testing.assert_array_equal(bw[sel].cpu().numpy(), obw)
    np.testing.assert_array_equal(p[sel].cpu().numpy().reshape(-1), opath)
