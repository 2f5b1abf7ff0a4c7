"""BASELINE.json full size (100 000 utterances x 1 s, the bench workload) through size-independent
properties, plus an oracle check on a random sample — the oracle cannot walk 10 M frames in a test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_full_size_pipeline_properties():
    import torch
    import bench
    from oracle import c_oracle, mfcc_oracle as mo
    from sapr_amd.frontend import BENCH, MfccPlan
    from sapr_amd.pipeline import RecognizerPipeline
    from sapr_amd.trellis import DiagModelPack

    dev = torch.device("cuda", 0)
    N, T, D, W = 100_000, bench.T_FRAMES, bench.D, bench.W
    pcm = bench.synth_pcm(torch, N, seed=99, device=dev)
    # property set-up: the last 1000 utterances are copies of the first 1000
    pcm.view(N, bench.N_SAMP)[N - 1000:] = pcm.view(N, bench.N_SAMP)[:1000]
    lens = np.full(N, bench.N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH, max_frames=T)
    f_all, _ = plan(pcm, lens)
    models = bench.build_models(f_all[: 2200 * T].cpu().numpy().reshape(2200, T, D))
    del f_all
    pack = DiagModelPack.from_params(*models, device=dev)
    pipe = RecognizerPipeline(plan, pack, lens, mode="full")
    bw, bs, path = (x.clone() for x in pipe.run(pcm))
    scores = pipe.scores.clone()
    feats = pipe.feats.clone()
    torch.cuda.synchronize()

    # 0. the pruned decoder (what bench.py times) returns the all-vocabulary evaluation's bits at full size,
    #    its intervals contain every exact score, and it really prunes
    fast = RecognizerPipeline(plan, pack, lens)
    assert fast.mode == "pruned"
    pbw, pbs, ppath = fast.run(pcm)
    torch.cuda.synchronize()
    assert torch.equal(fast.feats, feats)
    assert torch.equal(pbw, bw) and torch.equal(pbs, bs) and torch.equal(ppath, path)
    asc, aeps, exs, cslot, ccnt = fast.pruned_views()
    assert bool(((asc - scores).abs() <= aeps).all())
    kept = cslot >= 0
    assert torch.equal(exs[kept], scores[kept])
    assert int(ccnt.sum()) == int(kept.sum()) and int(kept.sum()) < 2 * N    # ~1 word per utterance survives
    assert bool(kept[torch.arange(N, device=dev), bw.long()].all())           # the winner is never dropped
    pbw2, pbs2, ppath2 = (x.clone() for x in fast.run(pcm))                   # list order may differ, results not
    torch.cuda.synchronize()
    assert torch.equal(pbw2, bw) and torch.equal(pbs2, bs) and torch.equal(ppath2, path)
    del fast

    # 1. run-to-run determinism, bit for bit (features, scores, words, paths)
    bw2, bs2, path2 = pipe.run(pcm)
    torch.cuda.synchronize()
    assert torch.equal(feats, pipe.feats) and torch.equal(scores, pipe.scores)
    assert torch.equal(bw, bw2) and torch.equal(bs, bs2) and torch.equal(path, path2)

    # 2. identical utterances anywhere in the batch get identical results (no cross-utterance leakage)
    f3 = feats.view(N, T, D)
    assert torch.equal(f3[:1000], f3[N - 1000:])
    assert torch.equal(scores[:1000], scores[N - 1000:])
    assert torch.equal(bw[:1000], bw[N - 1000:])
    assert torch.equal(path.view(N, T)[:1000], path.view(N, T)[N - 1000:])

    # 3. structural invariants of every decoded path / score
    assert bool(torch.isfinite(feats).all()) and bool(torch.isfinite(scores).all())
    assert torch.equal(bs, scores.max(dim=1).values)
    assert torch.equal(bw.long(), scores.argmax(dim=1))        # first maximum = decoder.py's strict '>'
    p = path.view(N, T)
    assert int(p.min()) >= 0 and int(p.max()) < pack.S_model
    assert bool((p[:, 0] == 0).all())                           # startprob = e_0
    step = p[:, 1:] - p[:, :-1]
    assert int(step.min()) >= 0 and int(step.max()) <= 1        # left-to-right, no skips

    # 4. oracle on a random sample of the full-size batch: features to tolerance, decode bit for bit
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(N, 192, replace=False))
    host = pcm.view(N, bench.N_SAMP)[torch.from_numpy(idx).to(dev)].cpu().numpy()
    o_feats = np.stack([mo.mfcc(host[i], **mo.BENCH).T for i in range(len(idx))])
    g_feats = f3[torch.from_numpy(idx).to(dev)].cpu().numpy()
    assert np.abs(g_feats - o_feats).max() < 1e-3               # +-600-range coefficients, float32 chain
    sp, A, mu, cv = models
    packed = np.ascontiguousarray(g_feats.reshape(-1, D))
    offs = (np.arange(len(idx) + 1) * T).astype(np.int64)
    c_oracle.load()
    sc, obw, opath = c_oracle.decode_batch(packed, offs, sp, A, mu, cv, tie=1, sum_order=1)
    sel = torch.from_numpy(idx).to(dev)
    np.testing.assert_array_equal(scores[sel].cpu().numpy(), sc)
    np.testing.assert_array_equal(bw[sel].cpu().numpy(), obw)
    np.testing.assert_array_equal(p[sel].cpu().numpy().reshape(-1), opath)


def test_full_size_estep_properties():
    """BASELINE config 4 size: 100 000 utterances, 10 words.  Sufficient statistics are additive over any
    partition of the utterances (what the cross-rank all-reduce relies on), posteriors sum to the number
    of frames, and a second run is bit-identical."""
    import torch
    from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch, split_stats
    from tests._synth import trained_like_models
    dev = torch.device("cuda", 0)
    N, T, D, W, S = 100_000, 101, 13, 10, 10
    g = torch.Generator(device=dev).manual_seed(5)
    feats = torch.randn(N * T, D, device=dev, generator=g) * 20
    feats[:, 0] -= 300
    sp, A, mu, cv = trained_like_models(W, 8, D, seed=3)
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    utt_model = np.arange(N) % W
    lengths = np.full(N, T)

    def stats_of(lo, hi):
        batch = FeatureBatch.from_packed(feats[lo * T:hi * T].contiguous(), lengths[lo:hi])
        es = EStep(batch, utt_model[lo:hi], W, S)
        return es.run(pack).cpu().numpy().copy(), es

    full, es = stats_of(0, N)
    again = es.run(pack).cpu().numpy()
    np.testing.assert_array_equal(full, again)
    parts = stats_of(0, 37_001)[0] + stats_of(37_001, N)[0]
    np.testing.assert_allclose(parts, full, rtol=1e-10, atol=1e-6)
    for w in range(W):
        st = split_stats(full[w], S, D)
        n_w = int((utt_model == w).sum())
        assert st["nobs"] == n_w and np.isfinite(st["logprob"])
        np.testing.assert_allclose(st["post"].sum(), n_w * T, rtol=1e-10)        # sum_t sum_s gamma = frames
        np.testing.assert_allclose(st["start"].sum(), n_w, rtol=1e-10)           # gamma_0 rows sum to 1
        np.testing.assert_allclose(st["trans"].sum(), n_w * (T - 1), rtol=1e-10)  # xi sums to T-1 per sequence
        # obs = sum gamma x : checksum against the plain column sums of this word's frames
        col = feats.view(N, T, D)[torch.from_numpy(np.nonzero(utt_model == w)[0]).to(dev)].double().sum(dim=(0, 1))
        np.testing.assert_allclose(st["obs"].sum(axis=0), col.cpu().numpy(), rtol=1e-9)
