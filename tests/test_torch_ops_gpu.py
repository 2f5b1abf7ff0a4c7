"""torch.ops.sapr.* (torch.library registration over the C ABI) against the ctypes-based Python wrappers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_custom_ops_match_the_wrappers():
    import torch
    import bench
    import sapr_amd.torch_ops  # noqa: F401  (registers torch.ops.sapr.*)
    from sapr_amd import _lib
    from sapr_amd.frontend import BENCH, MfccPlan
    from sapr_amd.pipeline import RecognizerPipeline
    from sapr_amd.trellis import DiagModelPack, EStep, FeatureBatch
    dev = torch.device("cuda", 0)
    n = 600
    pcm = bench.synth_pcm(torch, n, seed=5, device=dev)
    lens = np.full(n, bench.N_SAMP, dtype=np.int64)
    plan = MfccPlan(**BENCH, max_frames=bench.T_FRAMES)
    feats_ref, _ = plan(pcm, lens)
    models = bench.build_models(feats_ref.cpu().numpy().reshape(n, bench.T_FRAMES, bench.D))
    pack = DiagModelPack.from_params(*models, device=dev)
    pipe = RecognizerPipeline(plan, pack, lens)
    bw, bs, path = pipe.run(pcm)
    # 16-bit PCM -> float on the device
    pcm16 = torch.clamp((pcm * 32768.0).round(), -32768, 32767).to(torch.int16)
    assert torch.equal(torch.ops.sapr.pcm16_to_f32(pcm16), pcm16.float() / 32768.0)
    # MFCC
    feats = torch.ops.sapr.mfcc_batch(pcm, pipe.sample_offsets, pipe.frame_offsets, plan.handle, plan.d_out)
    assert torch.equal(feats, feats_ref)
    # decode
    w2, s2, p2 = torch.ops.sapr.viterbi_decode_best(feats, pipe.frame_offsets, pipe.order, pack.blob, pack.W, pack.S,
                                                    pack.D, pipe.max_T, _lib.TIE_HIGH, _lib.SUM_TVIEW, pack.flags)
    assert torch.equal(w2, bw) and torch.equal(s2, bs) and torch.equal(p2, path)
    # E-step
    batch = FeatureBatch.from_packed(feats, np.full(n, bench.T_FRAMES))
    utt_model = np.arange(n) % pack.W
    es = EStep(batch, utt_model, pack.W, pack.S)
    stats_ref = es.run(pack).clone()
    lay = es.layout
    ll, stats = torch.ops.sapr.hmm_estep(feats, batch.offsets, lay.slot_utt, lay.tile_model, lay.model_tile_off,
                                         pack.blob, pack.W, pack.S, pack.D, batch.max_T, pack.topology, pack.fast_div)
    assert torch.equal(stats, stats_ref) and torch.equal(ll, es.loglik)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.sapr.pcm16_to_f32(pcm16.cpu())      # no CPU implementation


def test_custom_hmm_ops_match_the_wrappers():
    """torch.ops.sapr.custom_estep / custom_decode (the reference's from-scratch HMM, custom_hmm.py) against the
    Python mirror: decode bit for bit, the E-step's log-likelihood against HMM.baum_welch's first iteration."""
    import contextlib
    import io
    import torch
    import sapr_amd.torch_ops  # noqa: F401
    from sapr_amd.custom_hmm import HMM, decode_batch, model_arrays
    from tests._synth import VOCAB, synth_feature_set
    words = VOCAB[:3]
    by_word, flat = synth_feature_set(words, 8, D=13, seed=21)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        models = []
        for w in words:
            h = HMM(8, 13, feature_set=flat, model_name=w)
            h.baum_welch(by_word[w], max_iter=2)
            models.append(h)
        fresh = HMM(8, 13, feature_set=flat, model_name="fresh")
        arrs1 = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in model_arrays([fresh])]
        hist = fresh.baum_welch(by_word[words[0]], max_iter=1)
    utts = by_word[words[0]]
    feats = torch.from_numpy(np.ascontiguousarray(np.concatenate([f.T for f in utts]), dtype=np.float32)).cuda()
    offs = torch.from_numpy(np.r_[0, np.cumsum([f.shape[1] for f in utts])].astype(np.int64)).cuda()
    gamma, utt = torch.ops.sapr.custom_estep(feats, offs, *arrs1)
    assert gamma.shape == (feats.shape[0], 10) and utt.shape == (len(utts), 2 + 10 + 100)
    np.testing.assert_allclose(float(utt[:, 0].sum()), hist[0], rtol=1e-9)
    np.testing.assert_allclose(gamma.sum(1).cpu().numpy(), 1.0, rtol=1e-9)
    # (custom_hmm.py:434 aggregates gamma over t < T - 1)
    np.testing.assert_allclose(utt[:, 2:12].sum().item(), feats.shape[0] - len(utts), rtol=1e-9)
    arrs = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in model_arrays(models)]
    sc, pa, bw, bs, bp = torch.ops.sapr.custom_decode(feats, offs, *arrs, 8, 13)
    rsc, rpa, rbw, rbs, rbp = decode_batch(models, utts, with_best=True)
    assert np.array_equal(sc.cpu().numpy(), rsc) and np.array_equal(pa.cpu().numpy(), rpa)
    assert np.array_equal(bw.cpu().numpy(), rbw) and np.array_equal(bs.cpu().numpy(), rbs)
    assert np.array_equal(bp.cpu().numpy(), rbp)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.sapr.custom_decode(feats.cpu(), offs.cpu(), *[a.cpu() for a in arrs], 8, 13)
