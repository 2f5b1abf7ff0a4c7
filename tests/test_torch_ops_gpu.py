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
