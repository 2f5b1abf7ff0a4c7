"""Error behaviour of the C ABI (include/sapr_hip.h): bad arguments come back as negative codes with a
message in sapr_last_error(), never as a crash or a silent no-op.  Only argument validation is
exercised — no kernel is launched with inconsistent shapes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ERR_ARG, ERR_UNSUPPORTED, ERR_WORKSPACE = -1, -2, -3


def _err(lib):
    return lib.sapr_last_error().decode()


def test_abi_version_and_device_info():
    from sapr_amd import _lib
    lib = _lib.load()
    assert lib.sapr_abi_version() >= 1
    cus, wave = C.c_int(0), C.c_int(0)
    name = C.create_string_buffer(128)
    assert lib.sapr_device_info(0, C.byref(cus), C.byref(wave), name, 128) == 0
    assert cus.value >= 64 and wave.value == 64 and b"gfx" in name.value


def test_viterbi_argument_validation():
    import torch
    from sapr_amd import _lib
    from sapr_amd.trellis import DiagModelPack, FeatureBatch
    from tests._synth import trained_like_models
    lib = _lib.load()
    sp, A, mu, cv = trained_like_models(2, 8, 13, seed=1)
    pack = DiagModelPack.from_params(sp, A, mu, cv)
    utts = [np.zeros((20, 13), np.float32) for _ in range(3)]
    batch = FeatureBatch.from_arrays(utts, layout="TD")
    need = C.c_size_t(0)
    assert lib.sapr_viterbi_workspace_bytes(3, 2, 10, 20, pack.topology, C.byref(need)) == 0 and need.value > 0
    ws = torch.empty(need.value, dtype=torch.uint8, device="cuda")
    scores = torch.empty((3, 2), dtype=torch.float64, device="cuda")
    last = torch.empty((3, 2), dtype=torch.int32, device="cuda")
    args = lambda **kw: [  # noqa: E731
        _lib.ptr(batch.feats), _lib.ptr(batch.offsets), _lib.ptr(batch.order), 3, kw.get("D", 13), 20,
        _lib.ptr(pack.blob), 2, kw.get("S", 10), kw.get("topo", pack.topology), kw.get("tie", 1), 1, 1,
        kw.get("ws", _lib.ptr(ws)), kw.get("ws_bytes", need.value), _lib.ptr(scores), _lib.ptr(last), None]
    assert lib.sapr_viterbi_diag_scores(*args()) == 0
    assert lib.sapr_viterbi_diag_scores(*args(ws_bytes=need.value - 1)) == ERR_WORKSPACE and "workspace" in _err(lib)
    assert lib.sapr_viterbi_diag_scores(*args(ws=None)) == ERR_ARG and "NULL" in _err(lib)
    assert lib.sapr_viterbi_diag_scores(*args(tie=7)) == ERR_ARG
    assert lib.sapr_viterbi_diag_scores(*args(topo=9)) == ERR_ARG
    assert lib.sapr_viterbi_diag_scores(*args(D=12)) == ERR_UNSUPPORTED and "instantiated" in _err(lib)
    assert lib.sapr_viterbi_workspace_bytes(3, 2, 10, 20, pack.topology, None) == ERR_ARG
    torch.cuda.synchronize()


def test_pruned_decoder_refuses_non_bidiagonal_packs_at_the_c_abi():
    """sapr_diag_pack scans log_trans: a skip transition (a[i, i+2] > 0) clears SAPR_PACK_BIDIAG, and
    sapr_viterbi_decode_pruned — whose passes read only a[i, i] and a[i, i+1] — then returns SAPR_ERR_UNSUPPORTED
    instead of decoding silently wrong (through the C ABI and through torch.ops.sapr.viterbi_decode_best alike)."""
    import torch
    from sapr_amd import _lib, torch_ops  # noqa: F401  (registers torch.ops.sapr.*)
    from sapr_amd.trellis import DiagModelPack, FeatureBatch
    from tests._synth import trained_like_models
    lib = _lib.load()
    sp, A, mu, cv = trained_like_models(3, 8, 13, seed=2)
    ok = DiagModelPack.from_params(sp, A, mu, cv)
    assert ok.flags & _lib.PACK_BIDIAG and ok.prunable
    A2 = A.copy()
    A2[:, 3, 3] -= 0.1
    A2[:, 3, 5] += 0.1
    skip = DiagModelPack.from_params(sp, A2, mu, cv)
    assert not (skip.flags & _lib.PACK_BIDIAG) and skip.flags & _lib.PACK_BOUND_OK and not skip.prunable
    # even a caller that lies about the topology on the Python side cannot get past the C entry point
    batch = FeatureBatch.from_arrays([np.zeros((20, 13), np.float32) for _ in range(3)], layout="TD")
    nb = C.c_size_t(0)
    assert lib.sapr_viterbi_pruned_workspace_bytes(3, 3, 10, 20, C.byref(nb)) == 0
    ws = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
    bw = torch.empty(3, dtype=torch.int32, device="cuda")
    bs = torch.empty(3, dtype=torch.float64, device="cuda")
    path = torch.empty(60, dtype=torch.int32, device="cuda")
    call = lambda pack: lib.sapr_viterbi_decode_pruned(  # noqa: E731
        _lib.ptr(batch.feats), _lib.ptr(batch.offsets), _lib.ptr(batch.order), 3, 13, 20, _lib.ptr(pack.blob), 3, 10,
        1, 1, pack.flags, _lib.ptr(ws), nb.value, _lib.ptr(bw), _lib.ptr(bs), _lib.ptr(path), None)
    assert call(ok) == 0
    assert call(skip) == ERR_UNSUPPORTED and "bidiagonal" in _err(lib)
    with pytest.raises(_lib.SaprHipError):
        torch.ops.sapr.viterbi_decode_best(batch.feats, batch.offsets, batch.order, skip.blob, 3, 10, 13, 20, 1, 1,
                                           skip.flags)
    torch.cuda.synchronize()


def test_mfcc_plan_validation_and_two_pass_workspace():
    import torch
    from sapr_amd import _lib
    from sapr_amd.frontend import BENCH, MfccPlan
    lib = _lib.load()
    h = C.c_void_p()
    base = [16000.0, 512, 400, 160, 40, 13, 0.0, 0.0, 80.0, 0.0, 0]
    assert lib.sapr_mfcc_plan_create(*base, 101, C.byref(h)) == 0 and h.value
    assert lib.sapr_mfcc_plan_destroy(h) == 0
    for bad in ([16000.0, 1000] + base[2:], base[:2] + [600] + base[3:], base[:4] + [200] + base[5:],
                base[:5] + [17] + base[6:]):
        assert lib.sapr_mfcc_plan_create(*bad, 101, C.byref(h)) == ERR_ARG, bad
        assert _err(lib)
    assert lib.sapr_mfcc_plan_create(*base, -1, C.byref(h)) == ERR_ARG
    assert lib.sapr_mfcc_plan_create(*base, 101, None) == ERR_ARG
    # a two-pass plan refuses to run without its workspace
    plan = MfccPlan(**BENCH, max_frames=0)
    assert plan.two_pass
    pcm = torch.zeros(16000, device="cuda")
    so = torch.tensor([0, 16000], dtype=torch.int64, device="cuda")
    fo = torch.tensor([0, 101], dtype=torch.int64, device="cuda")
    out = torch.empty((101, 13), device="cuda")
    rc = lib.sapr_mfcc_batch(plan._h, _lib.ptr(pcm), _lib.ptr(so), _lib.ptr(fo), 1, 101, _lib.ptr(out), 0, None, 0, None)
    assert rc == ERR_WORKSPACE and "workspace" in _err(lib)
    assert lib.sapr_mfcc_batch(None, _lib.ptr(pcm), _lib.ptr(so), _lib.ptr(fo), 1, 101, _lib.ptr(out), 0, None, 0, None) == ERR_ARG
    torch.cuda.synchronize()


def test_mfcc_frame_count_smaller_than_the_device_offsets_writes_nan_not_past_the_buffers():
    """include/sapr_hip.h: total_frames == frame_offsets[n_utts] is a hard precondition of sapr_mfcc_batch (buffers
    and launch are sized from total_frames, the offsets are read on the device only).  The wave-private core checks
    it on the device: offsets that describe MORE frames make it write NaN into out[0 : total_frames] and nothing
    anywhere else — neither behind `out` nor behind the workspace (whose tail holds the utterance maxima)."""
    import torch
    from sapr_amd import _lib
    from sapr_amd.frontend import BENCH, MfccPlan
    lib = _lib.load()
    plan = MfccPlan(**BENCH, max_frames=101)
    assert plan.two_pass   # the 512-point presets run the wave-private core through the log-mel workspace
    n, T = 8, 101
    pcm = torch.randn(n * 16000, device="cuda") * 0.1
    so = (torch.arange(n + 1, dtype=torch.int64) * 16000).cuda()
    fo = (torch.arange(n + 1, dtype=torch.int64) * T).cuda()            # the device says 808 frames ...
    claimed = (n - 1) * T                                               # ... the caller sizes everything for 707
    need = C.c_size_t(0)
    assert lib.sapr_mfcc_workspace_bytes(plan._h, claimed, n, C.byref(need)) == 0 and need.value > 0
    ws = torch.full((need.value + 4096,), 0x5A, dtype=torch.uint8, device="cuda")
    out = torch.full((n * T, 13), 7.0, device="cuda")
    rc = lib.sapr_mfcc_batch(plan._h, _lib.ptr(pcm), _lib.ptr(so), _lib.ptr(fo), n, claimed, _lib.ptr(out), 0,
                             _lib.ptr(ws), need.value, _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert bool(torch.isnan(out[:claimed]).all()), "a frame-count mismatch must not pass for features"
    assert bool((out[claimed:] == 7.0).all()), "rows behind the caller's frame count were written"
    assert bool((ws[need.value:] == 0x5A).all()), "bytes behind the workspace were written"
    # the consistent call on the same buffers still works
    rc = lib.sapr_mfcc_batch(plan._h, _lib.ptr(pcm), _lib.ptr(so[: n]), _lib.ptr(fo[: n]), n - 1, claimed, _lib.ptr(out),
                             0, _lib.ptr(ws), need.value, _lib.current_stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out[:claimed]).all()) and bool((out[claimed:] == 7.0).all())


def test_custom_path_limits():
    from sapr_amd import _lib
    lib = _lib.load()
    n = C.c_size_t(0)
    assert lib.sapr_custom_update_b_workspace_bytes(10, 1, 13, 10, C.byref(n)) == 0 and n.value > 0
    assert lib.sapr_custom_update_b_workspace_bytes(10, 1, 41, 10, C.byref(n)) == ERR_UNSUPPORTED   # D <= 40
    assert lib.sapr_custom_update_b_workspace_bytes(10, 1, 13, 21, C.byref(n)) == ERR_UNSUPPORTED   # S <= 20
    assert "custom-HMM kernels support" in _err(lib)
