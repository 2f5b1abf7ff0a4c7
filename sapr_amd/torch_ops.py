"""``torch.ops.sapr.*``: the hot-path entry points of ``include/sapr_hip.h`` registered as PyTorch custom ops
(``torch.library``), so that callers holding torch tensors on a ROCm device can use them without the ctypes
plumbing of ``_lib.py``.  The C ABI stays the boundary: every op below is a thin launch of ONE C entry point
on the current stream; there is no CPU implementation (calling an op on CPU tensors raises).

    import sapr_amd.torch_ops                       # registers the ops
    pcm  = torch.ops.sapr.pcm16_to_f32(pcm16)
    feat = torch.ops.sapr.mfcc_batch(pcm, sample_offsets, frame_offsets, plan.handle, plan.d_out)
    word, score, path = torch.ops.sapr.viterbi_decode_best(feat, frame_offsets, order, pack.blob, W, S, D,
                                                          max_T, tie, sum_order, pack.flags)
    loglik, stats = torch.ops.sapr.hmm_estep(feat, offsets, slot_utt, tile_model, model_tile_off, pack.blob,
                                             W, S, D, max_T, topology, fast_div)
    # the reference's from-scratch HMM (custom_hmm.py), model arrays as custom_hmm.model_arrays() prepares them
    gamma, utt = torch.ops.sapr.custom_estep(feat, offsets, means, inv, cterm, A, logA)
    scores, paths, word, best, best_path = torch.ops.sapr.custom_decode(feat, offsets, means, inv, cterm, A, logA,
                                                                        num_states, Tq)

Reference call sites replaced: ``librosa.feature.mfcc`` (mfcc_extract.py:15-23), ``GaussianHMM.decode`` over the
vocabulary + arg-max (decoder.py:35-49), the E-step of ``GaussianHMM.fit`` (hmmlearn_hmm.py:103), the E-step of
``HMM.baum_welch`` (custom_hmm.py:422-439) and ``HMM.decode`` over the vocabulary (custom_hmm.py:462-514,
decoder.py:42-47).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

_LIB = torch.library.Library("sapr", "DEF")
_LIB.define("pcm16_to_f32(Tensor pcm16) -> Tensor")
_LIB.define("mfcc_batch(Tensor pcm, Tensor sample_offsets, Tensor frame_offsets, int plan, int d_out) -> Tensor")
_LIB.define("viterbi_decode_best(Tensor feats, Tensor offsets, Tensor order, Tensor pack, int W, int S, int D, "
            "int max_T, int tie, int sum_order, int pack_flags) -> (Tensor, Tensor, Tensor)")
_LIB.define("hmm_estep(Tensor feats, Tensor offsets, Tensor slot_utt, Tensor tile_model, Tensor model_tile_off, "
            "Tensor pack, int W, int S, int D, int max_T, int topology, int fast_div) -> (Tensor, Tensor)")
_LIB.define("custom_estep(Tensor feats, Tensor offsets, Tensor means, Tensor inv, Tensor cterm, Tensor A, Tensor logA) "
            "-> (Tensor, Tensor)")
_LIB.define("custom_decode(Tensor feats, Tensor offsets, Tensor means, Tensor inv, Tensor cterm, Tensor A, Tensor logA, "
            "int num_states, int Tq) -> (Tensor, Tensor, Tensor, Tensor, Tensor)")


def _check_dev(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _lib.SaprHipError("torch.ops.sapr.* run on the GPU only (no CPU implementation)")
        if not t.is_contiguous():
            raise ValueError("torch.ops.sapr.*: tensors must be contiguous")


def _pcm16_to_f32(pcm16):
    _check_dev(pcm16)
    if pcm16.dtype != torch.int16:
        raise ValueError("pcm16 must be int16")
    out = torch.empty(pcm16.shape, dtype=torch.float32, device=pcm16.device)
    _lib.check(_lib.load().sapr_pcm16_to_f32(_lib.ptr(pcm16), pcm16.numel(), _lib.ptr(out), _lib.current_stream()),
               "sapr_pcm16_to_f32")
    return out


def _mfcc_batch(pcm, sample_offsets, frame_offsets, plan, d_out):
    _check_dev(pcm, sample_offsets, frame_offsets)
    lib = _lib.load()
    n_utts = sample_offsets.numel() - 1
    total_frames = int(frame_offsets[-1].item())
    out = torch.empty((total_frames, d_out), dtype=torch.float32, device=pcm.device)
    h = C.c_void_p(plan)
    nb = C.c_size_t(0)
    _lib.check(lib.sapr_mfcc_workspace_bytes(h, total_frames, n_utts, C.byref(nb)), "sapr_mfcc_workspace_bytes")
    ws = torch.empty(int(nb.value), dtype=torch.uint8, device=pcm.device) if nb.value else None
    _lib.check(lib.sapr_mfcc_batch(h, _lib.ptr(pcm), _lib.ptr(sample_offsets), _lib.ptr(frame_offsets), n_utts,
                                   total_frames, _lib.ptr(out), 0, _lib.ptr(ws), int(nb.value), _lib.current_stream()),
               "sapr_mfcc_batch")
    return out


def _viterbi_decode_best(feats, offsets, order, pack, W, S, D, max_T, tie, sum_order, pack_flags):
    _check_dev(feats, offsets, order, pack)
    lib = _lib.load()
    n = offsets.numel() - 1
    dev = feats.device
    nb = C.c_size_t(0)
    _lib.check(lib.sapr_viterbi_pruned_workspace_bytes(n, W, S, max_T, C.byref(nb)), "sapr_viterbi_pruned_workspace_bytes")
    ws = torch.empty(max(int(nb.value), 1), dtype=torch.uint8, device=dev)
    bw = torch.empty(n, dtype=torch.int32, device=dev)
    bs = torch.empty(n, dtype=torch.float64, device=dev)
    path = torch.empty(feats.shape[0], dtype=torch.int32, device=dev)
    _lib.check(lib.sapr_viterbi_decode_pruned(_lib.ptr(feats), _lib.ptr(offsets), _lib.ptr(order), n, D, max_T,
                                              _lib.ptr(pack), W, S, tie, sum_order, pack_flags, _lib.ptr(ws),
                                              int(nb.value), _lib.ptr(bw), _lib.ptr(bs), _lib.ptr(path),
                                              _lib.current_stream()), "sapr_viterbi_decode_pruned")
    return bw, bs, path


def _hmm_estep(feats, offsets, slot_utt, tile_model, model_tile_off, pack, W, S, D, max_T, topology, fast_div):
    _check_dev(feats, offsets, slot_utt, tile_model, model_tile_off, pack)
    lib = _lib.load()
    n = offsets.numel() - 1
    n_tiles = tile_model.numel()
    dev = feats.device
    nb, width = C.c_size_t(0), C.c_int32(0)
    _lib.check(lib.sapr_fb_workspace_bytes(n, n_tiles, S, D, max_T, C.byref(nb)), "sapr_fb_workspace_bytes")
    _lib.check(lib.sapr_stats_width(S, D, C.byref(width)), "sapr_stats_width")
    ws = torch.empty(max(int(nb.value), 1), dtype=torch.uint8, device=dev)
    loglik = torch.zeros(n, dtype=torch.float64, device=dev)
    stats = torch.zeros((W, int(width.value)), dtype=torch.float64, device=dev)
    _lib.check(lib.sapr_estep_diag(_lib.ptr(feats), _lib.ptr(offsets), _lib.ptr(slot_utt), _lib.ptr(tile_model),
                                   _lib.ptr(model_tile_off), n, n_tiles, D, max_T, _lib.ptr(pack), W, S, topology,
                                   fast_div, _lib.ptr(ws), int(nb.value), _lib.ptr(loglik), _lib.ptr(stats),
                                   _lib.current_stream()), "sapr_estep_diag")
    return loglik, stats


def _custom_shapes(feats, means, inv, cterm, A, logA):
    W, S, D = means.shape
    if feats.dim() != 2 or feats.shape[1] != D or feats.dtype != torch.float32:
        raise ValueError(f"feats must be float32 [total_frames, {D}]")
    for name, t, shape in (("inv", inv, (W, S, D, D)), ("cterm", cterm, (W, S)), ("A", A, (W, S, S)),
                           ("logA", logA, (W, S, S))):
        if tuple(t.shape) != shape or t.dtype != torch.float64:
            raise ValueError(f"{name} must be float64 {shape} (custom_hmm.model_arrays)")
    if means.dtype != torch.float64:
        raise ValueError("means must be float64 [W, S, D]")
    return W, S, D


def _custom_estep(feats, offsets, means, inv, cterm, A, logA):
    """One model (W = 1): custom_hmm.py:146-322 for every utterance -> gamma[total_frames][S] (the reference's row
    layout) and utt[n_utts][2 + S + S*S] = {LL, scale, sum_t gamma, sum_t xi} (custom_hmm.py:434-439's summands)."""
    _check_dev(feats, offsets, means, inv, cterm, A, logA)
    W, S, D = _custom_shapes(feats, means, inv, cterm, A, logA)
    if W != 1:
        raise ValueError("custom_estep trains one model: means must be [1, S, D]")
    n, total, dev = offsets.numel() - 1, feats.shape[0], feats.device
    E, al, be, ga = (torch.zeros((max(total, 1), S), dtype=torch.float64, device=dev) for _ in range(4))
    utt = torch.zeros((n, 2 + S + S * S), dtype=torch.float64, device=dev)
    _lib.check(_lib.load().sapr_custom_estep(_lib.ptr(feats), _lib.ptr(offsets), None, n, D, S, 1, _lib.ptr(means),
                                             _lib.ptr(inv), _lib.ptr(cterm), _lib.ptr(A), _lib.ptr(logA), 0,
                                             _lib.ptr(E), _lib.ptr(al), _lib.ptr(be), _lib.ptr(ga), None, _lib.ptr(utt),
                                             _lib.current_stream()), "sapr_custom_estep")
    return ga[:total], utt


def _custom_decode(feats, offsets, means, inv, cterm, A, logA, num_states, Tq):
    """custom_hmm.py:462-514 for every (utterance, model) + decoder.py:35-49's arg-max on the device ->
    scores[n][W], paths[n][W][Tq], best_word[n] (-1: none), best_score[n], best_path[n][Tq]."""
    _check_dev(feats, offsets, means, inv, cterm, A, logA)
    W, S, D = _custom_shapes(feats, means, inv, cterm, A, logA)
    n, dev = offsets.numel() - 1, feats.device
    e_rows = torch.empty(max(n * W * Tq * S, 1), dtype=torch.float64, device=dev)
    scores = torch.zeros((n, W), dtype=torch.float64, device=dev)
    paths = torch.zeros((n, W, Tq), dtype=torch.int32, device=dev)
    bw = torch.full((n,), -1, dtype=torch.int32, device=dev)
    bs = torch.full((n,), float("-inf"), dtype=torch.float64, device=dev)
    bp = torch.zeros((n, Tq), dtype=torch.int32, device=dev)
    _lib.check(_lib.load().sapr_custom_decode(_lib.ptr(feats), _lib.ptr(offsets), n, W, D, S, num_states, Tq,
                                              _lib.ptr(means), _lib.ptr(inv), _lib.ptr(cterm), _lib.ptr(A),
                                              _lib.ptr(logA), _lib.ptr(e_rows), _lib.ptr(scores), _lib.ptr(paths),
                                              _lib.ptr(bw), _lib.ptr(bs), _lib.ptr(bp), _lib.current_stream()),
               "sapr_custom_decode")
    return scores, paths, bw, bs, bp


for _name, _fn in (("pcm16_to_f32", _pcm16_to_f32), ("mfcc_batch", _mfcc_batch),
                   ("viterbi_decode_best", _viterbi_decode_best), ("hmm_estep", _hmm_estep),
                   ("custom_estep", _custom_estep), ("custom_decode", _custom_decode)):
    _LIB.impl(_name, _fn, "CUDA")
