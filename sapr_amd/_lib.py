"""ctypes binding of libsapr_hip.so (the C ABI declared in include/sapr_hip.h).

This is the stub a maintainer of the reference would add: the
reference has no FFI of its own, its hot path is numpy / hmmlearn / librosa calls.

The library is REQUIRED: there is no CPU fallback anywhere in ``sapr_amd``.  If the
shared object is missing or a symbol is absent, import of the product modules works
(so models can be unpickled on a CPU box, SURVEY.md §8b "Ownership") but the first
call into a kernel raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SAPR_LIB: developer override (an experimental build of the same ABI, e.g. scripts/experiments/ablate_mfcc.sh)
LIB_PATH = os.environ.get("SAPR_LIB") or os.path.join(HERE, "libsapr_hip.so")

c_void_p, c_int, c_int32, c_int64, c_size_t, c_double, c_float = (
    C.c_void_p, C.c_int, C.c_int32, C.c_int64, C.c_size_t, C.c_double, C.c_float)

# name -> (restype, argtypes); must list every function include/sapr_hip.h declares
# (tests/test_capi_symbols.py parses the header and checks this table and the .so).
SIGNATURES = {
    "sapr_abi_version": (c_int, []),
    "sapr_last_error": (C.c_char_p, []),
    "sapr_device_info": (c_int, [c_int, C.POINTER(c_int), C.POINTER(c_int), C.c_char_p, c_size_t]),
    "sapr_selftest_lse": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sapr_viterbi_workspace_bytes": (c_int, [c_int64, c_int32, c_int32, c_int32, c_int32,
                                             C.POINTER(c_size_t)]),
    "sapr_diag_pack_bytes": (c_int, [c_int32, c_int32, c_int32, C.POINTER(c_size_t)]),
    "sapr_diag_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                               c_void_p, c_size_t, C.POINTER(c_int32), c_void_p]),
    "sapr_viterbi_diag_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                         c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "sapr_viterbi_backtrace": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                       c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p]),
    "sapr_viterbi_pruned_workspace_bytes": (c_int, [c_int64, c_int32, c_int32, c_int32, C.POINTER(c_size_t)]),
    "sapr_viterbi_decode_pruned": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p,
                                           c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    "sapr_viterbi_pruned_views": (c_int, [c_int64, c_int32, c_int32, c_void_p] + [C.POINTER(c_void_p)] * 5),
    "sapr_fb_workspace_bytes": (c_int, [c_int64, c_int64, c_int32, c_int32, c_int32, C.POINTER(c_size_t)]),
    "sapr_stats_width": (c_int, [c_int32, c_int32, C.POINTER(c_int32)]),
    "sapr_forward_diag": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                  c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "sapr_estep_diag": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32,
                                c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_size_t,
                                c_void_p, c_void_p, c_void_p]),
    "sapr_colsum_f32": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "sapr_custom_estep": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32] + [c_void_p] * 5
                          + [c_int64] + [c_void_p] * 6 + [c_void_p]),
    "sapr_custom_stage_features": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int64, c_void_p, c_void_p,
                                           c_void_p]),
    "sapr_custom_estep_staged": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32] + [c_void_p] * 5
                                 + [c_int64] + [c_void_p] * 6 + [c_void_p, c_void_p, c_void_p]),
    "sapr_custom_piece": (c_int, [c_int32, c_void_p, c_int32, c_int32, c_int32] + [c_void_p] * 11 + [c_void_p]),
    "sapr_custom_emission_exact": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32]
                                   + [c_void_p] * 4 + [c_void_p]),
    "sapr_custom_decode": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32]
                           + [c_void_p] * 11 + [c_void_p]),
    "sapr_custom_update_b_workspace_bytes": (c_int, [c_int64, c_int32, c_int32, c_int32, C.POINTER(c_size_t)]),
    "sapr_custom_update_b": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_int64]
                             + [c_void_p] * 4 + [c_size_t, c_void_p]),
    "sapr_custom_update_b_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                          c_int64] + [c_void_p] * 3 + [c_size_t, c_void_p]),
    "sapr_custom_update_b_scatter": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p,
                                             c_int64] + [c_void_p] * 3 + [c_size_t, c_void_p]),
    "sapr_custom_update_b_moments": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p,
                                             c_void_p, c_void_p, c_size_t, c_void_p]),
    "sapr_custom_normalise": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
    "sapr_custom_fold_rows": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "sapr_custom_global_workspace_bytes": (c_int, [c_int64, c_int64, c_int32, C.POINTER(c_size_t)]),
    "sapr_custom_global_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sapr_custom_global_cov": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sapr_pcm16_to_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "sapr_resample_poly": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_int32, c_void_p, c_int32,
                                   c_int32, c_void_p, c_void_p]),
    "sapr_mfcc_plan_create": (c_int, [c_double, c_int32, c_int32, c_int32, c_int32, c_int32, c_double,
                                      c_double, c_double, c_double, c_int32, c_int32,
                                      C.POINTER(c_void_p)]),
    "sapr_mfcc_plan_destroy": (c_int, [c_void_p]),
    "sapr_mfcc_plan_info": (c_int, [c_void_p, C.POINTER(c_int32), C.POINTER(c_int32),
                                    C.POINTER(c_int64), C.POINTER(c_int32)]),
    "sapr_mfcc_workspace_bytes": (c_int, [c_void_p, c_int64, c_int64, C.POINTER(c_size_t)]),
    "sapr_mfcc_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32,
                                c_void_p, c_size_t, c_void_p]),
    "sapr_mfcc_batch_stamped": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32,
                                        c_void_p, c_void_p]),
}

TOPO_DENSE, TOPO_BIDIAG = 0, 1
TIE_LOW, TIE_HIGH = 0, 1
PACK_FAST_DIV, PACK_BOUND_OK, PACK_GEMM_OK, PACK_BIDIAG, PACK_EXACT_ONLY = 1, 2, 4, 8, 16
ESTEP_STAGED = 256
SUM_PAIRWISE, SUM_TVIEW, SUM_SEQ = 0, 1, 2

_lib = None


class SaprHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SaprHipError(
            f"{LIB_PATH} is missing: build it with `python -m sapr_amd.build` "
            "(hipcc --offload-arch=gfx950).  sapr_amd has no CPU fallback.")
    # PyTorch-ROCm bundles its own HIP/HSA runtime under the same sonames as /opt/rocm's.  It must be
    # loaded FIRST so that libsapr_hip.so binds to the runtime that owns torch's device memory and
    # streams; loaded the other way round, two runtimes fight over the device
    # ("no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # stale .so
            raise SaprHipError(f"{LIB_PATH} lacks symbol {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sapr_abi_version() != 2:
        raise SaprHipError("libsapr_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sapr_last_error().decode("utf-8", "replace")
        raise SaprHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise SaprHipError("no HIP device visible: sapr_amd kernels need an MI355X (gfx950); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_host(*tensors):
    """Device results as numpy arrays through page-locked buffers (torch's caching host allocator keeps them between
    calls): large results cross PCIe at the link rate instead of through a pageable staging copy.  One synchronisation
    for all of them."""
    import torch
    outs = []
    for t in tensors:
        if t.is_cuda:
            o = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            o.copy_(t, non_blocking=True)
        else:
            o = t
        outs.append(o)
    if any(t.is_cuda for t in tensors):
        torch.cuda.current_stream().synchronize()
    return tuple(o.numpy() for o in outs)

