"""CPU: libsapr_hip.so loads and exports every function include/sapr_hip.h declares
(no compute call is made — there is no GPU in the build container)."""
import ctypes
import os
import re

from sapr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "sapr_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sapr_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_table_agree():
    names = _declared()
    assert names, "no declarations parsed"
    assert sorted(_lib.SIGNATURES) == names


def test_shared_object_exports_every_declared_symbol():
    from sapr_amd.build import build
    path = build(verbose=False)
    lib = ctypes.CDLL(path)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.sapr_abi_version.restype = ctypes.c_int
    assert lib.sapr_abi_version() == 2
    lib.sapr_last_error.restype = ctypes.c_char_p
    assert isinstance(lib.sapr_last_error(), bytes)


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    n = ctypes.c_size_t(0)
    assert lib.sapr_viterbi_workspace_bytes(10, 11, 10, 101, 1, ctypes.byref(n)) == 0
    assert n.value == 11 * 101 * 256 * 4
    assert lib.sapr_viterbi_workspace_bytes(10, 0, 10, 101, 1, ctypes.byref(n)) == -1
    assert b"bad sizes" in lib.sapr_last_error()
