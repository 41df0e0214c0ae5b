"""CPU-side checks of the boundary: the library loads and exports every symbol the header declares
(no compute calls without a GPU); argument validation that does not need a device."""
import os
import re

from clip_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "clip_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.clip_abi_version() == 3


def test_argument_errors_without_a_device():
    lib = _lib.load()
    assert lib.clip_create(None, 0, None) == _lib.CLIP_ERR_ARG
    assert b"null" in lib.clip_last_error()
    assert lib.clip_load_weight(None, b"x", None, 0) == _lib.CLIP_ERR_ARG
    assert lib.clip_epilogue(None, None, 0.05, 1, 7, None, None, None, None, None) == _lib.CLIP_ERR_ARG
    assert lib.clip_gemm_bf16(None, None, None, None, None, 16, 16, 100, 100, 100, 16, 0, 0, 1, 0, None) == _lib.CLIP_ERR_ARG
