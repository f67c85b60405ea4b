"""The C ABI: every function include/sdag_hip.h declares is exported and bound (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "sdag_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdag_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from sdag_amd import _lib

    names = _declared()
    assert len(names) >= 13
    assert sorted(_lib.SIGNATURES) == names
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert getattr(handle, n) is not None


def test_error_reporting_without_gpu():
    from sdag_amd import _lib

    lib = _lib.lib()
    assert lib.sdag_version() >= 1
    need = ctypes.c_int64(0)
    rc = lib.sdag_desc_build(10, 20, 0, None, 0, None, None, 0, 0, None, 0, ctypes.byref(need))
    assert rc == _lib.SDAG_ERR_ARG and b"sys_user_len" in lib.sdag_last_error()
    rc = lib.sdag_desc_build(10, 2, 0, None, 5, None, None, 0, 0, None, 0, ctypes.byref(need))
    assert rc == _lib.SDAG_ERR_CAPACITY and need.value > 32
    assert lib.sdag_dense_workspace_bytes(32, 1000, 64, 100) < 0   # k above the maximum
    assert lib.sdag_attn_workspace_bytes(None, 1, 4, 1) < 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sparse-document-attention-rag-_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
