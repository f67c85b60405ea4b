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
    # sdag_attn_fwd checks its arguments before it touches the device (the pointers are never dereferenced here)
    fake = 0x10000

    def attn(L=64, D=128, o_sl=4 * 128, ptr=fake, scale=0.1):
        q_sl = 4 * 128
        return lib.sdag_attn_fwd(ptr, fake, fake, fake, L * q_sl, 128, q_sl, L * 128, 128, 128, L * 128, 128, 128,
                                 0, o_sl, 128, 1, 4, 1, L, D, scale, None, None, None, 0, 0, None)  # B = 1: o_sb unused

    assert attn(D=64) == _lib.SDAG_ERR_ARG and b"head_dim" in lib.sdag_last_error()
    assert attn(scale=0.0) == _lib.SDAG_ERR_ARG and b"scale" in lib.sdag_last_error()
    assert attn(ptr=fake + 8) == _lib.SDAG_ERR_ARG and b"aligned" in lib.sdag_last_error()
    assert attn(L=1 << 20, o_sl=2048) == _lib.SDAG_ERR_ARG and b"slice of O" in lib.sdag_last_error()
    assert attn() == _lib.SDAG_ERR_DESC   # everything else in order: the missing descriptor is what is reported
    # the decode entry points check their arguments before anything is launched as well
    dec = (fake, fake, fake, fake, 4 * 128, 128, 8 * 128, 8 * 128, 128, 8 * 128, 8 * 128, 128, 4 * 128, 128, 1, 4, 1, 8, 128, 0.1)
    assert lib.sdag_decode_attn_prealloc(*dec, None, None, 0, fake, 1 << 20, 0, None) == _lib.SDAG_ERR_ARG and b"n_rows" in lib.sdag_last_error()
    assert lib.sdag_decode_attn_prealloc(*dec, fake, fake, 9, fake, 1 << 20, 0, None) == _lib.SDAG_ERR_ARG and b"pad_end" in lib.sdag_last_error()
    assert lib.sdag_decode_attn_padded(*dec, None, 0, fake, 1 << 20, 0, None) == _lib.SDAG_ERR_ARG and b"valid_len" in lib.sdag_last_error()
    assert lib.sdag_decode_attn_prealloc(*dec, fake, None, 0, None, 0, 0, None) == _lib.SDAG_ERR_CAPACITY


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sparse-document-attention-rag-_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
