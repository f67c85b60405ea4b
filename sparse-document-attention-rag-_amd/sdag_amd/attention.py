"""SDAG masked attention prefill on the HIP kernel (``sdag_attn_fwd``).

Replaces the attention call the reference reaches through ``model(input_ids, attention_mask=[1,1,L,L])``
(src/pipeline/sparse_attention_RAG/SDAG.py:176): same inputs and output layout as a transformers attention
function -- query [B, Hq, L, D], key / value [B, Hkv, L, D] (any batch / head / row strides, D contiguous),
output [B, L, Hq, D].
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from .descriptor import BatchDescriptor, Descriptor

_workspaces: Dict[torch.device, torch.Tensor] = {}


class _on_device:
    """``torch.cuda.device(dev)`` only when ``dev`` is not already current (the context manager costs more than the
    kernel launch it would wrap in a single-token step)."""

    __slots__ = ("ctx",)

    def __init__(self, dev: torch.device):
        self.ctx = None if dev.index is None or torch.cuda.current_device() == dev.index else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _workspace(device: torch.device, nbytes: int) -> Optional[torch.Tensor]:
    if nbytes == 0:
        return None
    ws = _workspaces.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    return ws


def sdag_attention(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    desc: Descriptor,
    scaling: Optional[float] = None,
    out_dtype: torch.dtype = torch.bfloat16,
    out: Optional[torch.Tensor] = None,
) -> torch.Tensor:
    """softmax(Q K^T * scaling + M) V with M described by ``desc``; returns [B, L, Hq, D]."""
    if not (query.is_cuda and key.is_cuda and value.is_cuda):
        raise RuntimeError("sdag_attention runs on the GPU only (no CPU fallback)")
    if query.dtype != torch.bfloat16 or key.dtype != torch.bfloat16 or value.dtype != torch.bfloat16:
        raise TypeError("sdag_attention takes bf16 query/key/value")
    B, Hq, L, D = query.shape
    Bk, Hkv, S, Dk = key.shape
    if value.shape != key.shape or Bk != B or Dk != D or S != L:
        raise ValueError(f"shape mismatch: q {tuple(query.shape)} k {tuple(key.shape)} v {tuple(value.shape)}")
    if desc.L != L:
        raise ValueError(f"descriptor is for L={desc.L}, tensors have L={L}")
    if query.stride(3) != 1:
        query = query.contiguous()
    if key.stride(3) != 1:
        key = key.contiguous()
    if value.stride(3) != 1:
        value = value.contiguous()
    if scaling is None:
        scaling = D ** -0.5
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("out_dtype must be bfloat16 or float32")
    dev = query.device
    if out is None:
        out = torch.empty((B, L, Hq, D), dtype=out_dtype, device=dev)
    elif out.shape != (B, L, Hq, D) or out.dtype != out_dtype or out.stride(3) != 1 or out.device != dev:
        raise ValueError("bad out tensor")
    ws_bytes = desc.workspace_bytes(B, Hq, Hkv)
    ws = _workspace(dev, ws_bytes)
    ddev = desc.device(dev)
    with _on_device(dev):
        rc = _lib.lib().sdag_attn_fwd(
            query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(),
            query.stride(0), query.stride(1), query.stride(2),
            key.stride(0), key.stride(1), key.stride(2),
            value.stride(0), value.stride(1), value.stride(2),
            out.stride(0), out.stride(1), out.stride(2),
            B, Hq, Hkv, L, D, float(scaling),
            desc.host_ptr, ddev.data_ptr(),
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
            _lib.OUT_F32 if out_dtype == torch.float32 else _lib.OUT_BF16,
            _lib.stream_ptr(dev))
    _lib.check(rc)
    return out


class BoundAttention:
    """``sdag_attention`` with every argument resolved once, for callers that reuse the same tensors
    (static serving buffers, benchmarks): ``__call__`` is a single C call on torch's current stream."""

    def __init__(self, query, key, value, desc: Descriptor, out: torch.Tensor, scaling: Optional[float] = None):
        B, Hq, L, D = query.shape
        Hkv = key.shape[1]
        if query.dtype != torch.bfloat16 or key.dtype != torch.bfloat16 or value.dtype != torch.bfloat16:
            raise TypeError("bf16 query/key/value expected")
        if query.stride(3) != 1 or key.stride(3) != 1 or value.stride(3) != 1 or out.stride(3) != 1:
            raise ValueError("head_dim must be contiguous")
        if out.shape != (B, L, Hq, D) or out.dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("bad out tensor")
        self.device = query.device
        self._keep = (query, key, value, out, desc)
        ws = _workspace(self.device, desc.workspace_bytes(B, Hq, Hkv))
        self._ws = ws
        self._fn = _lib.lib().sdag_attn_fwd
        self._args = (
            query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(),
            query.stride(0), query.stride(1), query.stride(2),
            key.stride(0), key.stride(1), key.stride(2),
            value.stride(0), value.stride(1), value.stride(2),
            out.stride(0), out.stride(1), out.stride(2),
            B, Hq, Hkv, L, D, float(D ** -0.5 if scaling is None else scaling),
            desc.host_ptr, desc.device(self.device).data_ptr(),
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
            _lib.OUT_F32 if out.dtype == torch.float32 else _lib.OUT_BF16)

    def __call__(self, stream: Optional[int] = None) -> None:
        rc = self._fn(*self._args, _lib.stream_ptr(self.device) if stream is None else stream)
        if rc:
            _lib.check(rc)


class BoundBatchAttention:
    """Masked prefill attention of B sequences, each with its own descriptor, in one launch on bound tensors.

    query [B, Hq, Lmax, D], key / value [B, Hkv, Lmax, D], out [B, Lmax, Hq, D]; sequence b uses rows
    [0, L_b) of its slice (L_b from its descriptor); rows beyond L_b are left untouched in ``out``.
    """

    def __init__(self, query, key, value, batch: BatchDescriptor, out: torch.Tensor, scaling: Optional[float] = None):
        B, Hq, L, D = query.shape
        Hkv = key.shape[1]
        if not query.is_cuda:
            raise RuntimeError("sdag_attention_batch runs on the GPU only (no CPU fallback)")
        if query.dtype != torch.bfloat16 or key.dtype != torch.bfloat16 or value.dtype != torch.bfloat16:
            raise TypeError("bf16 query/key/value expected")
        if query.stride(3) != 1 or key.stride(3) != 1 or value.stride(3) != 1 or out.stride(3) != 1:
            raise ValueError("head_dim must be contiguous")
        if key.shape != value.shape or key.shape[0] != B or key.shape[2] != L or key.shape[3] != D:
            raise ValueError("shape mismatch")
        if out.shape != (B, L, Hq, D) or out.dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("bad out tensor")
        if batch.B != B or batch.Lmax > L:
            raise ValueError("batch descriptor does not match the tensors")
        self.device = query.device
        self._keep = (query, key, value, out, batch)
        ws = _workspace(self.device, batch.workspace_bytes(Hq, Hkv))
        self._ws = ws  # keeps the scratch alive even if the per-device cache is later replaced by a larger one
        self._fn = _lib.lib().sdag_attn_fwd_batch
        self._args = (
            query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(),
            query.stride(0), query.stride(1), query.stride(2),
            key.stride(0), key.stride(1), key.stride(2),
            value.stride(0), value.stride(1), value.stride(2),
            out.stride(0), out.stride(1), out.stride(2),
            B, Hq, Hkv, L, D, float(D ** -0.5 if scaling is None else scaling),
            batch.host_ptr, batch.device(self.device).data_ptr(),
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
            _lib.OUT_F32 if out.dtype == torch.float32 else _lib.OUT_BF16)

    def __call__(self, stream: Optional[int] = None) -> None:
        rc = self._fn(*self._args, _lib.stream_ptr(self.device) if stream is None else stream)
        if rc:
            _lib.check(rc)


def sdag_attention_batch(query, key, value, batch: BatchDescriptor, scaling: Optional[float] = None,
                         out_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """One-shot form of ``BoundBatchAttention``; rows beyond each sequence's length come back as zeros."""
    B, Hq, L, D = query.shape
    out = torch.zeros((B, L, Hq, D), dtype=out_dtype, device=query.device)
    BoundBatchAttention(query, key, value, batch, out, scaling)()
    return out


def decode_attention(query, key, value, scaling: Optional[float] = None, out_dtype: torch.dtype = torch.bfloat16,
                     valid_len: Optional[torch.Tensor] = None, pad_end: int = 0,
                     n_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention of ONE new token per sequence over the whole key/value cache (no mask): the attention of the
    reference's decode steps (src/pipeline/sparse_attention_RAG/SDAG.py:191-199).

    query [B, Hq, 1, D], key / value [B, Hkv, S, D] bf16 (any batch / head / row strides, D contiguous); returns
    [B, 1, Hq, D].  ``valid_len`` (int32 [B] on the device) with ``pad_end``: right-padded prompts decoded together --
    cache rows [valid_len[b], pad_end) of sequence b are padding and take no part.  ``n_rows`` (one int32 on the
    device): key / value are a preallocated cache of which only the first ``n_rows`` rows hold keys -- the count is read
    by the kernel, so a captured step can be replayed while the cache fills."""
    if not (query.is_cuda and key.is_cuda and value.is_cuda):
        raise RuntimeError("decode_attention runs on the GPU only (no CPU fallback)")
    if query.dtype != torch.bfloat16 or key.dtype != torch.bfloat16 or value.dtype != torch.bfloat16:
        raise TypeError("decode_attention takes bf16 query/key/value")
    B, Hq, Lq, D = query.shape
    Bk, Hkv, S, Dk = key.shape
    if Lq != 1 or value.shape != key.shape or Bk != B or Dk != D or S < 1:
        raise ValueError(f"shape mismatch: q {tuple(query.shape)} k {tuple(key.shape)} v {tuple(value.shape)}")
    if query.stride(3) != 1:
        query = query.contiguous()
    if key.stride(3) != 1:
        key = key.contiguous()
    if value.stride(3) != 1:
        value = value.contiguous()
    if out_dtype not in (torch.bfloat16, torch.float32):
        raise TypeError("out_dtype must be bfloat16 or float32")
    dev = query.device
    out = torch.empty((B, 1, Hq, D), dtype=out_dtype, device=dev)
    lib = _lib.lib()
    ws = _workspace(dev, int(lib.sdag_decode_workspace_bytes(B, Hq, S)))
    head = (query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(),
            query.stride(0), query.stride(1),
            key.stride(0), key.stride(1), key.stride(2),
            value.stride(0), value.stride(1), value.stride(2),
            out.stride(0), out.stride(2),
            B, Hq, Hkv, S, D, float(D ** -0.5 if scaling is None else scaling))
    tail = (ws.data_ptr(), ws.numel(), _lib.OUT_F32 if out_dtype == torch.float32 else _lib.OUT_BF16, _lib.stream_ptr(dev))
    if valid_len is not None and (valid_len.dtype != torch.int32 or valid_len.device != dev or valid_len.numel() != B
                                  or not valid_len.is_contiguous()):
        raise ValueError("valid_len must be a contiguous int32 [B] tensor on the tensors' device")
    if n_rows is not None and (n_rows.dtype != torch.int32 or n_rows.device != dev or n_rows.numel() != 1):
        raise ValueError("n_rows must be one int32 on the tensors' device")
    with _on_device(dev):
        if n_rows is not None:
            rc = lib.sdag_decode_attn_prealloc(*head, n_rows.data_ptr(), valid_len.data_ptr() if valid_len is not None else None,
                                               int(pad_end), *tail)
        elif valid_len is None:
            rc = lib.sdag_decode_attn(*head, *tail)
        else:
            rc = lib.sdag_decode_attn_padded(*head, valid_len.data_ptr(), int(pad_end), *tail)
    _lib.check(rc)
    return out
