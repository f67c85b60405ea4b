"""The SDAG kernel as a transformers attention function.

transformers calls ``attention_interface(module, query[B,Hq,L,D], key[B,Hkv,S,D], value[B,Hkv,S,D],
attention_mask, dropout=, scaling=, **kwargs) -> (attn_output[B,L,Hq,D], attn_weights | None)`` from every
attention layer (transformers/models/llama/modeling_llama.py:264-277).  The reference reaches it with a dense
additive [1,1,L,L] mask (src/pipeline/sparse_attention_RAG/SDAG.py:167-176); here the mask travels as a
block-sparse descriptor set for the duration of the prefill, and no dense mask exists.

The single-token decode steps (SDAG.py:191-211: one new position against the whole cache, no mask from the
reference) run the key/value-streaming kernel ``sdag_decode_attn``.  Any other call (an ordinary causal prefill, a
padded batch) is handed to transformers' stock SDPA attention unchanged.
"""
from __future__ import annotations

import contextlib
import threading
from typing import Optional

import torch

from .attention import decode_attention, sdag_attention
from .descriptor import Descriptor

ATTN_NAME = "sdag_hip"
_state = threading.local()


def current_descriptor() -> Optional[Descriptor]:
    return getattr(_state, "desc", None)


@contextlib.contextmanager
def active_descriptor(desc: Descriptor):
    """Make ``desc`` the mask of every prefill-shaped attention call on this thread inside the block."""
    prev = getattr(_state, "desc", None)
    _state.desc = desc
    try:
        yield desc
    finally:
        _state.desc = prev


def _mask_allows_everything(attention_mask) -> bool:
    """A decode step of an unpadded sequence: transformers passes no mask, or one that forbids nothing (looking
    into a mask tensor costs a device synchronisation, which only padded batches pay)."""
    if attention_mask is None:
        return True
    if attention_mask.dtype == torch.bool:
        return bool(attention_mask.all())
    return bool((attention_mask == 0).all())


def sdag_hip_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0,
                               scaling: Optional[float] = None, **kwargs):
    desc = current_descriptor()
    if desc is not None and query.shape[2] == key.shape[2] == desc.L:
        if dropout:
            raise NotImplementedError("attention dropout is not supported by the SDAG kernel (inference only)")
        q, k, v = query, key, value
        if q.dtype != torch.bfloat16:
            q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        out = sdag_attention(q, k, v, desc, scaling=scaling, out_dtype=torch.bfloat16)
        if out.dtype != query.dtype:
            out = out.to(query.dtype)
        return out, None
    if (query.shape[2] == 1 and key.shape[2] >= 1 and not dropout and query.is_cuda and query.shape[3] == 128
            and query.shape[1] % key.shape[1] == 0 and _mask_allows_everything(attention_mask)):
        q, k, v = query, key, value
        if q.dtype != torch.bfloat16:
            q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        out = decode_attention(q, k, v, scaling=scaling, out_dtype=torch.bfloat16)
        return (out if out.dtype == query.dtype else out.to(query.dtype)), None
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)


def register() -> str:
    """Register the function with transformers' AttentionInterface (idempotent); returns its name."""
    from transformers.modeling_utils import AttentionInterface

    AttentionInterface.register(ATTN_NAME, sdag_hip_attention_forward)
    return ATTN_NAME


@contextlib.contextmanager
def use_on_model(model):
    """Temporarily select the SDAG attention function on ``model`` (restores the previous one on exit)."""
    name = register()
    cfg = model.config
    prev = getattr(cfg, "_attn_implementation", None)
    cfg._attn_implementation = name
    try:
        yield
    finally:
        cfg._attn_implementation = prev
