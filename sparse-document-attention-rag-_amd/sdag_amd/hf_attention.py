"""The SDAG kernel as a transformers attention function.

transformers calls ``attention_interface(module, query[B,Hq,L,D], key[B,Hkv,S,D], value[B,Hkv,S,D],
attention_mask, dropout=, scaling=, **kwargs) -> (attn_output[B,L,Hq,D], attn_weights | None)`` from every
attention layer (transformers/models/llama/modeling_llama.py:264-277).  The reference reaches it with a dense
additive [1,1,L,L] mask (src/pipeline/sparse_attention_RAG/SDAG.py:167-176); here the mask travels as a
block-sparse descriptor set for the duration of the prefill, and no dense mask exists.

The single-token decode steps (SDAG.py:191-211: one new position against the whole cache, no mask from the
reference) run the key/value-streaming kernel ``sdag_decode_attn``.  While a descriptor is active every call with more
than one query row must fit it (same batch, same length) or raises: the reference masks every prefill
(SDAG.py:167-176), so none may fall through to plain causal attention.  Calls outside a descriptor (an ordinary causal
prefill, a padded batch) are handed to transformers' stock SDPA attention unchanged.

Masks: the function is registered with transformers' mask interface under SDPA's mask builder, which hands the
layers ``None`` whenever nothing but causality has to be enforced and decides that ONCE per forward pass; the layers
here only test ``attention_mask is None`` and never read a mask tensor back (no device synchronisation per layer).
"""
from __future__ import annotations

import contextlib
import threading
from typing import Optional

import torch

from .attention import decode_attention, sdag_attention, sdag_attention_batch
from .descriptor import BatchDescriptor, Descriptor

ATTN_NAME = "sdag_hip"
_state = threading.local()


def current_descriptor() -> Optional[Descriptor]:
    return getattr(_state, "desc", None)


@contextlib.contextmanager
def active_descriptor(desc):
    """Make ``desc`` (a ``Descriptor``, or a ``BatchDescriptor`` for a right-padded batch of prompts) the mask of
    every prefill-shaped attention call on this thread inside the block."""
    prev = getattr(_state, "desc", None)
    _state.desc = desc
    try:
        yield desc
    finally:
        _state.desc = prev


@contextlib.contextmanager
def active_padding(valid_len: Optional[torch.Tensor], pad_end: int, n_rows: Optional[torch.Tensor] = None):
    """How the key/value cache of the single-token steps inside the block (on this thread) is laid out.

    ``valid_len`` (int32 [B] on the model's device) with ``pad_end``: right-padded prompts decoded together -- cache
    rows [valid_len[b], pad_end) of sequence b are padding and are left out.  ``n_rows`` (one int32 on the device): the
    cache is preallocated and only its first ``n_rows`` rows hold keys; the kernel reads the count, the caller keeps it
    up to date (steps replayed from a captured graph).  With ``n_rows`` the caller vouches that this is ALL the masking
    a step needs, so a mask tensor transformers built for the preallocated cache is not looked at."""
    prev = getattr(_state, "padding", None)
    _state.padding = (valid_len, int(pad_end), n_rows)
    try:
        yield
    finally:
        _state.padding = prev


def _on_gpu(t: torch.Tensor) -> bool:
    return t.is_cuda


def _mask_allows_everything(attention_mask) -> bool:
    """A decode step of an unpadded sequence gets no mask from transformers (see ``register``).  A mask tensor means
    that something is forbidden (padding): such a step goes to SDPA with that mask; its contents are not inspected,
    which would cost a device synchronisation per layer and step."""
    return attention_mask is None


def _additive_mask(desc: Descriptor, S: int, dtype, device) -> torch.Tensor:
    """[S, S] additive mask (0 / finfo.min, SDAG.py:167-169) of one descriptor inside a padded length S; rows and
    columns past the descriptor's own length allow nothing."""
    add = torch.full((S, S), torch.finfo(dtype).min, dtype=dtype, device=device)
    if desc.L:
        add[:desc.L, :desc.L].masked_fill_(desc.materialize(device), 0)
    return add


def _dense_mask_fallback(module, query, key, value, desc, dropout, scaling, **kwargs):
    """Prefill under a descriptor (or a batch of them) on a model the kernel does not cover (head_dim above 128,
    query heads not a multiple of key/value heads): SDPA under the dense mask expanded from the descriptor on the GPU."""
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    if not query.is_cuda:
        raise RuntimeError("the SDAG attention path runs on the GPU only (no CPU fallback)")
    S = query.shape[2]
    if isinstance(desc, BatchDescriptor):
        add = torch.stack([_additive_mask(d, S, query.dtype, query.device) for d in desc.descs])[:, None]
    else:
        add = _additive_mask(desc, S, query.dtype, query.device)[None, None]
    return sdpa_attention_forward(module, query, key, value, add, dropout=dropout, scaling=scaling, **kwargs)


def _masked_prefill(module, query, key, value, desc, dropout, scaling, **kwargs):
    """A prefill-shaped call (more than one query row) while a descriptor is active.  The reference applies its mask to
    every prefill (SDAG.py:167-176), so a call the descriptor does not fit is an error: it must never run as plain
    causal attention, which would silently drop the document isolation."""
    B, Hq, Lq, D = query.shape
    if Lq != key.shape[2]:
        raise RuntimeError(f"SDAG prefill: {Lq} query rows against {key.shape[2]} keys while a mask descriptor is active "
                           "(a prefill over an existing cache cannot be document-isolated by this descriptor)")
    if isinstance(desc, BatchDescriptor):
        if B != desc.B or Lq < desc.Lmax:
            raise RuntimeError(f"SDAG prefill: batch of {B} x {Lq} tokens does not match the active batch descriptor "
                               f"({desc.B} prompts, longest {desc.Lmax} tokens)")
    elif Lq != desc.L:
        raise RuntimeError(f"SDAG prefill: {Lq} tokens do not match the active descriptor (L = {desc.L})")
    if dropout:
        raise NotImplementedError("attention dropout is not supported by the SDAG kernel (inference only)")
    if not (D <= 128 and Hq % key.shape[1] == 0):  # (a CPU tensor: the kernel wrappers raise)
        return _dense_mask_fallback(module, query, key, value, desc, dropout, scaling, **kwargs)
    q, k, v = query, key, value
    if q.dtype != torch.bfloat16:
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    if D < 128:
        # narrower heads (64, 96, ...) run the 128-wide kernel on zero-padded copies: the padding adds nothing to a
        # score and yields zero output columns, which are dropped; the softmax scale stays that of the true width
        if scaling is None:
            scaling = D ** -0.5
        q, k, v = (torch.nn.functional.pad(t, (0, 128 - D)) for t in (q, k, v))
    if isinstance(desc, BatchDescriptor):
        out = sdag_attention_batch(q, k, v, desc, scaling=scaling, out_dtype=torch.bfloat16)
    else:
        out = sdag_attention(q, k, v, desc, scaling=scaling, out_dtype=torch.bfloat16)
    if D < 128:
        out = out[..., :D]
    return (out if out.dtype == query.dtype else out.to(query.dtype)), None


def sdag_hip_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0,
                               scaling: Optional[float] = None, **kwargs):
    desc = current_descriptor()
    if desc is not None and query.shape[2] > 1:
        return _masked_prefill(module, query, key, value, desc, dropout, scaling, **kwargs)
    padding = getattr(_state, "padding", None)
    mask_ok = _mask_allows_everything(attention_mask) or (padding is not None and padding[2] is not None)
    if (query.shape[2] == 1 and key.shape[2] >= 1 and not dropout and _on_gpu(query) and query.shape[3] == 128
            and query.shape[1] % key.shape[1] == 0 and mask_ok):
        q, k, v = query, key, value
        if q.dtype != torch.bfloat16:
            q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
        if padding is not None and (padding[0] is None or padding[0].numel() == query.shape[0]):
            out = decode_attention(q, k, v, scaling=scaling, out_dtype=torch.bfloat16, valid_len=padding[0], pad_end=padding[1],
                                   n_rows=padding[2])
        else:
            out = decode_attention(q, k, v, scaling=scaling, out_dtype=torch.bfloat16)
        return (out if out.dtype == query.dtype else out.to(query.dtype)), None
    if padding is not None and query.shape[2] == 1:
        raise RuntimeError("batched decoding of padded prompts needs the GPU decode kernel (head_dim 128, no dropout, no mask)")
    from transformers.integrations.sdpa_attention import sdpa_attention_forward

    return sdpa_attention_forward(module, query, key, value, attention_mask, dropout=dropout, scaling=scaling, **kwargs)


def register() -> str:
    """Register the function with transformers' AttentionInterface (idempotent); returns its name."""
    from transformers.modeling_utils import AttentionInterface

    AttentionInterface.register(ATTN_NAME, sdag_hip_attention_forward)
    try:  # masks as for SDPA: None unless padding has to be enforced, decided once per forward pass
        from transformers.masking_utils import AttentionMaskInterface, sdpa_mask

        AttentionMaskInterface.register(ATTN_NAME, sdpa_mask)
    except ImportError:  # a transformers without the mask interface: it builds no mask for unknown functions
        pass
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
