"""Block-sparse attention descriptors: the replacement for the reference's dense L x L mask.

The reference builds ``torch.bool [L, L]`` with Python row loops
(src/pipeline/sparse_attention_RAG/SDAG.py:104-127).  A descriptor holds the same relation as work items
over 32-row x 64-key tiles plus bit masks for partially allowed tiles; it is built on the host by
``sdag_desc_build`` (C++) in time proportional to L x (documents + tiles), uploaded once per prompt and
shared by every layer and head.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib

H_L, H_NRB, H_NKT, H_NUNITS, H_NITEMS, H_NMASKS, H_NSPLIT, H_NUNIFORM = 2, 3, 4, 5, 6, 7, 8, 9
H_NSLOTS = 17
H_NSTREAMS = 20

_tls = threading.local()  # per-thread build buffer: descriptors may be built on a helper thread while the GPU runs


class Descriptor:
    """Host copy (numpy int32) + lazily created device copies of one descriptor."""

    def __init__(self, words: np.ndarray):
        self.host = np.ascontiguousarray(words, dtype=np.int32)
        self._dev: Dict[torch.device, torch.Tensor] = {}

    # ---- header fields
    @property
    def L(self) -> int:
        return int(self.host[H_L])

    @property
    def n_units(self) -> int:
        return int(self.host[H_NUNITS])

    @property
    def n_items(self) -> int:
        return int(self.host[H_NITEMS])

    @property
    def n_partial_tiles(self) -> int:
        return int(self.host[H_NMASKS])

    @property
    def n_uniform_rows(self) -> int:
        return int(self.host[H_NUNIFORM])

    @property
    def n_slots(self) -> int:
        return int(self.host[H_NSLOTS])

    @property
    def n_streams(self) -> int:
        return int(self.host[H_NSTREAMS])

    @property
    def allowed_pairs(self) -> int:
        pairs = ctypes.c_int64(0)
        _lib.check(_lib.lib().sdag_desc_stats(self.host.ctypes.data, ctypes.byref(pairs), None, None, None, None))
        return int(pairs.value)

    @property
    def host_ptr(self) -> int:
        return self.host.ctypes.data

    def device(self, device) -> torch.Tensor:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        t = self._dev.get(device)
        if t is None:
            t = torch.from_numpy(self.host).to(device)
            self._dev[device] = t
        return t

    def workspace_bytes(self, B: int, Hq: int, Hkv: int) -> int:
        n = _lib.lib().sdag_attn_workspace_bytes(self.host_ptr, B, Hq, Hkv)
        if n < 0:
            raise _lib.SdagError(-1, "bad workspace query")
        return int(n)

    def materialize(self, device) -> torch.Tensor:
        """Dense ``torch.bool [L, L]`` on ``device`` (what build_blocked_causal_mask_full returns)."""
        dev = self.device(device)
        out = torch.empty((self.L, self.L), dtype=torch.uint8, device=dev.device)
        if self.L:
            with torch.cuda.device(dev.device):
                _lib.check(_lib.lib().sdag_mask_materialize(self.host_ptr, dev.data_ptr(), out.data_ptr(),
                                                            _lib.stream_ptr(dev.device)))
        return out.view(torch.bool)


def _run_builder(call, out: Optional[np.ndarray] = None) -> np.ndarray:
    """Run a C builder into this thread's scratch buffer and return a copy of the words it wrote, or build straight
    into ``out`` (a caller-owned int32 array, e.g. a view of pinned memory) and return the view of what was written."""
    need = ctypes.c_int64(0)
    if out is not None:
        if out.dtype != np.int32 or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError("out must be a writable contiguous int32 array")
        _lib.check(call(out.ctypes.data, out.size, ctypes.byref(need)))
        return out[: int(need.value)]
    scratch = getattr(_tls, "scratch", None)
    if scratch is None:
        scratch = _tls.scratch = np.empty(1 << 16, dtype=np.int32)
    rc = call(scratch.ctypes.data, scratch.size, ctypes.byref(need))
    if rc == _lib.SDAG_ERR_CAPACITY:
        scratch = _tls.scratch = np.empty(int(need.value) * 2, dtype=np.int32)
        rc = call(scratch.ctypes.data, scratch.size, ctypes.byref(need))
    _lib.check(rc)
    return scratch[: int(need.value)].copy()


def from_spans(
    base_seq_len: int,
    sys_user_len: int,
    doc_token_spans: Sequence[Tuple[int, int]],
    qa_start: int,
    doc_neighbors: Optional[Sequence[Sequence[int]]] = None,
    max_items_per_unit: int = 0,
    n_streams: int = 0,
) -> Descriptor:
    """Descriptor of the SDAG mask; argument meaning as in build_blocked_causal_mask_full (SDAG.py:68-76).

    ``doc_neighbors`` is ignored unless it has one entry per document (SDAG.py:111).  Values the reference
    would index out of range with (negative positions, spans past L) raise ``ValueError``.
    ``n_streams`` is the number of work streams the units are dealt to (0 = 64, see ``streams_for``).
    """
    n_docs = len(doc_token_spans)
    spans = np.asarray([int(x) for se in doc_token_spans for x in se], dtype=np.int32).reshape(-1)
    use_nbr = doc_neighbors is not None and len(doc_neighbors) == n_docs
    nbr_off = nbr_idx = None
    if use_nbr:
        lens = [len(n) for n in doc_neighbors]
        nbr_off = np.zeros(n_docs + 1, dtype=np.int32)
        nbr_off[1:] = np.cumsum(lens)
        flat = [int(x) for n in doc_neighbors for x in n]
        # ids that do not fit int32 are out of range by definition; keep them out of range
        flat = [x if -(2 ** 31) <= x < 2 ** 31 else -1 for x in flat]
        nbr_idx = np.asarray(flat, dtype=np.int32)
    lib = _lib.lib()

    def call(out_ptr, cap, need):
        return lib.sdag_desc_build(int(base_seq_len), int(sys_user_len), n_docs, spans.ctypes.data if n_docs else None,
                                   int(qa_start), nbr_off.ctypes.data if use_nbr else None,
                                   nbr_idx.ctypes.data if use_nbr and nbr_idx.size else None,
                                   int(max_items_per_unit), int(n_streams), out_ptr, cap, need)

    try:
        return Descriptor(_run_builder(call))
    except _lib.SdagError as e:
        if e.code == _lib.SDAG_ERR_ARG:
            raise ValueError(str(e)) from None
        raise


_pool = None
_pool_lock = threading.Lock()


def from_spans_many(layouts, max_items_per_unit: int = 0, n_streams: int = 0, threads: int = 0):
    """``from_spans`` for several prompts at once -- ``layouts`` is a sequence of
    ``(base_seq_len, sys_user_len, doc_token_spans, qa_start, doc_neighbors)`` -- on a small pool of host threads (the
    builder is a C call that releases the interpreter lock and keeps its scratch per thread): a batch of 32 long
    prompts is built in the time of four or five.  Returns the descriptors in the order of ``layouts``; an error in
    any of them is raised as from ``from_spans``."""
    global _pool
    layouts = list(layouts)
    if len(layouts) <= 1 or threads == 1:
        return [from_spans(*lay, max_items_per_unit=max_items_per_unit, n_streams=n_streams) for lay in layouts]
    with _pool_lock:
        if _pool is None:
            from concurrent.futures import ThreadPoolExecutor
            import os

            _pool = ThreadPoolExecutor(max_workers=threads or max(2, min(8, os.cpu_count() or 2)), thread_name_prefix="sdag-desc")
    return list(_pool.map(lambda lay: from_spans(*lay, max_items_per_unit=max_items_per_unit, n_streams=n_streams), layouts))


def streams_for(Hq: int, Hkv: int, compute_units: int = 256, batch: int = 1) -> int:
    """Streams that fill ``compute_units`` at two workgroups each: one workgroup serves a stream for up to four
    query heads that share a key/value head."""
    slots = Hkv * ((Hq // Hkv + 3) // 4) * max(1, batch)
    return max(1, -(-2 * compute_units // slots))


def from_dense(mask, max_items_per_unit: int = 0, n_streams: int = 0) -> Descriptor:
    """Descriptor of an arbitrary boolean [L, L] mask (numpy or torch, any device)."""
    if isinstance(mask, torch.Tensor):
        mask = mask.detach().to("cpu").numpy()
    m = np.ascontiguousarray(np.asarray(mask) != 0, dtype=np.uint8)
    if m.ndim != 2 or m.shape[0] != m.shape[1]:
        raise ValueError("mask must be square [L, L]")
    lib = _lib.lib()

    def call(out_ptr, cap, need):
        return lib.sdag_desc_from_dense(m.ctypes.data, m.shape[0], int(max_items_per_unit), int(n_streams), out_ptr, cap, need)

    return Descriptor(_run_builder(call))


class BatchDescriptor:
    """Descriptors of several sequences packed for one launch (``sdag_attn_fwd_batch``)."""

    def __init__(self, descs: Sequence[Descriptor], n_streams: int = 0, out: Optional[np.ndarray] = None):
        """``out``: build the table straight into this int32 array (pinned host memory for an asynchronous upload;
        SdagError with code SDAG_ERR_CAPACITY if it is too small) instead of into memory of its own."""
        if not descs:
            raise ValueError("empty batch")
        self.descs = list(descs)
        ptrs = (ctypes.c_void_p * len(descs))(*[d.host_ptr for d in descs])
        lib = _lib.lib()

        def call(out_ptr, cap, need):
            return lib.sdag_batch_build(ptrs, len(descs), int(n_streams), out_ptr, cap, need)

        self.host = _run_builder(call, out)
        self._dev: Dict[torch.device, torch.Tensor] = {}
        self.B = len(descs)
        self.Lmax = max(d.L for d in descs)

    @property
    def host_ptr(self) -> int:
        return self.host.ctypes.data

    def device(self, device) -> torch.Tensor:
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        t = self._dev.get(device)
        if t is None:
            t = torch.from_numpy(self.host).to(device)
            self._dev[device] = t
        return t

    def bind_device_copy(self, tensor: torch.Tensor) -> None:
        """Adopt ``tensor`` (int32, on a GPU, holding a copy of ``host`` that the caller uploaded, e.g. asynchronously
        from pinned memory on a side stream) as this table's device copy."""
        if tensor.dtype != torch.int32 or not tensor.is_cuda or tensor.numel() < self.host.size:
            raise ValueError("device copy must be an int32 GPU tensor of at least host.size words")
        self._dev[tensor.device] = tensor

    def workspace_bytes(self, Hq: int, Hkv: int) -> int:
        n = _lib.lib().sdag_attn_batch_workspace_bytes(self.host_ptr, Hq, Hkv)
        if n < 0:
            raise _lib.SdagError(-1, "bad workspace query")
        return int(n)
