"""Exact inner-product index resident in HBM: the object the reference calls ``index``.

The reference does ``scores, indices = index.search(q.astype(np.float32), top_k)`` on whatever
``faiss.read_index`` returned (src/pipeline/retrieval/dense.py:36,82-84).  ``FlatIPIndex`` offers that duck
type (``search``, ``ntotal``, ``d``, ``add``) with the corpus kept as one fp32 [N, d] array on the GPU and the
search done by ``sdag_dense_topk``.  ``ShardedFlatIPIndex`` row-shards the corpus over the ranks of a
``torch.distributed`` process group: every rank searches its shard with global ids, one all-gather moves the
packed partial lists, and every rank merges them with the same (score desc, id asc) rule.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

MAX_K = 32  # entries per query one kernel call returns; larger k is answered page by page


def _paged(first_page, next_page, k: int):
    """Best k per query for any k: the first 32 from ``first_page(32)``, every further page of up to 32 from
    ``next_page(kk, after_s, after_i)`` with the previous page's last entry as the cursor (the pages are disjoint and
    consecutive in the (score desc, id asc) order, so their concatenation is the top k)."""
    s, i = first_page(min(k, MAX_K))
    pages_s, pages_i = [s], [i]
    done = s.shape[1]
    while done < k:
        kk = min(MAX_K, k - done)
        s, i = next_page(kk, pages_s[-1][:, -1].contiguous(), pages_i[-1][:, -1].contiguous())
        pages_s.append(s)
        pages_i.append(i)
        done += kk
    if len(pages_s) == 1:
        return pages_s[0], pages_i[0]
    return torch.cat(pages_s, dim=1), torch.cat(pages_i, dim=1)


def _as_f32_2d(x, name: str) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float32)
    if not a.flags.writeable:  # e.g. a slice of a read-only memory map: torch.from_numpy wants a writable array
        a = a.copy()
    if a.ndim != 2:
        raise ValueError(f"{name} must be 2-D")
    return a


class FlatIPIndex:
    """faiss.IndexFlatIP-shaped exact search on one GPU."""

    def __init__(self, d: int, device="cuda:0", id_base: int = 0, two_stage: bool = True):
        """``two_stage``: search through a bf16 copy of the corpus (built on first use, +50 % memory) followed by
        exact rescoring of the candidates -- same results bit for bit, half the bytes scanned; needs d % 64 == 0,
        otherwise (or with ``two_stage=False``) every search scans the fp32 corpus."""
        if d <= 0 or d % 32:
            raise ValueError("d must be a positive multiple of 32")
        self.two_stage = bool(two_stage) and d % 64 == 0
        self._c16: Optional[torch.Tensor] = None       # [N, d] bf16 copy of _corpus
        self._c16_of: Optional[torch.Tensor] = None    # the tensor _c16 was made from
        self._cmax: Optional[torch.Tensor] = None      # [1] fp32: bound of the largest row norm
        self.d = int(d)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("FlatIPIndex lives on the GPU (no CPU fallback)")
        self.id_base = int(id_base)
        self._corpus: Optional[torch.Tensor] = None  # [N, d] fp32 (a view of _buf while a reservation is being filled)
        self._buf: Optional[torch.Tensor] = None     # rows reserved ahead of add() (see reserve)
        self._ws: Optional[torch.Tensor] = None

    @classmethod
    def from_tensor(cls, corpus: torch.Tensor, id_base: int = 0, two_stage: bool = True) -> "FlatIPIndex":
        """Index over rows that already live on the GPU (fp32 [N, d], contiguous): no copy is made, so several
        indexes can cover row ranges of one resident corpus (shards)."""
        if not corpus.is_cuda or corpus.dtype != torch.float32 or corpus.dim() != 2 or not corpus.is_contiguous():
            raise ValueError("corpus must be a contiguous fp32 [N, d] tensor on the GPU")
        idx = cls(corpus.shape[1], device=corpus.device, id_base=id_base, two_stage=two_stage)
        idx._corpus = corpus
        return idx

    # ---- faiss-like surface
    @property
    def ntotal(self) -> int:
        return 0 if self._corpus is None else int(self._corpus.shape[0])

    def reserve(self, n_rows: int) -> None:
        """Allocate room for ``n_rows`` rows once, so that a corpus loaded slice by slice is written in place instead
        of being re-concatenated (and held twice) by every add()."""
        n_rows = int(n_rows)
        if n_rows <= (0 if self._buf is None else self._buf.shape[0]):
            return
        buf = torch.empty((n_rows, self.d), dtype=torch.float32, device=self.device)
        n = self.ntotal
        if n:
            buf[:n] = self._corpus
        self._buf = buf
        self._corpus = buf[:n] if n else None

    def add(self, x) -> None:
        """Append rows (numpy on the host or a torch tensor on any device)."""
        if isinstance(x, torch.Tensor):
            t = x.detach().to(self.device, dtype=torch.float32)
            if t.dim() != 2:
                raise ValueError("x must be 2-D")
        else:
            t = torch.from_numpy(_as_f32_2d(x, "x")).to(self.device)
        if t.shape[1] != self.d:
            raise ValueError(f"x has dimension {t.shape[1]}, index has {self.d}")
        n = self.ntotal
        if self._buf is not None and n + t.shape[0] <= self._buf.shape[0]:
            self._buf[n:n + t.shape[0]] = t
            self._corpus = self._buf[:n + t.shape[0]]
            return
        self._buf = None
        t = t.contiguous()
        self._corpus = t if self._corpus is None else torch.cat([self._corpus, t], dim=0)

    def reset(self) -> None:
        self._corpus = self._buf = None
        self._c16 = self._c16_of = self._cmax = None

    def search(self, q, k: int) -> Tuple[np.ndarray, np.ndarray]:
        """(scores f32 [nq, k], ids i64 [nq, k]) as numpy, best first -- the faiss return order."""
        qa = _as_f32_2d(q, "q")
        s, i = self.search_device(torch.from_numpy(qa).to(self.device), k)
        return s.cpu().numpy(), i.cpu().numpy()

    # ---- device-resident variant (no host round trip)
    def search_device(self, q: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Any k >= 1, like faiss: up to 32 entries per query come from one search, the rest page by page from the
        exact scan with a cursor (``search_after``), one pass over the fp32 corpus per 32 further entries."""
        q, k = self._check_query(q, k)
        return _paged(lambda kk: self._search_page(q, kk), lambda kk, a_s, a_i: self.search_after(q, kk, a_s, a_i), k)

    def _check_query(self, q: torch.Tensor, k: int):
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"q must be [nq, {self.d}]")
        k = int(k)
        if k <= 0:
            raise ValueError("k must be positive")
        return q.to(self.device, dtype=torch.float32).contiguous(), k

    def _search_page(self, q: torch.Tensor, kk: int, after=None) -> Tuple[torch.Tensor, torch.Tensor]:
        nq = q.shape[0]
        out_s = torch.empty((nq, kk), dtype=torch.float32, device=self.device)
        out_i = torch.empty((nq, kk), dtype=torch.int64, device=self.device)
        N = self.ntotal
        lib = _lib.lib()
        if self.two_stage and N > 0 and after is None:
            return self._search_two_stage(q, nq, kk, out_s, out_i)
        need = lib.sdag_dense_workspace_bytes(nq, N, self.d, kk)
        if need < 0:
            raise _lib.SdagError(-1, "bad workspace query")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=self.device)
        cptr = self._corpus.data_ptr() if N else None
        with torch.cuda.device(self.device):
            if after is None:
                rc = lib.sdag_dense_topk(q.data_ptr(), nq, cptr, N, self.d, kk, out_s.data_ptr(), out_i.data_ptr(),
                                         self.id_base, self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr(self.device))
            else:
                rc = lib.sdag_dense_topk_after(q.data_ptr(), nq, cptr, N, self.d, kk, after[0].data_ptr(), after[1].data_ptr(),
                                               out_s.data_ptr(), out_i.data_ptr(), self.id_base, self._ws.data_ptr(),
                                               self._ws.numel(), _lib.stream_ptr(self.device))
        _lib.check(rc)
        return out_s, out_i

    def search_after(self, q: torch.Tensor, k: int, after_s: torch.Tensor, after_i: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """The best k <= 32 rows per query among those that come strictly after the entry (after_s[j], after_i[j]) in
        the order of the results (ids as returned, i.e. global): the next page of a search that ended there."""
        q, k = self._check_query(q, k)
        if k > MAX_K:
            raise ValueError(f"a page holds at most {MAX_K} entries")
        nq = q.shape[0]
        a_s = after_s.to(self.device, dtype=torch.float32).contiguous()
        a_i = after_i.to(self.device, dtype=torch.int64).contiguous()
        if a_s.shape != (nq,) or a_i.shape != (nq,):
            raise ValueError("one cursor entry per query")
        return self._search_page(q, k, (a_s, a_i))


    def prepare(self) -> None:
        """Build what the first search would otherwise build: the bf16 copy of the corpus for the two-stage search
        (index construction work, like ``add``; a no-op when it is up to date or the index scans fp32)."""
        N = self.ntotal
        if not self.two_stage or N == 0 or (self._c16 is not None and self._c16_of is self._corpus):
            return
        with torch.cuda.device(self.device):
            self._c16 = torch.empty((N, self.d), dtype=torch.bfloat16, device=self.device)
            self._cmax = torch.zeros(1, dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().sdag_corpus_to_bf16(self._corpus.data_ptr(), N, self.d, self._c16.data_ptr(),
                                                      self._cmax.data_ptr(), _lib.stream_ptr(self.device)))
            self._c16_of = self._corpus

    def _search_two_stage(self, q, nq, kk, out_s, out_i):
        lib = _lib.lib()
        N = self.ntotal
        self.prepare()
        with torch.cuda.device(self.device):
            need = lib.sdag_dense_topk2_workspace_bytes(nq, N, self.d, kk)
            if need < 0:
                raise _lib.SdagError(-1, "bad workspace query")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(max(int(need), 256), dtype=torch.uint8, device=self.device)
            rc = lib.sdag_dense_topk2(q.data_ptr(), nq, self._corpus.data_ptr(), self._c16.data_ptr(), self._cmax.data_ptr(),
                                      N, self.d, kk, out_s.data_ptr(), out_i.data_ptr(), self.id_base,
                                      self._ws.data_ptr(), self._ws.numel(), _lib.stream_ptr(self.device))
        _lib.check(rc)
        self._last = (nq, N, kk)
        return out_s, out_i

    def last_search_fell_back(self) -> bool:
        """Diagnostics: did the last two-stage search (the last of its passes over the corpus) have to run the exact scan?"""
        if not self.two_stage or getattr(self, "_last", None) is None:
            return False
        nq, N, kk = self._last
        off = int(_lib.lib().sdag_dense_topk2_flag_offset(nq, N, self.d, kk))
        return bool(self._ws[off:off + 4].view(torch.int32).item())


def merge_partial_topk(parts_s: torch.Tensor, parts_i: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """[n_parts, nq, k] partial lists (device) -> best k per query by (score desc, id asc)."""
    n_parts, nq, kk = parts_s.shape
    if kk != k or parts_i.shape != parts_s.shape:
        raise ValueError("partial lists must be [n_parts, nq, k]")
    parts_s = parts_s.contiguous().float()
    parts_i = parts_i.contiguous().to(torch.int64)
    out_s = torch.empty((nq, k), dtype=torch.float32, device=parts_s.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=parts_s.device)
    with torch.cuda.device(parts_s.device):
        rc = _lib.lib().sdag_topk_merge(parts_s.data_ptr(), parts_i.data_ptr(), n_parts, nq, k, out_s.data_ptr(),
                                        out_i.data_ptr(), _lib.stream_ptr(parts_s.device))
    _lib.check(rc)
    return out_s, out_i


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row range of ``rank``: shards of ceil(N / world) rows, the last ones possibly short or empty."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    return lo, min(n_total, lo + per)


def pack_partial(scores: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """One int64 buffer [nq, k, 2] per rank (score bits, id) so a single all-gather moves both."""
    return torch.stack([scores.contiguous().view(torch.int32).to(torch.int64), ids], dim=-1).contiguous()


def unpack_partial(buf: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return buf[..., 0].to(torch.int32).view(torch.float32), buf[..., 1].contiguous()


def gather_partials(mine: torch.Tensor, group=None) -> torch.Tensor:
    """The one collective of the sharded search: all-gather of the packed [nq, k, 2] buffers -> [world, nq, k, 2]."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if mine.is_cuda and dist.get_backend(group) == "gloo":
        # a CPU-only backend (rehearsals on one GPU): stage through the host explicitly
        host = mine.contiguous().cpu()
        flat = torch.empty((world * host.shape[0],) + tuple(host.shape[1:]), dtype=host.dtype)
        dist.all_gather_into_tensor(flat, host, group=group)
        return flat.view((world,) + tuple(mine.shape)).to(mine.device)
    flat = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
    dist.all_gather_into_tensor(flat, mine.contiguous(), group=group)
    return flat.view((world,) + tuple(mine.shape))


class ShardedFlatIPIndex:
    """Row-sharded exact search: local top-k with global ids -> one all-gather -> identical merge on every rank."""

    def __init__(self, d: int, n_total: int, device, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total = int(n_total)
        self.lo, self.hi = shard_bounds(self.n_total, self.world, self.rank)
        self.local = FlatIPIndex(d, device=device, id_base=self.lo)
        self.d = d

    @property
    def ntotal(self) -> int:
        return self.n_total

    def add_local(self, x) -> None:
        """Rows [lo, hi) of the global corpus, in order."""
        self.local.add(x)
        if self.local.ntotal > self.hi - self.lo:
            raise ValueError("more rows than this rank's shard holds")

    def search_device(self, q: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """Any k >= 1.  Every page of up to 32 entries is one local search + one all-gather + one merge; the cursor of
        the pages after the first is the GLOBAL last entry of the page before (the same on every rank)."""
        def merged(local):
            ps, pi = unpack_partial(gather_partials(pack_partial(*local), self.group))
            return merge_partial_topk(ps, pi, local[0].shape[1])

        return _paged(lambda kk: merged(self.local.search_device(q, kk)),
                      lambda kk, a_s, a_i: merged(self.local.search_after(q, kk, a_s, a_i)), int(k))

    def search(self, q, k: int) -> Tuple[np.ndarray, np.ndarray]:
        qa = _as_f32_2d(q, "q")
        s, i = self.search_device(torch.from_numpy(qa).to(self.local.device), k)
        return s.cpu().numpy(), i.cpu().numpy()
