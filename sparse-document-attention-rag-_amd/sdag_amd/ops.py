"""Small similarity ops of the pipeline on the GPU (document kNN, Near/Far centroid scores)."""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from . import _lib


def _dev_f32(x, device) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.detach().to(device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device)


def gram(emb, device="cuda:0") -> torch.Tensor:
    """E E^T for E [n, d]; replaces np.matmul(emb, emb.T) (src/pipeline/sparse_attention_RAG/SDAG.py:50)."""
    e = _dev_f32(emb, device)
    n, d = e.shape
    sims = torch.empty((n, n), dtype=torch.float32, device=e.device)
    with torch.cuda.device(e.device):
        _lib.check(_lib.lib().sdag_gram(e.data_ptr(), n, d, sims.data_ptr(), _lib.stream_ptr(e.device)))
    return sims


def doc_knn(emb, nonempty_idx: Sequence[int], n_docs: int, k_neighbors: int, device="cuda:0") -> List[List[int]]:
    """Neighbour lists from the embeddings of the non-empty documents (SDAG.py:38-65).

    Similarities come from the GPU; the per-row ordering is numpy's ``argsort(-row)`` as in the reference
    (SDAG.py:55), self excluded, first k kept, ids mapped back to positions in the full list.
    """
    out: List[List[int]] = [[] for _ in range(n_docs)]
    if k_neighbors <= 0 or n_docs == 0 or len(nonempty_idx) <= 1:
        return out
    sims = gram(emb, device).cpu().numpy()
    for r, i_full in enumerate(nonempty_idx):
        order = np.argsort(-sims[r])
        knn = [int(nonempty_idx[j]) for j in order if j != r][:k_neighbors]
        out[i_full] = knn
    return out


def centroid_sims(retrieved_emb, candidate_emb, device="cuda:0") -> torch.Tensor:
    """<candidate, mean(retrieved)> per candidate (src/pipeline/attack/malicious_selection.py:89-95,117-123)."""
    r = _dev_f32(retrieved_emb, device)
    c = _dev_f32(candidate_emb, device)
    nr, d = r.shape
    nc = c.shape[0]
    centroid = torch.empty(d, dtype=torch.float32, device=r.device)
    sims = torch.empty(nc, dtype=torch.float32, device=r.device)
    with torch.cuda.device(r.device):
        _lib.check(_lib.lib().sdag_centroid_sims(r.data_ptr(), nr, c.data_ptr() if nc else None, nc, d,
                                                 centroid.data_ptr(), sims.data_ptr() if nc else None,
                                                 _lib.stream_ptr(r.device)))
    return sims


def centroid_sims_batch(retrieved_embs: Sequence, candidate_embs: Sequence, device="cuda:0") -> List[np.ndarray]:
    """``centroid_sims`` for every query of a batch in ONE launch and one copy back to the host: query b has
    ``retrieved_embs[b]`` [nr_b >= 1, d] and ``candidate_embs[b]`` [nc_b, d]; returns one fp32 array [nc_b] per query
    (the same bits as per-query calls)."""
    nb = len(retrieved_embs)
    if nb != len(candidate_embs):
        raise ValueError("one candidate set per query expected")
    if nb == 0:
        return []
    rs = [_dev_f32(r, device) for r in retrieved_embs]
    cs = [_dev_f32(c, device) for c in candidate_embs]
    d = rs[0].shape[1]
    if any(r.dim() != 2 or r.shape[0] < 1 or r.shape[1] != d for r in rs) or any(c.dim() != 2 or c.shape[1] != d for c in cs):
        raise ValueError("embeddings must be [n, d] with the same d and at least one retrieved row per query")
    dev = rs[0].device
    r_off = np.zeros(nb + 1, dtype=np.int32)
    c_off = np.zeros(nb + 1, dtype=np.int32)
    r_off[1:] = np.cumsum([r.shape[0] for r in rs])
    c_off[1:] = np.cumsum([c.shape[0] for c in cs])
    R = torch.cat(rs, dim=0)
    C = torch.cat(cs, dim=0) if int(c_off[-1]) else torch.zeros((1, d), dtype=torch.float32, device=dev)
    offs = torch.from_numpy(np.concatenate([r_off, c_off])).to(dev)
    centroid = torch.empty((nb, d), dtype=torch.float32, device=dev)
    sims = torch.empty(max(int(c_off[-1]), 1), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().sdag_centroid_sims_batch(R.data_ptr(), offs.data_ptr(), C.data_ptr(), offs[nb + 1:].data_ptr(), nb, d,
                                                       centroid.data_ptr(), sims.data_ptr(), _lib.stream_ptr(dev)))
    host = sims.cpu().numpy()
    return [host[c_off[b]:c_off[b + 1]].copy() for b in range(nb)]
