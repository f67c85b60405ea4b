"""CPU restatement of the SDAG mask, masked attention, doc-kNN and centroid selection.

Test infrastructure only (see ``oracle/__init__.py``).  File:line citations are
relative to the reference tree.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------- mask
def mask_from_spans(
    base_seq_len: int,
    sys_user_len: int,
    doc_token_spans: Sequence[Tuple[int, int]],
    qa_start: int,
    doc_neighbors: Optional[Sequence[Sequence[int]]] = None,
) -> np.ndarray:
    """Boolean [L, L] "row may attend to column" mask.

    Follows src/pipeline/sparse_attention_RAG/SDAG.py:104-127:
      * rows < sys_user_len are causal (:107-108);
      * a row inside document d sees [0, sys_user_len) and [d_start, row]
        (:115-116) plus the whole span of every neighbour of d, also when that
        span lies to the right of the row (:118-122); neighbour ids outside
        [0, num_docs) are skipped (:119-120) and neighbours only apply when the
        list has one entry per document (:111);
      * rows >= qa_start are causal (:124-125);
      * every write only sets True, so overlapping regions take the union and a
        row covered by no region stays all False (:105).

    Domain: 0 <= sys_user_len <= L and 0 <= start, end <= L for every span
    (start > end is an empty document); qa_start >= 0.  The reference indexes a
    torch tensor with these values, so anything outside that domain either raises
    IndexError there or wraps around through negative indexing; this restatement
    raises ValueError instead.
    """
    L = int(base_seq_len)
    if not (0 <= sys_user_len <= L):
        raise ValueError("sys_user_len outside [0, L]")
    if qa_start < 0:
        raise ValueError("qa_start negative")
    for s, e in doc_token_spans:
        if s < 0 or e < 0 or s > L or e > L:
            raise ValueError("document span outside [0, L]")
    m = np.zeros((L, L), dtype=bool)
    if L == 0:
        return m
    tril = np.tril(np.ones((L, L), dtype=bool))
    m[:sys_user_len] = tril[:sys_user_len]
    n_docs = len(doc_token_spans)
    use_nbr = doc_neighbors is not None and len(doc_neighbors) == n_docs
    for d, (s, e) in enumerate(doc_token_spans):
        if e <= s:
            continue
        m[s:e, :sys_user_len] = True
        m[s:e, s:e] |= tril[s:e, s:e]
        if use_nbr:
            for nb in doc_neighbors[d]:
                if nb < 0 or nb >= n_docs:
                    continue
                ns, ne = doc_token_spans[nb]
                if ne > ns:
                    m[s:e, ns:ne] = True
    if qa_start < L:
        m[qa_start:] |= tril[qa_start:]
    return m


def additive_mask(mask: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """bool [L, L] -> additive [1, 1, L, L]: 0 where allowed, finfo(dtype).min elsewhere.

    Follows SDAG.py:164-173.
    """
    neg = torch.finfo(dtype).min
    a = torch.zeros(mask.shape, dtype=dtype)
    a = a.masked_fill(~mask, neg)
    return a.unsqueeze(0).unsqueeze(1)


def masked_attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    mask: torch.Tensor,
    scale: float,
    compute_dtype: torch.dtype = torch.float32,
) -> torch.Tensor:
    """softmax(q k^T * scale + M) v with GQA key/value repetition.

    q [B, Hq, L, D]; k, v [B, Hkv, S, D]; mask bool [L, S]; returns [B, L, Hq, D]
    in ``compute_dtype``.  Follows the eager attention the reference reaches through
    ``model(input_ids, attention_mask=[1,1,L,L])`` (SDAG.py:176):
    transformers/models/llama/modeling_llama.py:191-213 -- repeat_kv, matmul * scaling,
    add the additive mask, softmax in fp32, matmul with V, transpose(1, 2).

    With ``compute_dtype=float32`` a row whose mask is all False has every score equal to
    finfo.min after the add (|score| << ulp(finfo.min)), so its softmax is uniform over
    all S keys and its output is mean(V) -- the behaviour of separator tokens in the
    reference.
    """
    B, Hq, L, D = q.shape
    Hkv = k.shape[1]
    G = Hq // Hkv
    add = additive_mask(mask, compute_dtype)[0, 0]
    out = torch.empty(B, L, Hq, D, dtype=compute_dtype)
    for b in range(B):
        for h in range(Hq):
            qh = q[b, h].to(compute_dtype)
            kh = k[b, h // G].to(compute_dtype)
            vh = v[b, h // G].to(compute_dtype)
            w = torch.matmul(qh, kh.t()) * scale
            w = w + add
            w = torch.softmax(w, dim=-1, dtype=torch.float32).to(compute_dtype)
            out[b, :, h, :] = torch.matmul(w, vh)
    return out


# ------------------------------------------------------------------------ doc kNN
def doc_knn_from_embeddings(emb: np.ndarray, nonempty_idx: Sequence[int], n_docs: int, k_neighbors: int) -> List[List[int]]:
    """Neighbour lists from the embeddings of the non-empty documents.

    Follows SDAG.py:38-65: Gram matrix emb @ emb.T (:50), per row argsort of the negated
    similarities (:55), self skipped (:58-59), first k kept (:61-62), ids mapped back to
    positions in the full document list (:60); documents that were empty keep [] (:51).
    """
    out: List[List[int]] = [[] for _ in range(n_docs)]
    if k_neighbors <= 0 or n_docs == 0 or len(nonempty_idx) <= 1:
        return out
    emb = np.asarray(emb, dtype=np.float32)
    sims = np.matmul(emb, emb.T)
    for r, i_full in enumerate(nonempty_idx):
        order = np.argsort(-sims[r])
        knn: List[int] = []
        for j in order:
            if j == r:
                continue
            knn.append(int(nonempty_idx[j]))
            if len(knn) >= k_neighbors:
                break
        out[i_full] = knn
    return out


# --------------------------------------------------------------- centroid selection
def centroid_order(retrieved_emb: np.ndarray, candidate_emb: np.ndarray, strategy: str) -> np.ndarray:
    """Candidate indices ordered by the Near / Far centroid rule.

    Follows src/pipeline/attack/malicious_selection.py:117-128: centroid = mean of the
    retrieved embeddings, sims = cand @ centroid^T, argsort(-sims) for
    "closest_to_centroid", argsort(sims) for "furthest_from_centroid".  The one-document
    case (:89-100) takes argmax / argmin, i.e. the first occurrence of the extreme.
    """
    centroid = np.asarray(retrieved_emb, np.float32).mean(axis=0, keepdims=True)
    sims = np.matmul(np.asarray(candidate_emb, np.float32), centroid.T).reshape(-1)
    if strategy == "closest_to_centroid":
        return np.argsort(-sims)
    if strategy == "furthest_from_centroid":
        return np.argsort(sims)
    raise ValueError(strategy)


def centroid_pick_one(retrieved_emb: np.ndarray, candidate_emb: np.ndarray, strategy: str) -> int:
    centroid = np.asarray(retrieved_emb, np.float32).mean(axis=0, keepdims=True)
    sims = np.matmul(np.asarray(candidate_emb, np.float32), centroid.T).reshape(-1)
    if strategy == "closest_to_centroid":
        return int(np.argmax(sims))
    if strategy == "furthest_from_centroid":
        return int(np.argmin(sims))
    raise ValueError(strategy)


# ---------------------------------------------------------------- hit materialisation
def materialize_hits(indices, scores, meta):
    """ids -> (texts, ids, scores) lists with the "NA" fallback.

    Follows src/pipeline/retrieval/dense.py:109-130: an index outside [0, len(meta)) gives
    ("", "NA", score); a valid one gives (meta.text or "", str(meta.id or "NA"), score).
    """
    texts, ids, scs = [], [], []
    for row_i, row_s in zip(indices, scores):
        t, i_, s_ = [], [], []
        for idx, sc in zip(row_i, row_s):
            idx = int(idx)
            if 0 <= idx < len(meta):
                t.append(meta[idx].get("text", ""))
                i_.append(f"{meta[idx].get('id', 'NA')}")
            else:
                t.append("")
                i_.append("NA")
            s_.append(float(sc))
        texts.append(t)
        ids.append(i_)
        scs.append(s_)
    return texts, ids, scs
