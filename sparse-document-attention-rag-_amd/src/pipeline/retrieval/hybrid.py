"""Hybrid retrieval: reciprocal-rank fusion of a sparse and a dense ranking (drop-in for the reference's
``src/pipeline/retrieval/hybrid.py``).

Same public names, arguments and results as the reference (``hybrid.py:10-27`` k split, ``:30-104`` fusion of one
query, ``:107-163`` batch, ``:166-225`` ``HybridRetriever``).  The work is a few dictionary operations on at most
``max_k_needed`` strings per query, so it stays on the host; the dense half runs on the GPU through
``DenseRetriever``.  One thing is defined here that the reference leaves to chance: documents with EQUAL fused
scores come out in the order of their first appearance (sparse list first, then dense), where the reference's
order is that of iterating a ``set`` of strings (``hybrid.py:74,92``), which changes from process to process.
"""
from __future__ import annotations

import random
from typing import Dict, List, Sequence, Tuple

from sdag_amd.datamodels import RetrievalBatch

from src.pipeline.retrieval.retriever import Retriever


def split_k_between_sparse_and_dense(top_k: int, rng: random.Random) -> Tuple[int, int]:
    """(k_sparse, k_dense) with k_sparse + k_dense == top_k; an odd top_k gives the extra slot to one side by one
    draw of ``rng.random()`` (below 0.5: sparse), and an even one draws nothing (hybrid.py:22-27)."""
    half, odd = divmod(top_k, 2)
    if not odd:
        return half, half
    return (half + 1, half) if rng.random() < 0.5 else (half, half + 1)


def _key(doc_id, doc_text) -> str:
    # documents are identified by id; a missing id ("", "NA", None) falls back to the text (hybrid.py:61-64)
    return doc_text if doc_id is None or doc_id == "" or doc_id == "NA" else doc_id


def rrf_fuse_one_query(sparse_docs: List[str], sparse_ids: List[str], dense_docs: List[str], dense_ids: List[str],
                       k0: int = 60) -> Tuple[List[str], List[str], List[float]]:
    """score(doc) = sum over the rankings holding it of 1 / (k0 + rank), rank from 1; best first.

    A key that occurs twice inside one ranking counts with its LAST rank (the reference overwrites the dictionary
    entry, hybrid.py:67-72) and is represented by the text / id of its FIRST occurrence over sparse-then-dense
    (hybrid.py:79-89)."""
    rank_s: Dict[str, int] = {}
    rank_d: Dict[str, int] = {}
    rep: Dict[str, Tuple[str, str]] = {}           # insertion order = order of first appearance
    for ranks, docs, ids in ((rank_s, sparse_docs, sparse_ids), (rank_d, dense_docs, dense_ids)):
        for pos, (text, did) in enumerate(zip(docs, ids), start=1):
            k = _key(did, text)
            ranks[k] = pos
            rep.setdefault(k, (text, did))
    scored = []
    for k, (text, did) in rep.items():
        s = 0.0
        if k in rank_s:
            s += 1.0 / float(k0 + rank_s[k])
        if k in rank_d:
            s += 1.0 / float(k0 + rank_d[k])
        scored.append((s, text, did))
    scored.sort(key=lambda t: -t[0])               # stable: equal scores keep first-appearance order
    return [t[1] for t in scored], [t[2] for t in scored], [t[0] for t in scored]


def fuse_sparse_and_dense_batch(sparse_texts: List[List[str]], sparse_ids: List[List[str]], dense_texts: List[List[str]],
                                dense_ids: List[List[str]], top_k: int, seed: int, k0: int = 60
                                ) -> Tuple[List[List[str]], List[List[str]], List[List[float]]]:
    """Per query: split top_k between the two rankings (one ``random.Random(seed)`` for the whole batch, consumed
    in query order), fuse the two prefixes, keep the best top_k (hybrid.py:137-163)."""
    rng = random.Random(seed)
    out_t, out_i, out_s = [], [], []
    for s_t, s_i, d_t, d_i in zip(sparse_texts, sparse_ids, dense_texts, dense_ids):
        ks, kd = split_k_between_sparse_and_dense(top_k, rng)
        t, i, s = rrf_fuse_one_query(s_t[:ks], s_i[:ks], d_t[:kd], d_i[:kd], k0=k0)
        out_t.append(t[:top_k])
        out_i.append(i[:top_k])
        out_s.append(s[:top_k])
    return out_t, out_i, out_s


class HybridRetriever(Retriever):
    """Runs the dense and the sparse retriever on the same queries and fuses their rankings; the query embeddings
    of the result are the dense retriever's (hybrid.py:196-225)."""

    def __init__(self, dense_retriever: Retriever, sparse_retriever: Retriever, seed: int, k0: int = 60) -> None:
        self.dense = dense_retriever
        self.sparse = sparse_retriever
        self.seed = seed
        self.k0 = k0

    def retrieve_batch(self, queries: Sequence[str], max_k_needed: int, embed_batch_size: int) -> RetrievalBatch:
        dense = self.dense.retrieve_batch(queries, max_k_needed, embed_batch_size)
        sparse = self.sparse.retrieve_batch(queries, max_k_needed, embed_batch_size)
        texts, ids, scores = fuse_sparse_and_dense_batch(sparse.docs_texts_full, sparse.ids_full, dense.docs_texts_full,
                                                         dense.ids_full, top_k=max_k_needed, seed=self.seed, k0=self.k0)
        return RetrievalBatch(q_embs=dense.q_embs, docs_texts_full=texts, ids_full=ids, scores_full=scores)
