"""Dense retrieval on MI355X behind the reference's module surface (src/pipeline/retrieval/dense.py).

Same importable names and meanings: ``load_faiss_index_and_meta``, ``embed_queries``, ``search_index``,
``materialize_dense_hits``, ``DenseRetriever``.  The difference is below ``index.search``: the index is a
``sdag_amd.FlatIPIndex`` whose corpus lives in HBM and whose search is the fused HIP scan + top-k, with a
defined arithmetic (dimension-ordered fp32 FMA chain, ties by lower id).
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from sdag_amd.datamodels import RetrievalBatch
from sdag_amd.index import FlatIPIndex
from sdag_amd.index_io import read_flat_vectors
from src.pipeline.retrieval.retriever import Retriever

try:  # frozen at import like the reference does (dense.py:10): later config overrides do not reach it
    from src.pipeline.config import RANKER_MODEL_NAME
except Exception:  # pragma: no cover - standalone use
    RANKER_MODEL_NAME = "intfloat/e5-large-v2"

try:
    from src.pipeline.config import DEVICE as _CFG_DEVICE
except Exception:  # pragma: no cover
    _CFG_DEVICE = "cuda"


def load_faiss_index_and_meta(index_path: str, meta_path: str, device: str | None = None) -> Tuple[Any, List[Dict[str, Any]]]:
    """Load the corpus vectors into HBM and the JSONL metadata (dense.py:15-41).

    Raises FileNotFoundError with the reference's messages when a file is missing (dense.py:31-34).
    """
    if not os.path.exists(index_path):
        raise FileNotFoundError(f"FAISS index not found at {index_path}")
    if not os.path.exists(meta_path):
        raise FileNotFoundError(f"Metadata jsonl not found at {meta_path}")
    vecs = read_flat_vectors(index_path)
    dev = device or ("cuda:0" if str(_CFG_DEVICE).startswith("cuda") else str(_CFG_DEVICE))
    index = FlatIPIndex(vecs.shape[1], device=dev)
    index.reserve(vecs.shape[0])  # N is known from the file header: one allocation, slices written in place
    step = 1 << 18
    for lo in range(0, vecs.shape[0], step):  # stream the (possibly memory-mapped) file in slices
        index.add(np.array(vecs[lo:lo + step], dtype=np.float32, order="C"))  # a writable copy of the mapped slice
    meta: List[Dict[str, Any]] = []
    with open(meta_path, "r", encoding="utf-8") as f:
        for line in f:
            meta.append(json.loads(line))
    return index, meta


def embed_queries(ranker, queries: List[str], batch_size: int) -> np.ndarray:
    """Normalised query embeddings, fp32 [nq, d]; "query: " prefix for e5 rankers (dense.py:44-65)."""
    use_prefix = "e5" in RANKER_MODEL_NAME.lower()
    texts = [("query: " + q) if use_prefix else q for q in queries]
    chunks = []
    for lo in range(0, len(texts), batch_size):
        emb = ranker.encode(texts[lo:lo + batch_size], convert_to_tensor=True, batch_size=batch_size,
                            normalize_embeddings=True)
        chunks.append(emb.cpu().numpy())
    return np.vstack(chunks)


def search_index(index: Any, query_embeddings: np.ndarray, top_k: int) -> Tuple[np.ndarray, np.ndarray]:
    """(indices [nq, k], scores [nq, k]) -- note the order, as in dense.py:82-84."""
    scores, indices = index.search(query_embeddings.astype(np.float32), top_k)
    return indices, scores


def materialize_dense_hits(indices, scores, meta: Sequence[Dict[str, Any]]):
    """ids -> (texts, ids, scores) lists; ids outside the metadata give ("", "NA", score) (dense.py:109-130)."""
    n_meta = len(meta)
    texts_b: List[List[str]] = []
    ids_b: List[List[str]] = []
    scores_b: List[List[float]] = []
    for row_i, row_s in zip(indices, scores):
        texts, ids, scs = [], [], []
        for idx, sc in zip(row_i, row_s):
            idx = int(idx)
            entry = meta[idx] if 0 <= idx < n_meta else None
            texts.append(entry.get("text", "") if entry is not None else "")
            ids.append(f"{entry.get('id', 'NA')}" if entry is not None else "NA")
            scs.append(float(sc))
        texts_b.append(texts)
        ids_b.append(ids)
        scores_b.append(scs)
    return texts_b, ids_b, scores_b


class DenseRetriever(Retriever):
    """Encode queries, search the HBM-resident index, materialise hits (dense.py:133-178)."""

    def __init__(self, ranker, index: Any, meta: List[Dict[str, Any]]) -> None:
        self.ranker = ranker
        self.index = index
        self.meta = meta

    def retrieve_batch(self, queries: Sequence[str], max_k_needed: int, embed_batch_size: int) -> RetrievalBatch:
        q_embs = embed_queries(self.ranker, list(queries), batch_size=embed_batch_size)
        indices, scores = search_index(self.index, q_embs, top_k=max_k_needed)
        docs_texts, doc_ids, doc_scores = materialize_dense_hits(indices, scores, self.meta)
        return RetrievalBatch(q_embs=list(q_embs), docs_texts_full=docs_texts, ids_full=doc_ids, scores_full=doc_scores)
