"""Near / Far selection of injected documents on MI355X behind the reference's module surface
(src/pipeline/attack/malicious_selection.py).

Same names, arguments, return values and use of the ``random`` module as the reference.  What changes: the
centroid of the retrieved documents' embeddings and the candidates' similarities to it
(malicious_selection.py:89-95,117-123) are computed on the GPU, for ALL queries of a batch in one launch
(``sdag_centroid_sims_batch``) with one copy back; the argmax / argmin / argsort over at most a handful of
candidates stays numpy's, as in the reference.
"""
from __future__ import annotations

import random
from typing import List, Optional

import numpy as np
import torch

import sdag_amd

try:  # frozen at import like the reference does (malicious_selection.py:9)
    from src.pipeline.config import RANKER_MODEL_NAME
except Exception:  # pragma: no cover - standalone use
    RANKER_MODEL_NAME = "intfloat/e5-large-v2"


def _encode_on_device(ranker, texts: List[str]) -> torch.Tensor:
    inputs = ["passage: " + t if "e5" in RANKER_MODEL_NAME.lower() else t for t in texts]
    emb = ranker.encode(inputs, convert_to_tensor=True, normalize_embeddings=True)
    return emb.float()


def encode_texts_with_ranker(ranker, texts: List[str]) -> np.ndarray:
    """Normalised fp32 embeddings [len(texts), d] as numpy; "passage: " prefix for e5 rankers
    (malicious_selection.py:12-34)."""
    if not texts:
        return np.zeros((0, ranker.get_sentence_embedding_dimension()), dtype=np.float32)
    return _encode_on_device(ranker, texts).cpu().numpy().astype(np.float32)


def _target_n(candidate_docs: List[str], max_docs: Optional[int]) -> int:
    if max_docs is None or max_docs < 0 or max_docs >= len(candidate_docs):
        return len(candidate_docs)
    return max_docs


def _random_pick(candidate_docs: List[str], target_n: int) -> List[str]:
    # one document: random.choice; several: random.sample (malicious_selection.py:79,107)
    return [random.choice(candidate_docs)] if target_n == 1 else random.sample(candidate_docs, target_n)


def _pick_from_sims(candidate_docs: List[str], sims: np.ndarray, strategy: str, target_n: int) -> List[str]:
    if target_n == 1:  # first occurrence of the extreme (malicious_selection.py:97-104)
        idx = int(np.argmax(sims)) if strategy == "closest_to_centroid" else int(np.argmin(sims))
        return [candidate_docs[idx]]
    order = np.argsort(-sims) if strategy == "closest_to_centroid" else np.argsort(sims)
    return [candidate_docs[int(i)] for i in order[:target_n]]


def select_malicious_docs_for_batch(ranker, retrieved_docs_batch_full: List[List[str]],
                                    malicious_doc_groups_batch: List[List[str]], strategy: str,
                                    max_docs: int) -> List[List[str]]:
    """Per query: up to ``max_docs`` of its candidates by ``strategy`` (malicious_selection.py:140-180).

    Queries are visited in order, so every use of ``random`` (the "random" strategy and the reference's fallbacks
    for a query without retrieved text) happens at the same point of the sequence as in the reference; the queries
    that need similarities are then scored together in one launch.
    """
    chosen: List[Optional[List[str]]] = []
    jobs = []  # (position in the batch, retrieved embeddings, candidate embeddings, target_n)
    centroid_strategy = strategy in ("closest_to_centroid", "furthest_from_centroid")
    for retrieved_docs, candidates in zip(retrieved_docs_batch_full, malicious_doc_groups_batch):
        if not candidates:
            chosen.append([])
            continue
        target_n = _target_n(candidates, max_docs)
        if strategy == "random":
            chosen.append(_random_pick(candidates, target_n))
            continue
        nonempty = [d for d in retrieved_docs if d and d.strip()]
        if not nonempty:
            chosen.append(_random_pick(candidates, target_n))
            continue
        r_emb = _encode_on_device(ranker, nonempty)
        c_emb = _encode_on_device(ranker, candidates) if r_emb.shape[0] else None
        if r_emb.shape[0] == 0 or c_emb.shape[0] == 0:
            chosen.append(_random_pick(candidates, target_n))
            continue
        if not centroid_strategy:  # unknown name: randrange for one document, sample for several (:102,130)
            chosen.append([candidates[random.randrange(len(candidates))]] if target_n == 1
                          else random.sample(candidates, target_n))
            continue
        jobs.append((len(chosen), r_emb, c_emb, target_n))
        chosen.append(None)
    if jobs:
        dev = next((j[1].device for j in jobs if j[1].is_cuda), torch.device("cuda:0"))
        sims = sdag_amd.centroid_sims_batch([j[1] for j in jobs], [j[2] for j in jobs], device=dev)
        for (pos, _, _, target_n), s in zip(jobs, sims):
            chosen[pos] = _pick_from_sims(malicious_doc_groups_batch[pos], s, strategy, target_n)
    return chosen  # type: ignore[return-value]


def _select_malicious_docs_for_query(ranker, retrieved_docs: List[str], candidate_docs: List[str], strategy: str,
                                     max_docs: int) -> List[str]:
    """One query (malicious_selection.py:37-137): the batch routine with a batch of one."""
    return select_malicious_docs_for_batch(ranker, [retrieved_docs], [candidate_docs], strategy, max_docs)[0]
