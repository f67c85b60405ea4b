"""Retriever plug-in interface (same contract as the reference's src/pipeline/retrieval/retriever.py:9-19)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Sequence

from sdag_amd.datamodels import RetrievalBatch


class Retriever(ABC):
    """Implementations return RetrievalBatch(q_embs, docs_texts_full, ids_full, scores_full)."""

    @abstractmethod
    def retrieve_batch(self, queries: Sequence[str], max_k_needed: int, embed_batch_size: int) -> RetrievalBatch:
        raise NotImplementedError
