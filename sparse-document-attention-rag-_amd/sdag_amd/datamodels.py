"""Boundary types of the retrieval path when the reference tree is not importable.

``RetrievalBatch`` has the fields of src/pipeline/models/datamodels.py:29-41 (q_embs, docs_texts_full, ids_full,
scores_full; outer list per query, inner lists ranked best first).  When the reference package is on the
path its own class is used instead, so objects cross the boundary unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List

try:  # the caller's own type, if the reference pipeline is importable
    from src.pipeline.models.datamodels import RetrievalBatch  # type: ignore
except Exception:  # pragma: no cover - exercised when running standalone

    @dataclass
    class RetrievalBatch:
        q_embs: List[Any]
        docs_texts_full: List[List[str]]
        ids_full: List[List[str]]
        scores_full: List[List[float]]
