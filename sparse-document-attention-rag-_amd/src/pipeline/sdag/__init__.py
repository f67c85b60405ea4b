"""Alias of ``src.pipeline.sparse_attention_RAG.SDAG`` under the name the reference's README uses for it."""
from src.pipeline.sparse_attention_RAG.SDAG import (  # noqa: F401
    build_blocked_causal_mask_full,
    build_rag_chat_and_spans,
    compute_doc_knn_for_docs,
    generate_with_custom_mask,
    run_rag_with_doc_isolation,
    run_rag_with_doc_isolation_batch,
)
