"""MI355X-native SDAG attention prefill and dense-retrieval search (host side).

Python stays thin: device memory, streams and process groups come from PyTorch-ROCm, the arithmetic runs in
the HIP kernels of libsdag_hip.so (``csrc/``), reached through the C ABI declared in ``include/sdag_hip.h``.
"""
from . import _lib
from ._lib import SdagError, build
from .attention import BoundAttention, BoundBatchAttention, decode_attention, sdag_attention, sdag_attention_batch
from .descriptor import BatchDescriptor, Descriptor, from_dense, from_spans, from_spans_many
from .index import FlatIPIndex, ShardedFlatIPIndex, merge_partial_topk, shard_bounds
from .ops import centroid_sims, centroid_sims_batch, doc_knn, gram

__all__ = [
    "SdagError", "build", "sdag_attention", "decode_attention", "BoundAttention", "BoundBatchAttention", "sdag_attention_batch", "BatchDescriptor", "Descriptor", "from_dense", "from_spans", "from_spans_many", "FlatIPIndex",
    "ShardedFlatIPIndex", "merge_partial_topk", "shard_bounds", "centroid_sims", "centroid_sims_batch", "doc_knn", "gram",
]
