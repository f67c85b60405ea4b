"""Loading a flat inner-product index without faiss.

The reference does ``faiss.read_index(index_path)`` (src/pipeline/retrieval/dense.py:36).  faiss is not part of
this build; for the one index type whose search this package implements exactly -- a flat inner-product index --
the file is read directly and its vectors are placed in HBM.  The layout below restates the published faiss
serialisation of IndexFlat (fourcc "IxFI", header d:int32 ntotal:int64 two int64 fillers is_trained:uint8
metric_type:int32, then the float count as uint64 and the raw little-endian fp32 rows).  No faiss-written file
exists in the reference tree to check this reader against, so it is unpinned; ``.npy`` files are also accepted.
"""
from __future__ import annotations

import os
import struct

import numpy as np

METRIC_INNER_PRODUCT = 0


def write_flat_ip(path: str, x: np.ndarray) -> None:
    x = np.ascontiguousarray(x, dtype="<f4")
    n, d = x.shape
    with open(path, "wb") as f:
        f.write(b"IxFI")
        f.write(struct.pack("<iqqqBi", d, n, 1 << 20, 1 << 20, 1, METRIC_INNER_PRODUCT))
        f.write(struct.pack("<Q", n * d))
        f.write(x.tobytes())


def read_flat_vectors(path: str) -> np.ndarray:
    """fp32 [N, d] rows of a flat IP index file or a .npy file (memory-mapped)."""
    if not os.path.exists(path):
        raise FileNotFoundError(f"FAISS index not found at {path}")
    if path.endswith(".npy"):
        a = np.load(path, mmap_mode="r", allow_pickle=False)
        if a.ndim != 2 or a.dtype != np.float32:
            raise ValueError(".npy index must be float32 [N, d]")
        return a
    with open(path, "rb") as f:
        fourcc = f.read(4)
        if fourcc in (b"IxF2", b"IxFl"):
            raise NotImplementedError("only inner-product flat indexes (IxFI) are supported; this file is an L2 / legacy flat index")
        if fourcc != b"IxFI":
            raise NotImplementedError(f"unsupported index type {fourcc!r}: only a flat inner-product index can be searched exactly")
        d, n, _, _, _, metric = struct.unpack("<iqqqBi", f.read(4 + 8 * 3 + 1 + 4))
        if metric != METRIC_INNER_PRODUCT:
            raise NotImplementedError("flat index with a non inner-product metric")
        (count,) = struct.unpack("<Q", f.read(8))
        if count != n * d:
            raise ValueError(f"corrupt flat index: {count} floats for {n} x {d}")
        offset = f.tell()
    return np.memmap(path, dtype="<f4", mode="r", offset=offset, shape=(n, d))
