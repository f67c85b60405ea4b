"""CPU oracle for the SDAG attention + dense-retrieval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package may import this
directory: it is the checker for ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Every function restates, in plain
numpy / torch-CPU / C, what the reference computes on this path and cites the
reference file:line it follows.

Parity status: the reference ships no tests or golden vectors of its own.  The
mask / doc-kNN / centroid / hit-materialisation / attention restatements are
pinned by outputs of the reference's own Python, generated in the development
container by ``tests/golden/make_golden.py`` and committed under
``tests/golden/``.  The inner-product search arithmetic lives in faiss (absent
from the reference tree and from this image), so the top-k restatement defines
its own arithmetic (k-ordered fp32 FMA chain, ties by lower id): that one
function is "parity unpinned" against faiss itself.
"""
