#!/usr/bin/env python3
"""Event-timed runs of the exact inner-product search (diagnostic, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import torch, sdag_amd
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
C = torch.empty((N, d), device=dev)
for lo in range(0, N, 1 << 18):
    x = torch.randn((min(N, lo + (1 << 18)) - lo, d), device=dev); C[lo:lo + x.shape[0]] = x / x.norm(dim=1, keepdim=True)
idx = sdag_amd.FlatIPIndex(d, device=dev, two_stage="exact" not in sys.argv); idx._corpus = C
for nq in ((256,) if "only256" in sys.argv else (128,) if "only128" in sys.argv else (32, 64, 100, 128, 256, 1024) if "wide" in sys.argv else (32, 64)):
    Q = torch.randn((nq, d), device=dev); Q = Q / Q.norm(dim=1, keepdim=True)
    idx.search_device(Q, K); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): idx.search_device(Q, K)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print("two-stage" if idx.two_stage else "exact", "fell back" if idx.last_search_fell_back() else "", end=" ")
    print(f"N={N} d={d} nq={nq} k={K}: {t:.3f} ms/search  {N * d * 4 / t / 1e9 * (nq / 32 if nq > 32 else 1):.2f} TB/s-equivalent scan, {nq / t * 1e3:.0f} q/s", flush=True)
