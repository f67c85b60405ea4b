#!/usr/bin/env python3
"""Batched cfg-3 attention launch timed with 32 / 64 / 128 work streams per key/value head slot (256 / 512 / 1024
workgroups): how much two co-resident waves per SIMD overlap (diagnostic, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import torch, sdag_amd
dev = torch.device("cuda:0")
spans, pos = [], 60
for _ in range(10):
    spans.append((pos, pos + 300)); pos += 301
qa, L, B = pos, pos + 30, 32
descs = [sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=64) for _ in range(B)]
q = torch.randn((B, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
k = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
v = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
for ns in (int(a) for a in (sys.argv[1:] or ("32", "64", "128"))):
    bd = sdag_amd.BatchDescriptor(descs, n_streams=ns)
    call = sdag_amd.BoundBatchAttention(q, k, v, bd, o)
    call(); call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / 10
    nsteps = int(bd.host[13]) if hasattr(bd, "host") else -1
    print(f"streams {ns}: {t:.1f} us/launch", flush=True)
