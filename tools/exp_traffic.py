#!/usr/bin/env python3
"""Which memory stream costs the batched attention launch its time?  cfg 3 batch of 32, the tensors of all sequences
made to alias one sequence's memory, one stream at a time (diagnostic, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import torch, sdag_amd
dev = torch.device("cuda:0")
spans, pos = [], 60
for _ in range(10):
    spans.append((pos, pos + 300)); pos += 301
qa, L = pos, pos + 30
B, Hq, Hkv, D = 32, 32, 8, 128
descs = [sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=64) for _ in range(B)]
bd = sdag_amd.BatchDescriptor(descs)
def mk(h, b=B): return torch.randn((b, L, h, D), device=dev).to(torch.bfloat16).transpose(1, 2)
sets = [(mk(Hq), mk(Hkv), mk(Hkv), torch.empty((B, L, Hq, D), device=dev, dtype=torch.bfloat16)) for _ in range(3)]
def run(label, fq, fk, fo):
    calls = []
    for q, k, v, o in sets:
        q2 = q[:1].expand(B, -1, -1, -1) if fq else q
        k2 = k[:1].expand(B, -1, -1, -1) if fk else k
        v2 = v[:1].expand(B, -1, -1, -1) if fk else v
        o2 = o[:1].expand(B, -1, -1, -1) if fo else o
        calls.append(sdag_amd.BoundBatchAttention(q2, k2, v2, bd, o2))
    for c in calls: c()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        for c in calls: c()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:28s} {e0.elapsed_time(e1) * 1e3 / 30:8.1f} us/launch", flush=True)
run("all distinct", 0, 0, 0)
run("Q aliased", 1, 0, 0)
run("K/V aliased", 0, 1, 0)
run("O aliased", 0, 0, 1)
run("Q and O aliased", 1, 0, 1)
run("everything aliased", 1, 1, 1)
run("all distinct (again)", 0, 0, 0)
