#!/usr/bin/env python3
"""Does the batched attention launch run faster on data that toggles fewer bits?  Same instructions, same addresses:
random, all-zero and mixed Q/K/V (diagnostic, GPU only).  A large gap means the chip is holding its clock down."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import torch
import sdag_amd

dev = torch.device("cuda:0")
B, L0 = 32, 0
spans, pos = [], 60
for _ in range(10):
    spans.append((pos, pos + 300)); pos += 301
L = pos + 30
descs = [sdag_amd.from_spans(L, 60, spans, pos, None, max_items_per_unit=64) for _ in range(B)]
bd = sdag_amd.BatchDescriptor(descs)

def make(kind, H):
    if kind == "rand":
        return torch.randn((B, L, H, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
    if kind == "zero":
        return torch.zeros((B, L, H, 128), device=dev, dtype=torch.bfloat16).transpose(1, 2)
    if kind == "ones":
        return torch.ones((B, L, H, 128), device=dev, dtype=torch.bfloat16).transpose(1, 2)
    raise ValueError(kind)

def run(qk, kk, vk, seconds=2.0):
    q, k, v = make(qk, 32), make(kk, 8), make(vk, 8)
    o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
    call = sdag_amd.BoundBatchAttention(q, k, v, bd, o)
    call(); torch.cuda.synchronize()
    n = int(seconds / 0.8e-3)
    for _ in range(n): call()          # reach the clock the chip settles at
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    print(f"Q {qk:5s} K {kk:5s} V {vk:5s}: {e0.elapsed_time(e1) * 1e3 / 200:.1f} us/launch", flush=True)

for combo in (("rand", "rand", "rand"), ("zero", "zero", "zero"), ("rand", "rand", "zero"), ("zero", "zero", "rand"),
              ("ones", "ones", "ones"), ("rand", "rand", "rand")):
    run(*combo)
