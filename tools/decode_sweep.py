#!/usr/bin/env python3
"""Event-timed decode attention over batch sizes (diagnostic, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import torch, sdag_amd
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3100
for B in (1, 4, 8, 32, 128):
    sets = []
    for i in range(4):
        q = torch.randn((B, 32, 1, 128), device=dev).to(torch.bfloat16)
        k = torch.randn((B, 8, S, 128), device=dev).to(torch.bfloat16)
        v = torch.randn((B, 8, S, 128), device=dev).to(torch.bfloat16)
        sets.append((q, k, v))
    for q, k, v in sets: sdag_amd.decode_attention(q, k, v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        for q, k, v in sets: sdag_amd.decode_attention(q, k, v)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e3 / (reps * len(sets))
    byts = B * 8 * S * 128 * 2 * 2
    print(f"S={S} B={B}: {t:.1f} us  {byts / t / 1e6:.2f} TB/s", flush=True)
