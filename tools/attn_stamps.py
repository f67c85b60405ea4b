#!/usr/bin/env python3
"""Per-phase cycle shares of one wave of the attention main loop (needs `make -C csrc stamps`; GPU only)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SDAG_HIP_LIB"] = os.path.join(ROOT, "sparse-document-attention-rag-_amd", "sdag_amd", "libsdag_hip_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import torch, sdag_amd
from sdag_amd import _lib
dev = torch.device("cuda:0")
names = ["top(loads,init)", "mask+QK^T", "Q load,vtag", "softmax", "PV", "stage_store", "barrier", "unit end"]
def run(desc, L, label):
    q = torch.randn((1, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
    k = torch.randn((1, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn((1, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
    for _ in range(3):
        sdag_amd.sdag_attention(q, k, v, desc)
    out = (ctypes.c_ulonglong * 16)()
    _lib.check(_lib.lib().sdag_debug_stamps(out))
    n = out[15]; tot = sum(out[i] for i in range(8))
    print(f"{label}: {n} tiles, {tot / n:.0f} cycles/tile: " + ", ".join(f"{names[i]} {out[i] / n:.0f}" for i in range(8)), flush=True)
L = 4096
run(sdag_amd.from_spans(L, L, [], L, None, max_items_per_unit=100000), L, "causal4096 (chip full)")
L = 2048
run(sdag_amd.from_dense(torch.ones(L, L, dtype=torch.bool)[:64].repeat(32, 1) if False else (torch.arange(L)[:, None] >= 0) & (torch.arange(L)[None, :] < L) & (torch.arange(L)[:, None] < 32), max_items_per_unit=100000), L, "one row block x 32 tiles (chip idle)")
spans, pos = [], 60
for _ in range(10):
    spans.append((pos, pos + 300)); pos += 301
run(sdag_amd.from_spans(pos + 30, 60, spans, pos, None, max_items_per_unit=64), pos + 30, "cfg3 T=64 (QA unit)")
run(sdag_amd.from_spans(pos + 30, 60, spans, pos, None), pos + 30, "cfg3 default units (stream 0)")
