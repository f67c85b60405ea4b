#!/usr/bin/env python3
"""Event-timed sweeps of sdag_attn_fwd over mask layouts and unit lengths (diagnostic, GPU only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sdag_amd

dev = torch.device("cuda:0")

def layout(sys_len, n_docs, doc_len, gap, qa_len):
    spans, pos = [], sys_len
    for _ in range(n_docs):
        spans.append((pos, pos + doc_len)); pos += doc_len + gap
    return spans, pos, pos + qa_len

def timeit(desc, L, Hq=32, Hkv=8, reps=20, nbuf=8, B=1):
    D = 128
    bufs = []
    for i in range(nbuf):
        q = torch.randn((B, L, Hq, D), device=dev).to(torch.bfloat16).transpose(1, 2)
        k = torch.randn((B, L, Hkv, D), device=dev).to(torch.bfloat16).transpose(1, 2)
        v = torch.randn((B, L, Hkv, D), device=dev).to(torch.bfloat16).transpose(1, 2)
        o = torch.empty((B, L, Hq, D), device=dev, dtype=torch.bfloat16)
        bufs.append(sdag_amd.BoundAttention(q, k, v, desc, o))
    for c in bufs: c()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        for c in bufs: c()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nbuf)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "one":
        L = 4096
        d = sdag_amd.from_spans(L, L, [], L, None, max_items_per_unit=100000)
        print(f"causal L={L}: {timeit(d, L, reps=2, nbuf=2):.1f} us", flush=True)
    if which == "batch":
        spans, qa, L = layout(60, 10, 300, 1, 30)
        for B in ((32,) if "quick" in sys.argv else (1, 2, 4, 8, 16, 32)):
            for T in ((64,) if "quick" in sys.argv else (11, 64)):
                d = sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=T)
                t = timeit(d, L, reps=5, nbuf=2, B=B)
                print(f"cfg3 B={B} T={T}: {t:.1f} us/launch = {t / B:.1f} us/query-layer", flush=True)
    if which == "vbatch":
        spans, qa, L = layout(60, 10, 300, 1, 30)
        for B in ((32,) if "quick" in sys.argv else (8, 32)):
            for T in ((64,) if "quick" in sys.argv else (11, 16, 32, 64)):
                descs = [sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=T) for _ in range(B)]
                if len(sys.argv) > 2 and sys.argv[2] == "same":
                    descs = [descs[0]] * B
                bd = sdag_amd.BatchDescriptor(descs)
                q = torch.randn((B, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
                k = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
                v = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
                o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
                call = sdag_amd.BoundBatchAttention(q, k, v, bd, o)
                call(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): call()
                e1.record(); torch.cuda.synchronize()
                t = e0.elapsed_time(e1) * 1e3 / 5
                print(f"cfg3 own-descriptor batch B={B} T={T}: {t:.1f} us/launch = {t / B:.1f} us/query-layer", flush=True)
    if which in ("all", "causal"):
        for L in (64, 512, 4096):
            for T in (100000, 8):
                d = sdag_amd.from_spans(L, L, [], L, None, max_items_per_unit=T)
                print(f"causal L={L} T={T}: units={d.n_units} items={d.n_items} slots={d.n_slots} -> {timeit(d, L):.1f} us", flush=True)
    if which in ("all", "cfg3"):
        spans, qa, L = layout(60, 10, 300, 1, 30)
        for T in (4, 6, 8, 11, 16, 32, 64):
            d = sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=T)
            print(f"cfg3 T={T}: units={d.n_units} items={d.n_items} slots={d.n_slots} -> {timeit(d, L):.1f} us", flush=True)
    if which == "cfg4batch":
        K = 20
        spans, qa, L = layout(60, K, 380, 1, 30)
        nbr = [[(i + 1) % K, (i + 2) % K] for i in range(K)]
        for B in (4, 8):
            descs = [sdag_amd.from_spans(L, 60, spans, qa, nbr, max_items_per_unit=64) for _ in range(B)]
            bd = sdag_amd.BatchDescriptor(descs)
            q = torch.randn((B, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            k = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            v = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
            call = sdag_amd.BoundBatchAttention(q, k, v, bd, o)
            call(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): call()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / 5
            fl = 4 * 128 * descs[0].allowed_pairs * 32
            print(f"cfg4 batch B={B}: {t:.1f} us/launch = {t / B:.1f} us/query-layer, {fl * B / t / 1e6:.0f} TFLOP/s, pairs={descs[0].allowed_pairs}", flush=True)
    if which in ("all", "cfg4"):
        K = 20
        spans, qa, L = layout(60, K, 380, 1, 30)
        nbr = [[(i + 1) % K, (i + 2) % K] for i in range(K)]
        for T in (8, 16, 32, 64):
            d = sdag_amd.from_spans(L, 60, spans, qa, nbr, max_items_per_unit=T)
            print(f"cfg4 T={T}: units={d.n_units} items={d.n_items} slots={d.n_slots} pairs={d.allowed_pairs} -> {timeit(d, L):.1f} us", flush=True)
    if which == "decode":
        # one new token per sequence against S cached keys; bytes = K and V of the cache read once
        for (B, Hq, Hkv, S) in ((1, 32, 8, 3100), (1, 32, 8, 8192), (8, 32, 8, 3100), (32, 32, 8, 3100), (1, 32, 8, 32768)):
            nbuf = 6
            bufs = []
            for i in range(nbuf):
                q = torch.randn((B, Hq, 1, 128), device=dev).to(torch.bfloat16)
                k = torch.randn((B, Hkv, S, 128), device=dev).to(torch.bfloat16)
                v = torch.randn((B, Hkv, S, 128), device=dev).to(torch.bfloat16)
                bufs.append((q, k, v))
            for q, k, v in bufs: sdag_amd.decode_attention(q, k, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for r in range(reps):
                for q, k, v in bufs: sdag_amd.decode_attention(q, k, v)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
            e0.record()
            for r in range(reps):
                for q, k, v in bufs: torch.nn.functional.scaled_dot_product_attention(q, k, v, enable_gqa=True)
            e1.record(); torch.cuda.synchronize()
            t_sdpa = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
            gb = 2 * B * Hkv * S * 128 * 2 / 1e9
            print(f"decode B={B} Hq={Hq} Hkv={Hkv} S={S}: {t:.1f} us = {gb / t * 1e6:.0f} GB/s  (torch sdpa {t_sdpa:.1f} us)", flush=True)
    if which == "cfg3b":
        # the bench's attention launch: 32 queries of cfg 3, one descriptor each, 4 rotating sets of tensors (8 GB)
        spans, qa, L = layout(60, 10, 300, 1, 30)
        B = 32
        Ts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64]
        NS = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
        tens = []
        for i in range(4):
            q = torch.randn((B, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            kb_ = 1 if "sharedkv" in sys.argv else B   # diagnostic: every sequence reads the same K/V memory
            k = torch.randn((kb_, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2).expand(B, 8, L, 128)
            v = torch.randn((kb_, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2).expand(B, 8, L, 128)
            o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
            tens.append((q, k, v, o))
        for T in Ts:
          for ns in NS:
            bd = sdag_amd.BatchDescriptor([sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=T) for _ in range(B)], n_streams=ns)
            calls = [sdag_amd.BoundBatchAttention(q, k, v, bd, o) for q, k, v, o in tens]
            for c in calls: c()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 8
            e0.record()
            for r in range(reps):
                for c in calls: c()
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / (reps * len(calls))
            print(f"cfg3 B=32 T={T} streams={ns}: {t:.1f} us/launch = {t / B:.2f} us/query-layer", flush=True)
        sys.exit(0)
        calls = []
        for i in range(4):
            q = torch.randn((B, L, 32, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            k = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            v = torch.randn((B, L, 8, 128), device=dev).to(torch.bfloat16).transpose(1, 2)
            o = torch.empty((B, L, 32, 128), device=dev, dtype=torch.bfloat16)
            calls.append(sdag_amd.BoundBatchAttention(q, k, v, bd, o))
        for c in calls: c()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 8
        e0.record()
        for r in range(reps):
            for c in calls: c()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e3 / (reps * len(calls))
        print(f"cfg3 B=32: {t:.1f} us/launch = {t / B:.2f} us/query-layer", flush=True)
