#!/usr/bin/env python3
"""Throughput of the SDAG attention prefill + dense-retrieval search on MI355X.

One step = one batch of 32 queries through the hot path, inputs resident in HBM:
  * retrieval: exact inner-product top-10 of Q[32, 1024] against a 2,000,000 x 1024 fp32 corpus (bf16 candidate scan +
    exact fp32 rescoring: the same ids and scores as a scan of the fp32 rows; --exact-scan for that scan)
    (one ``sdag_dense_topk`` call: scan kernel + merge kernel);
  * SDAG: for each of the 32 queries, the masked attention of all 32 layers of a Llama-3-8B-shaped prompt
    (32 query heads, 8 KV heads, head_dim 128, bf16; system prefix 60 tokens, 10 documents x 300 tokens with
    1-token separators, 30 question tokens -> L = 3100, no neighbours).  Default ``--mode batched``: one
    ``sdag_attn_fwd_batch`` launch per layer over the 32 queries, every query with its own descriptor and its own
    Q/K/V/O memory; ``--mode per_query``: one ``sdag_attn_fwd`` call per (query, layer) with B = 1, the way the
    reference calls its model once per query (its cost is also reported under "stage" in batched mode).
Metric: queries/sec over both stages.  With --gpus N every rank runs its own query batch against a replicated
corpus (weak scaling, no data-path collective); rank 0 prints ONE JSON line.

The JSON line also carries the roofline of the dominant kernel (the attention main kernel), timed with HIP
events on the launch stream inside the timed region, and a CPU baseline (the numpy / torch restatement of
the reference path under ``oracle/``) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd"))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec
BF16_PEAK = 2.5e15     # FLOP/s dense
FP32_MFMA_PEAK = 157.3e12

CFG = dict(n_layers=32, Hq=32, Hkv=8, D=128, sys_len=60, n_docs=10, doc_len=300, gap=1, qa_len=30,
           corpus_n=2_000_000, dim=1024, top_k=10, batch=32)


def layout(cfg):
    spans, pos = [], cfg["sys_len"]
    for _ in range(cfg["n_docs"]):
        spans.append((pos, pos + cfg["doc_len"]))
        pos += cfg["doc_len"] + cfg["gap"]
    return spans, pos, pos + cfg["qa_len"]


def make_corpus(n, d, device, seed=0):
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1 << 18
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        x = torch.randn((hi - lo, d), generator=g, device=device, dtype=torch.float32)
        out[lo:hi] = x / x.norm(dim=1, keepdim=True)
    return out


def cpu_baseline(cfg):
    """numpy / torch-CPU restatement of the reference path (oracle/), bounded sample."""
    from oracle import sdag_ref

    threads = torch.get_num_threads()
    # retrieval: sgemm + partial sort on a 200,000-row sample, scaled linearly to the corpus size
    n_s = 200_000
    rng = np.random.default_rng(0)
    C = rng.standard_normal((n_s, cfg["dim"]), dtype=np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = np.random.default_rng(1).standard_normal((cfg["batch"], cfg["dim"]), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    k = cfg["top_k"]
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        S = Q @ C.T
        part = np.argpartition(-S, k, axis=1)[:, :k]
        sc = np.take_along_axis(S, part, axis=1)
        order = np.argsort(-sc, axis=1, kind="stable")
        _ = np.take_along_axis(part, order, axis=1)
        best = min(best, time.perf_counter() - t)
    t_retr_batch = best * (cfg["corpus_n"] / n_s)
    del C
    # SDAG: mask build + additive mask + one layer of dense-masked SDPA, times n_layers
    spans, qa, L = layout(cfg)
    t = time.perf_counter()
    mask = torch.from_numpy(sdag_ref.mask_from_spans(L, cfg["sys_len"], spans, qa, None))
    add = sdag_ref.additive_mask(mask, torch.bfloat16)
    t_mask = time.perf_counter() - t
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, cfg["Hq"], L, cfg["D"], generator=g).to(torch.bfloat16)
    kk = torch.randn(1, cfg["Hkv"], L, cfg["D"], generator=g).to(torch.bfloat16)
    v = torch.randn(1, cfg["Hkv"], L, cfg["D"], generator=g).to(torch.bfloat16)
    G = cfg["Hq"] // cfg["Hkv"]
    best = 1e9
    with torch.no_grad():
        for _ in range(2):
            t = time.perf_counter()
            kr = kk.repeat_interleave(G, dim=1)
            vr = v.repeat_interleave(G, dim=1)
            _ = torch.nn.functional.scaled_dot_product_attention(q, kr, vr, attn_mask=add, scale=cfg["D"] ** -0.5)
            best = min(best, time.perf_counter() - t)
    t_sdag_query = t_mask + best * cfg["n_layers"]
    t_query = t_retr_batch / cfg["batch"] + t_sdag_query
    return dict(value=1.0 / t_query, unit="queries/sec", cores=threads, kind="port",
                sample=(f"retrieval: numpy sgemm+argpartition on {n_s} of {cfg['corpus_n']} rows x{cfg['corpus_n'] // n_s} "
                        f"({t_retr_batch * 1e3:.1f} ms/batch of {cfg['batch']}); SDAG: mask build + 1 of {cfg['n_layers']} layers of "
                        f"dense-masked torch SDPA at L={L} x{cfg['n_layers']} ({t_sdag_query:.2f} s/query)"),
                retrieval_qps=cfg["batch"] / t_retr_batch, sdag_qps=1.0 / t_sdag_query)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corpus-n", type=int, default=CFG["corpus_n"])
    ap.add_argument("--mode", choices=("batched", "per_query"), default="batched")
    ap.add_argument("--no-probe", action="store_true", help="skip the extra B=1 calls timed after the run (for profiling)")
    ap.add_argument("--exact-scan", action="store_true", help="retrieval scans the fp32 corpus (no bf16 candidate stage)")
    ap.add_argument("--attn-batch", type=int, default=CFG["batch"], help="queries per attention launch in batched mode (divides 32)")
    args = ap.parse_args()
    cfg = dict(CFG, corpus_n=args.corpus_n)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # BENCH_REHEARSE_ON_ONE_GPU=1: every rank uses GPU 0 and the timing barrier runs over gloo, to rehearse the
    # multi-rank control flow on a one-GPU box (numbers from such a run are meaningless).
    rehearse = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import sdag_amd

    spans, qa, L = layout(cfg)
    desc = sdag_amd.from_spans(L, cfg["sys_len"], spans, qa, None)
    pairs = desc.allowed_pairs
    nl, Hq, Hkv, D = cfg["n_layers"], cfg["Hq"], cfg["Hkv"], cfg["D"]

    # ---- inputs resident in HBM.  per_query: distinct Q/K/V/O per layer (2 GB).  batched: 4 rotating sets of
    # [32 queries, L, H, D] tensors (2 GB each, every query its own memory); both are far beyond the 256 MiB
    # last-level cache, so every launch streams its operands from HBM.
    def make_set(nb, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        mk = lambda h: torch.randn((nb, L, h, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16).transpose(1, 2)
        return mk(Hq), mk(Hkv), mk(Hkv), torch.empty((nb, L, Hq, D), device=dev, dtype=torch.bfloat16)

    batched = args.mode == "batched"
    nb = cfg["batch"]
    ab = args.attn_batch
    if batched:
        if nb % ab:
            raise SystemExit("--attn-batch must divide the query batch")
        descs = [sdag_amd.from_spans(L, cfg["sys_len"], spans, qa, None, max_items_per_unit=64) for _ in range(nb)]
        # one descriptor per query, as different prompts would have; ab queries share a launch
        bdescs = [sdag_amd.BatchDescriptor(descs[i:i + ab]) for i in range(0, nb, ab)]
        sets = [make_set(nb, 1000 * rank + i) for i in range(4)]
        bcalls = [[sdag_amd.BoundBatchAttention(q_[i:i + ab], k_[i:i + ab], v_[i:i + ab], bdescs[i // ab], o_[i:i + ab])
                   for i in range(0, nb, ab)] for q_, k_, v_, o_ in sets]
        layer_calls = [c for i in range(nl) for c in bcalls[i % len(bcalls)]]
        pq_sets = [tuple(t[:1] for t in sets[i]) for i in range(4)]
    else:
        pq_sets = [make_set(1, 1000 * rank + i) for i in range(nl)]
    pq_calls = [sdag_amd.BoundAttention(q_, k_, v_, desc, o_) for q_, k_, v_, o_ in pq_sets]
    corpus = make_corpus(cfg["corpus_n"], cfg["dim"], dev, seed=0)
    index = sdag_amd.FlatIPIndex(cfg["dim"], device=dev, two_stage=not args.exact_scan)
    index._corpus = corpus  # already resident; add() would copy
    index.prepare()         # index construction (the bf16 copy of the two-stage search), not part of a search
    gq = torch.Generator(device=dev).manual_seed(77 + rank)
    Q = torch.randn((cfg["batch"], cfg["dim"]), generator=gq, device=dev, dtype=torch.float32)
    Q = Q / Q.norm(dim=1, keepdim=True)

    stream = int(torch.cuda.current_stream(dev).cuda_stream)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    sdag_ms, retr_ms = [], []

    def step(timed):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        index.search_device(Q, cfg["top_k"])
        e1.record()
        if batched:
            for call in layer_calls:
                call(stream)
        else:
            for _ in range(cfg["batch"]):
                for call in pq_calls:
                    call(stream)
        e2.record()
        if timed:
            sdag_ms.append((e1, e2))
            retr_ms.append((e0, e1))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_launches = args.steps * nl * (nb // ab if batched else cfg["batch"])
    q_per_launch = ab if batched else 1
    t_sdag = sum(a.elapsed_time(b) for a, b in sdag_ms) * 1e-3
    t_retr = sum(a.elapsed_time(b) for a, b in retr_ms) * 1e-3
    per_launch = t_sdag / n_launches
    bytes_per_launch = q_per_launch * L * (2 * Hq + 2 * Hkv) * D * 2   # Q + K + V + O, each read/written once
    flops_per_launch = q_per_launch * 4 * D * pairs * Hq               # QK^T + PV over the allowed pairs
    achieved = bytes_per_launch / per_launch
    # the per-query call (B = 1, one launch per query and layer), timed outside the throughput region
    pq_us = float("nan")
    if not args.no_probe:
        for c in pq_calls[:4]:
            c(stream)
        torch.cuda.synchronize()
        ea, eb = ev(), ev()
        ea.record()
        for _ in range(8):
            for c in pq_calls[:4]:
                c(stream)
        eb.record()
        torch.cuda.synchronize()
        pq_us = ea.elapsed_time(eb) * 1e3 / 32
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get("sdag_attn_main_bytes_per_launch_" + args.mode)
        except Exception:
            traffic = None
    out = {
        "metric": "queries/sec (SDAG fwd + dense retrieval top-k), TOP_K=10",
        "value": world * args.steps * cfg["batch"] / elapsed,
        "unit": "queries/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16 attention (fp32 softmax/accumulate) + f32 retrieval scores (bf16 candidate scan)" if index.two_stage else "bf16 attention (fp32 softmax/accumulate) + f32 retrieval",
        "data": "synthetic",
        "config": {"workload": f"cfg2+cfg3: dense top-{cfg['top_k']} over {cfg['corpus_n']}x{cfg['dim']} f32 corpus + SDAG "
                               f"attention 32 layers x (Hq32,Hkv8,D128) L={L} (10 docs x 300 tok, no neighbours), "
                               f"{cfg['batch']} queries/step, attention launches {args.mode}", "batch_queries": cfg["batch"], "seq_len": L,
                   "parallelism": f"dp{world} over queries, corpus replicated"},
        "stage": {"retrieval_ms_per_batch": t_retr / args.steps * 1e3, "sdag_ms_per_query": t_sdag / (args.steps * cfg["batch"]) * 1e3,
                  "retrieval_qps": args.steps * cfg["batch"] / t_retr, "sdag_qps": args.steps * cfg["batch"] / t_sdag,
                  # bytes of the corpus copy the search scans: bf16 for the two-stage search, fp32 for the plain scan
                  "retrieval_scan": "bf16 candidates + exact fp32 rescoring" if index.two_stage else "fp32",
                  "retrieval_hbm_frac": (cfg["corpus_n"] * cfg["dim"] * (2 if index.two_stage else 4) / (t_retr / args.steps)) / HBM_PEAK,
                  "sdag_mfma_frac": flops_per_launch / per_launch / BF16_PEAK,
                  "sdag_per_query_call_us": None if pq_us != pq_us else pq_us,
                  "sdag_per_query_call_qps": None if pq_us != pq_us else 1e6 / (pq_us * nl)},
        "roofline": {"kernel": "sdag_attn_main + sdag_attn_combine, one launch = one layer of %d quer%s" % (q_per_launch, "ies" if q_per_launch > 1 else "y"),
                     "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": traffic, "us_per_launch": per_launch * 1e6,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
