#!/usr/bin/env python3
"""Throughput of the SDAG attention prefill + dense-retrieval search on MI355X.

One step = one batch of 32 queries per GPU through the hot path, inputs resident in HBM.  ``--config``:

  cfg3 (default; the configuration the metric is quoted on: retrieval of cfg 2 + attention of cfg 3)
      * retrieval: exact inner-product top-10 of Q[32, 1024] against a 2,000,000 x 1024 fp32 corpus (bf16 candidate
        scan + exact fp32 rescoring: the same ids and scores as a scan of the fp32 rows; --exact-scan for that scan);
      * SDAG: the masked attention of all 32 layers of 32 Llama-3-8B-shaped prompts (32 query heads, 8 KV heads,
        head_dim 128, bf16; system prefix 60 tokens, 10 documents with 1-token separators, 30 question tokens, no
        neighbours): 16 prompts with 300-token documents (L = 3100) and 16 ragged ones (document lengths U[200, 400]);
        one ``sdag_attn_fwd_batch`` launch per layer over the 32 prompts, every prompt with its own descriptor and
        its own Q/K/V/O memory;
      * the 32 descriptors and the batch table of a step are BUILT AND UPLOADED INSIDE THE TIMED REGION (host work of
        the step, done while the GPU runs the previous step's launches; upload from pinned memory on a side stream).
  cfg4  top-20 retrieval + prompts of 20 documents x 380 tokens, each document attending its 2 neighbours
        [(i+1) % 20, (i+2) % 20] (L = 7710).
  cfg5  corpus of 21,015,324 rows sharded over the ranks (local top-k with global ids -> ONE all-gather of the packed
        partial lists -> merge on every rank), batched Near/Far centroid scores, the cfg3 prefill, and 16 decode
        steps x 32 layers of single-token attention over the cache.

Metric: queries/sec over all stages.  ``--gpus N``: every rank runs its own query batch (weak scaling; the only
data-path collective is cfg5's all-gather); rank 0 prints ONE JSON line.  Without WORLD_SIZE in the environment
``--gpus N`` starts its N ranks itself, as child processes, before anything touches a GPU.

The JSON line also carries the roofline of the dominant kernel (the attention main kernel), timed with HIP events on
the launch stream inside the timed region, and a CPU baseline (the numpy / torch restatement of the reference path
under ``oracle/``) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd"))
sys.path.insert(0, ROOT)

RESULT_OUT = sys.stdout
T_START = time.perf_counter()
HBM_PEAK = 8.0e12      # B/s, MI355X HBM3E spec
BF16_PEAK = 2.5e15     # FLOP/s dense

BASE = dict(n_layers=32, Hq=32, Hkv=8, D=128, sys_len=60, gap=1, qa_len=30, dim=1024, batch=32)
CONFIGS = {
    "cfg3": dict(BASE, n_docs=10, doc_len=300, nbr=0, ragged=(200, 400), top_k=10, corpus_n=2_000_000, sharded=False,
                 centroid=False, decode_tokens=0),
    "cfg4": dict(BASE, n_docs=20, doc_len=380, nbr=2, ragged=None, top_k=20, corpus_n=2_000_000, sharded=False,
                 centroid=False, decode_tokens=0),
    "cfg5": dict(BASE, n_docs=10, doc_len=300, nbr=0, ragged=(200, 400), top_k=10, corpus_n=21_015_324, sharded=True,
                 centroid=True, decode_tokens=16),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--corpus-n", type=int, default=0, help="override the corpus size (rows over all ranks)")
    ap.add_argument("--no-probe", action="store_true", help="skip the extra launches timed after the run (for profiling)")
    ap.add_argument("--exact-scan", action="store_true", help="retrieval scans the fp32 corpus (no bf16 candidate stage)")
    ap.add_argument("--static-desc", action="store_true", help="build the descriptors once, outside the timed region")
    ap.add_argument("--dry-run", action="store_true", help="control flow only (launcher, rendezvous, timing), no GPU work")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """Start the ranks of ``--gpus N`` as child processes (one per GPU, fresh interpreters, nothing here has touched
    a GPU) and relay rank 0's line; the other ranks' stdout goes to stderr."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for other in pending:  # one rank failed: the others would wait in a barrier for good
                        other.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def prompts_of(cfg):
    """The 32 prompt layouts of a batch: (spans, qa_start, L, neighbours)."""
    import numpy as np

    rng = np.random.default_rng(2)
    out = []
    for i in range(cfg["batch"]):
        if cfg["ragged"] and i % 2:
            lens = [int(x) for x in rng.integers(cfg["ragged"][0], cfg["ragged"][1] + 1, size=cfg["n_docs"])]
        else:
            lens = [cfg["doc_len"]] * cfg["n_docs"]
        spans, pos = [], cfg["sys_len"]
        for n in lens:
            spans.append((pos, pos + n))
            pos += n + cfg["gap"]
        K = cfg["n_docs"]
        nbr = [[(d + j) % K for j in range(1, cfg["nbr"] + 1)] for d in range(K)] if cfg["nbr"] else None
        out.append((spans, pos, pos + cfg["qa_len"], nbr))
    return out


def trace(msg):
    """Progress lines on stderr when BENCH_TRACE=1 (to see where a multi-rank run stands)."""
    if os.environ.get("BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - T_START:.1f}s] {msg}", file=sys.stderr, flush=True)


def make_corpus(n, d, device, seed=0):
    import torch

    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    step = 1 << 18
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        x = torch.randn((hi - lo, d), generator=g, device=device, dtype=torch.float32)
        out[lo:hi] = x / x.norm(dim=1, keepdim=True)
    return out


def cpu_baseline(cfg, prompt):
    """numpy / torch-CPU restatement of the reference path (oracle/), bounded sample."""
    import numpy as np
    import torch

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
    # SDAG: mask build + additive mask + one layer of dense-masked SDPA, times n_layers (the first prompt of the batch)
    spans, qa, L, nbr = prompt
    t = time.perf_counter()
    mask = torch.from_numpy(sdag_ref.mask_from_spans(L, cfg["sys_len"], spans, qa, nbr))
    add = sdag_ref.additive_mask(mask, torch.bfloat16)
    t_mask = time.perf_counter() - t
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, cfg["Hq"], L, cfg["D"], generator=g).to(torch.bfloat16)
    kk = torch.randn(1, cfg["Hkv"], L, cfg["D"], generator=g).to(torch.bfloat16)
    v = torch.randn(1, cfg["Hkv"], L, cfg["D"], generator=g).to(torch.bfloat16)
    G = cfg["Hq"] // cfg["Hkv"]
    best = 1e9
    with torch.no_grad():
        for _ in range(2 if L < 5000 else 1):
            t = time.perf_counter()
            kr = kk.repeat_interleave(G, dim=1)
            vr = v.repeat_interleave(G, dim=1)
            _ = torch.nn.functional.scaled_dot_product_attention(q, kr, vr, attn_mask=add, scale=cfg["D"] ** -0.5)
            best = min(best, time.perf_counter() - t)
    t_sdag_query = t_mask + best * cfg["n_layers"]
    t_query = t_retr_batch / cfg["batch"] + t_sdag_query
    return dict(value=1.0 / t_query, unit="queries/sec", cores=threads, os_cpu_count=os.cpu_count(), kind="port",
                sample=(f"retrieval: numpy sgemm+argpartition on {n_s} of {cfg['corpus_n']} rows x{cfg['corpus_n'] / n_s:g} "
                        f"({t_retr_batch * 1e3:.1f} ms/batch of {cfg['batch']}); SDAG: mask build + 1 of {cfg['n_layers']} layers of "
                        f"dense-masked torch SDPA at L={L} x{cfg['n_layers']} ({t_sdag_query:.2f} s/query)"),
                retrieval_qps=cfg["batch"] / t_retr_batch, sdag_qps=1.0 / t_sdag_query)


def dry_run(args, world, rank):
    """Launcher / rendezvous / timing control flow without a GPU (what the CPU test of the self-launcher runs)."""
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = max(time.perf_counter() - t0, 1e-9)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({"metric": "queries/sec (SDAG fwd + dense retrieval top-k), TOP_K=10", "value": None, "unit": "queries/sec",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "config": {"workload": args.config}}), file=RESULT_OUT, flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class DescriptorPipeline:
    """Host side of a step's attention: the descriptors of the batch's prompts and their batch table, built into
    pinned memory and uploaded on a side stream.  ``slots`` tables can be in flight; a slot is reused only after the
    launches that read it have finished (its ``done`` event)."""

    def __init__(self, prompts, cfg, dev, slots=3):
        import torch

        import sdag_amd

        self.sdag, self.torch, self.prompts, self.cfg, self.dev = sdag_amd, torch, prompts, cfg, dev
        self.side = torch.cuda.Stream(device=dev)
        first = self._descs()
        words = sdag_amd.BatchDescriptor(first).host.size
        cap = int(words * 1.25) + 4096
        self.pinned = [torch.empty(cap, dtype=torch.int32, pin_memory=True) for _ in range(slots)]
        self.device = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(slots)]
        self.done = [None] * slots
        self.n = 0
        self.host_s = 0.0
        self.builds = 0

    def _descs(self):
        c = self.cfg
        return self.sdag.from_spans_many([(L, c["sys_len"], spans, qa, nbr) for spans, qa, L, nbr in self.prompts], max_items_per_unit=64)

    def build(self):
        """-> (batch table bound to its device copy, event after which the device copy is complete, slot)."""
        torch = self.torch
        slot = self.n % len(self.pinned)
        self.n += 1
        if self.done[slot] is not None:
            self.done[slot].synchronize()   # the host runs ahead of the GPU: wait until the slot's last reader is done
        t = time.perf_counter()
        batch = self.sdag.BatchDescriptor(self._descs(), out=self.pinned[slot].numpy())
        n = batch.host.size
        with torch.cuda.stream(self.side):
            self.device[slot][:n].copy_(self.pinned[slot][:n], non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(self.side)
        batch.bind_device_copy(self.device[slot])
        self.host_s += time.perf_counter() - t
        self.builds += 1
        return batch, ready, slot

    def mark_done(self, slot):
        ev = self.torch.cuda.Event()
        ev.record()
        self.done[slot] = ev


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the ONE JSON line and nothing else: whatever libraries print there (gloo and RCCL banners) is
    # sent to stderr, and the line goes to the descriptor stdout had at start
    global RESULT_OUT
    sys.stdout.flush()
    RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if args.dry_run:
        return dry_run(args, world, rank)

    import numpy as np
    import torch

    cfg = dict(CONFIGS[args.config])
    if args.corpus_n:
        cfg["corpus_n"] = args.corpus_n
    # BENCH_REHEARSE_ON_ONE_GPU=1: every rank uses GPU 0 and the collectives run over gloo, to rehearse the
    # multi-rank control flow on a one-GPU box (numbers from such a run are meaningless).
    rehearse = os.environ.get("BENCH_REHEARSE_ON_ONE_GPU") == "1"
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or cfg["sharded"]:
        import torch.distributed as dist_mod

        dist = dist_mod
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29671")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    trace("process group ready")
    import sdag_amd
    from sdag_amd import _lib

    lib = _lib.lib()
    prompts = prompts_of(cfg)
    nl, Hq, Hkv, D, nb = cfg["n_layers"], cfg["Hq"], cfg["Hkv"], cfg["D"], cfg["batch"]
    Ls = [p[2] for p in prompts]
    Lmax = max(Ls)
    S_cap = Lmax + cfg["decode_tokens"]

    # ---- inputs resident in HBM: rotating sets of [32 prompts, L, H, D] tensors (every prompt its own memory;
    # one set is 2-5 GB, far beyond the 256 MiB last-level cache, so every launch streams its operands from HBM)
    def make_set(seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        mk = lambda h, rows: torch.randn((nb, rows, h, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16).transpose(1, 2)
        return mk(Hq, Lmax), mk(Hkv, S_cap), mk(Hkv, S_cap), torch.empty((nb, Lmax, Hq, D), device=dev, dtype=torch.bfloat16)

    n_sets = 4 if Lmax < 5000 else 3
    sets = [make_set(1000 * rank + i) for i in range(n_sets)]
    pipe = DescriptorPipeline(prompts, cfg, dev)
    descs0 = pipe._descs()
    pairs = [d.allowed_pairs for d in descs0]

    # ---- retrieval index
    k = cfg["top_k"]
    nq_search = nb * world if cfg["sharded"] else nb
    if cfg["sharded"]:
        lo, hi = sdag_amd.shard_bounds(cfg["corpus_n"], world, rank)
        index = sdag_amd.ShardedFlatIPIndex(cfg["dim"], cfg["corpus_n"], dev)
        index.local = sdag_amd.FlatIPIndex.from_tensor(make_corpus(hi - lo, cfg["dim"], dev, seed=100 + rank), id_base=lo,
                                                       two_stage=not args.exact_scan)
        local_index = index.local
    else:
        index = local_index = sdag_amd.FlatIPIndex.from_tensor(make_corpus(cfg["corpus_n"], cfg["dim"], dev, seed=0),
                                                              two_stage=not args.exact_scan)
    local_index.prepare()   # index construction (the bf16 copy of the two-stage search), not part of a search
    trace("corpus resident")
    # cfg5: every rank holds the embeddings of all queries of the step (same seed everywhere) and answers its own 32
    gq = torch.Generator(device=dev).manual_seed(77 if cfg["sharded"] else 77 + rank)
    Q = torch.randn((nq_search, cfg["dim"]), generator=gq, device=dev, dtype=torch.float32)
    Q = Q / Q.norm(dim=1, keepdim=True)

    # ---- Near / Far inputs (cfg5): 10 retrieved + 5 candidate embeddings per query, one launch per batch
    if cfg["centroid"]:
        g = torch.Generator(device=dev).manual_seed(5 + rank)
        cen_R = torch.nn.functional.normalize(torch.randn((nb * 10, cfg["dim"]), generator=g, device=dev), dim=1)
        cen_C = torch.nn.functional.normalize(torch.randn((nb * 5, cfg["dim"]), generator=g, device=dev), dim=1)
        cen_off = torch.tensor([10 * i for i in range(nb + 1)] + [5 * i for i in range(nb + 1)], dtype=torch.int32, device=dev)
        cen_out = torch.empty((nb, cfg["dim"]), device=dev)
        cen_sims = torch.empty(nb * 5, device=dev)
        cen_host = torch.empty(nb * 5, pin_memory=True)
    if cfg["decode_tokens"]:
        dec_q = torch.randn((nb, Hq, 1, D), device=dev).to(torch.bfloat16)

    stream = int(torch.cuda.current_stream(dev).cuda_stream)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    sdag_ev, retr_ev = [], []
    enqueue_s = [0.0]

    static = None
    if args.static_desc:
        static = pipe.build()
        static[1].synchronize()
    nxt = [pipe.build() if static is None else static]

    def step(timed):
        t_host = time.perf_counter()
        batch, ready, slot = nxt[0]
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        index.search_device(Q, k)
        trace("search enqueued")
        if cfg["centroid"]:
            _lib.check(lib.sdag_centroid_sims_batch(cen_R.data_ptr(), cen_off.data_ptr(), cen_C.data_ptr(), cen_off[nb + 1:].data_ptr(),
                                                    nb, cfg["dim"], cen_out.data_ptr(), cen_sims.data_ptr(), stream))
            cen_host.copy_(cen_sims, non_blocking=True)
        e1.record()
        torch.cuda.current_stream(dev).wait_event(ready)
        calls = [sdag_amd.BoundBatchAttention(q_, k_[:, :, :Lmax], v_[:, :, :Lmax], batch, o_) for q_, k_, v_, o_ in sets]
        for layer in range(nl):
            calls[layer % n_sets](stream)
        e2.record()
        for t in range(cfg["decode_tokens"]):
            for layer in range(nl):
                _, k_, v_, _ = sets[layer % n_sets]
                sdag_amd.decode_attention(dec_q, k_[:, :, :Lmax + t + 1], v_[:, :, :Lmax + t + 1])
        if static is None:
            pipe.mark_done(slot)
        enqueue_s[0] += time.perf_counter() - t_host
        if static is None:
            nxt[0] = pipe.build()     # the next step's table: host work while the GPU runs this step
        if timed:
            sdag_ev.append((e1, e2))
            retr_ev.append((e0, e1))

    def sync():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
        trace("warmup step enqueued")
    sync()
    trace("warmup done")
    pipe.host_s, pipe.builds, enqueue_s[0] = 0.0, 0, 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    trace("timed steps done")
    if dist is not None and world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_launches = args.steps * nl
    t_sdag = sum(a.elapsed_time(b) for a, b in sdag_ev) * 1e-3
    t_retr = sum(a.elapsed_time(b) for a, b in retr_ev) * 1e-3
    per_launch = t_sdag / n_launches
    bytes_per_launch = sum(Ls) * (2 * Hq + 2 * Hkv) * D * 2          # Q + K + V + O of every prompt, each read/written once
    flops_per_launch = sum(pairs) * 4 * D * Hq                        # QK^T + PV over the allowed pairs
    achieved = bytes_per_launch / per_launch
    mfma_bound = args.config == "cfg4"

    # ---- probes outside the throughput region (rank 0 of a 1-GPU run)
    probe = {}
    if not args.no_probe and world == 1:
        def timed_us(fn, reps, inner):
            fn()
            torch.cuda.synchronize()
            a, b = ev(), ev()
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) * 1e3 / (reps * inner)

        # the per-query call: B = 1, one launch per query and layer, the way the reference calls its model
        spans, qa, L1, nbr = prompts[0]
        d1 = sdag_amd.from_spans(L1, cfg["sys_len"], spans, qa, nbr)
        pq = [sdag_amd.BoundAttention(q_[:1, :, :L1], k_[:1, :, :L1], v_[:1, :, :L1], d1, o_[:1, :L1]) for q_, k_, v_, o_ in sets]
        probe["sdag_per_query_call_us"] = timed_us(lambda: [c(stream) for c in pq], 8, len(pq))
        probe["sdag_per_query_call_qps"] = 1e6 / (probe["sdag_per_query_call_us"] * nl)
        # one decode step of 32 sequences over a cache of L rows
        dq = torch.randn((nb, Hq, 1, D), device=dev).to(torch.bfloat16)
        _, k_, v_, _ = sets[0]
        S = min(Ls)
        probe["decode_us_b32"] = timed_us(lambda: sdag_amd.decode_attention(dq, k_[:, :, :S], v_[:, :, :S]), 20, 1)
        probe["decode_hbm_frac_b32"] = nb * 2 * Hkv * S * D * 2 / (probe["decode_us_b32"] * 1e-6) / HBM_PEAK
        probe["decode_us_b1"] = timed_us(lambda: sdag_amd.decode_attention(dq[:1], k_[:1, :, :S], v_[:1, :, :S]), 20, 1)
        # the search with 256 queries in one call (two passes over the corpus of 128 queries each)
        Q256 = torch.nn.functional.normalize(torch.randn((256, cfg["dim"]), device=dev), dim=1)
        probe["retrieval_ms_nq256"] = timed_us(lambda: local_index.search_device(Q256, k), 5, 1) / 1e3
        probe["retrieval_hbm_frac_nq256"] = (local_index.ntotal * cfg["dim"] * (2 if local_index.two_stage else 4) * 2
                                             / (probe["retrieval_ms_nq256"] * 1e-3)) / HBM_PEAK
        del Q256
        if args.config != "cfg4":
            del pq
            # the "~4k-token context" variant of the prompt: 10 documents x 380 tokens, L = 3900, a batch of 32
            spans39, pos39 = [], cfg["sys_len"]
            for _ in range(10):
                spans39.append((pos39, pos39 + 380))
                pos39 += 381
            L39 = pos39 + 30
            g = torch.Generator(device=dev).manual_seed(39)
            mk = lambda h: torch.randn((nb, L39, h, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16).transpose(1, 2)
            q39, k39, v39 = mk(Hq), mk(Hkv), mk(Hkv)
            o39 = torch.empty((nb, L39, Hq, D), device=dev, dtype=torch.bfloat16)
            d39 = sdag_amd.from_spans(L39, cfg["sys_len"], spans39, pos39, None, max_items_per_unit=64)
            call39 = sdag_amd.BoundBatchAttention(q39, k39, v39, sdag_amd.BatchDescriptor([d39] * nb), o39)
            us39 = timed_us(lambda: call39(stream), 5, 1)
            probe["sdag_L3900_us_per_launch"] = us39
            probe["sdag_ms_per_query_L3900"] = us39 * nl / nb / 1e3
            probe["sdag_L3900_hbm_frac"] = nb * L39 * (2 * Hq + 2 * Hkv) * D * 2 / (us39 * 1e-6) / HBM_PEAK
            del q39, k39, v39, o39, call39
            # cfg 4 (20 documents x 380 tokens, 2 neighbours each, L = 7710) as a batch of 8: MFMA-bound by the accounting
            c4 = CONFIGS["cfg4"]
            p4 = prompts_of(c4)[:8]
            L4 = p4[0][2]
            g = torch.Generator(device=dev).manual_seed(4)
            mk = lambda h: torch.randn((8, L4, h, D), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16).transpose(1, 2)
            q4, k4, v4 = mk(Hq), mk(Hkv), mk(Hkv)
            o4 = torch.empty((8, L4, Hq, D), device=dev, dtype=torch.bfloat16)
            d4 = [sdag_amd.from_spans(L, c4["sys_len"], sp, qa, nb_, max_items_per_unit=64) for sp, qa, L, nb_ in p4]
            call4 = sdag_amd.BoundBatchAttention(q4, k4, v4, sdag_amd.BatchDescriptor(d4), o4)
            us4 = timed_us(lambda: call4(stream), 5, 8)
            probe["cfg4_us_per_query_layer"] = us4
            probe["cfg4_allowed_pairs"] = d4[0].allowed_pairs
            probe["cfg4_mfma_frac"] = 4 * D * d4[0].allowed_pairs * Hq / (us4 * 1e-6) / BF16_PEAK
            probe["cfg4_hbm_frac"] = L4 * (2 * Hq + 2 * Hkv) * D * 2 / (us4 * 1e-6) / HBM_PEAK
            del q4, k4, v4, o4, call4

    traffic, traffic_source = None, None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            # (the sharded-corpus workload launches the same 32-prompt attention as the default one)
            traffic = tj.get("sdag_attn_main_bytes_per_launch_" + ("cfg3" if args.config == "cfg5" else args.config))
            traffic_source = tj.get("source")
        except Exception:
            traffic = None
    k_desc = f"top-{k} over {cfg['corpus_n']}x{cfg['dim']} f32 corpus" + (f" sharded over {world} rank(s), one all-gather + merge" if cfg["sharded"] else "")
    shape = (f"{cfg['n_docs']} docs x {cfg['doc_len']} tok" + (f" (every other prompt ragged U{list(cfg['ragged'])})" if cfg["ragged"] else "")
             + (f", {cfg['nbr']} neighbours per doc" if cfg["nbr"] else ", no neighbours"))
    names = {"cfg3": "cfg2+cfg3", "cfg4": "cfg4", "cfg5": "cfg5"}
    out = {
        "metric": "queries/sec (SDAG fwd + dense retrieval top-k), TOP_K=10",
        "value": world * args.steps * nb / elapsed,
        "unit": "queries/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16 attention (fp32 softmax/accumulate) + f32 retrieval scores" + (" (bf16 candidate scan)" if local_index.two_stage else ""),
        "data": "synthetic",
        "config": {"workload": f"{names[args.config]}: dense {k_desc} + SDAG attention {nl} layers x (Hq{Hq},Hkv{Hkv},D{D}) "
                               f"L={min(Ls)}..{Lmax} ({shape}), {nb} queries/step/GPU, one batched launch per layer, descriptors "
                               + ("built once" if args.static_desc else "built and uploaded every step inside the timed region")
                               + (f", batched Near/Far scores, {cfg['decode_tokens']} decode steps" if cfg["centroid"] else ""),
                   "batch_queries": nb, "seq_len": Lmax, "parallelism": f"dp{world} over queries, corpus " + ("sharded" if cfg["sharded"] else "replicated")},
        "stage": dict({"retrieval_ms_per_batch": t_retr / args.steps * 1e3, "sdag_ms_per_query": t_sdag / (args.steps * nb) * 1e3,
                       "retrieval_qps": args.steps * nq_search / t_retr, "sdag_qps": args.steps * nb / t_sdag,
                       # bytes of the corpus copy the search scans: bf16 for the two-stage search, fp32 for the plain scan
                       "retrieval_scan": "bf16 candidates + exact fp32 rescoring" if local_index.two_stage else "fp32",
                       "retrieval_hbm_frac": (local_index.ntotal * cfg["dim"] * (2 if local_index.two_stage else 4) * (nq_search // nb) / (t_retr / args.steps)) / HBM_PEAK,
                       "sdag_hbm_frac": achieved / HBM_PEAK,
                       "sdag_mfma_frac": flops_per_launch / per_launch / BF16_PEAK,
                       "desc_build_ms_per_step": pipe.host_s / max(pipe.builds, 1) * 1e3,
                       "host_enqueue_ms_per_step": enqueue_s[0] / args.steps * 1e3}, **probe),
        "roofline": {"kernel": f"sdag_attn_main, one launch = one layer of {nb} prompts",
                     "bound": "mfma" if mfma_bound else "hbm",
                     "achieved": flops_per_launch / per_launch / 1e12 if mfma_bound else achieved / 1e9,
                     "peak": BF16_PEAK / 1e12 if mfma_bound else HBM_PEAK / 1e9, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                     "frac": flops_per_launch / per_launch / BF16_PEAK if mfma_bound else achieved / HBM_PEAK,
                     "traffic": traffic, "traffic_source": traffic_source, "us_per_launch": per_launch * 1e6,
                     "algorithmic_bytes_per_launch": bytes_per_launch, "flops_per_launch": flops_per_launch},
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, prompts[0])
        print(json.dumps(out), file=RESULT_OUT, flush=True)
    if dist is not None:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
