#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (diagnostic): the search against the C oracle (ids and score bits) over random
corpus sizes, dimensions, query counts and k, the attention kernels against the fp32 oracle over random layouts,
head counts, unit lengths, stream counts and batches, and the decode kernel (plain, right-padded, preallocated cache)
against an fp32 softmax on the same inputs.  usage: tools/gpu_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
import sdag_amd
from oracle import dense_ref, sdag_ref

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda:0"
bad = 0
for c in range(cases):
    d = int(rng.choice([32, 64, 96, 128, 256, 768, 1024]))
    N = int(rng.choice([0, 1, 5, 127, 128, 129, 1000, 4097, 33000, 70001]))
    if d >= 768 and N > 40000:
        N = 33000
    nq = int(rng.choice([1, 2, 31, 32, 33, 64, 65, 100, 129, 257]))
    k = int(rng.choice([1, 2, 5, 10, 11, 21, 32, 33, 50]))
    C = rng.standard_normal((N, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    if N > 10 and rng.random() < 0.5:
        C[rng.integers(0, N, 5)] = C[0]           # exact ties
        Q[0] = C[0]
    rs, ri = dense_ref.dense_topk(Q, C, k, id_base=9)
    for two_stage in (True, False):
        idx = sdag_amd.FlatIPIndex(d, device=dev, id_base=9, two_stage=two_stage)
        if N:
            idx.add(C)
        s, i = idx.search(Q, k)
        ok = np.array_equal(i, ri) and np.array_equal(s.view(np.uint32), rs.view(np.uint32))
        if not ok:
            bad += 1
            print("SEARCH MISMATCH", dict(N=N, d=d, nq=nq, k=k, two_stage=two_stage), flush=True)
    print(f"search {c}: N={N} d={d} nq={nq} k={k} ok", flush=True)

def rel_err(o, ref):
    den = ref.abs().amax(dim=(0, 1, 3)).clamp_min(1e-30)
    return float(((o - ref).abs().amax(dim=(0, 1, 3)) / den).max())

for c in range(cases):
    n_docs = int(rng.integers(0, 6))
    sys_len = int(rng.integers(0, 70))
    pos, spans = sys_len, []
    for _ in range(n_docs):
        n = int(rng.integers(0, 150))
        spans.append((pos, pos + n)); pos += n + int(rng.integers(0, 3))
    qa = pos
    L = pos + int(rng.integers(1, 50))
    nbr = None
    if n_docs and rng.random() < 0.6:
        nbr = [[int(x) for x in rng.integers(-1, n_docs + 1, int(rng.integers(0, 3)))] for _ in range(n_docs)]
    Hkv = int(rng.choice([1, 2, 8])); G = int(rng.choice([1, 2, 4, 5])); Hq = Hkv * G
    B = int(rng.choice([1, 1, 2, 3]))
    T = int(rng.choice([0, 1, 2, 8, 64])); ns = int(rng.choice([0, 1, 7, 64]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    q = torch.randn(B, L, Hq, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    kk = torch.randn(B, L, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, L, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    mask = torch.from_numpy(sdag_ref.mask_from_spans(L, sys_len, spans, qa, nbr))
    ref = sdag_ref.masked_attention(q, kk, v, mask, 128 ** -0.5)
    desc = sdag_amd.from_spans(L, sys_len, spans, qa, nbr, max_items_per_unit=T, n_streams=ns)
    o = sdag_amd.sdag_attention(q.to(dev), kk.to(dev), v.to(dev), desc, out_dtype=torch.float32).cpu()
    e1 = rel_err(o, ref)
    bd = sdag_amd.BatchDescriptor([desc] * B, n_streams=ns)
    ob = sdag_amd.sdag_attention_batch(q.to(dev), kk.to(dev), v.to(dev), bd, out_dtype=torch.float32).cpu()
    e2 = rel_err(ob, ref)
    m = torch.equal(desc.materialize(dev).cpu(), mask)
    if not (e1 <= 1e-3 and e2 <= 1e-3 and m):
        bad += 1
        print("ATTENTION MISMATCH", dict(L=L, sys=sys_len, spans=spans, qa=qa, nbr=nbr, Hq=Hq, Hkv=Hkv, B=B, T=T, ns=ns, e1=e1, e2=e2, mask=m), flush=True)
    print(f"attention {c}: L={L} docs={n_docs} Hq={Hq} Hkv={Hkv} B={B} T={T} streams={ns} err {e1:.1e} {e2:.1e}", flush=True)
# decode attention: random batch, head counts, cache lengths, right padding, preallocated caches with NaN tails
for c in range(cases):
    B = int(rng.choice([1, 2, 5, 32, 40])); Hkv = int(rng.choice([1, 2, 8])); G = int(rng.choice([1, 2, 4, 5, 7])); Hq = Hkv * G
    S = int(rng.choice([1, 2, 63, 64, 65, 200, 1000, 3100, 4500]))
    if B >= 32 and S > 1000:
        S = 1000
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    q = torch.randn(B, 1, Hq, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    kk = torch.randn(B, S, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, S, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    mode = int(rng.integers(0, 3))  # 0 plain, 1 right-padded, 2 preallocated (+ padded)
    live = torch.ones(B, S, dtype=torch.bool)
    valid = pad_end = n_rows = None
    kd, vd = kk.clone(), v.clone()
    if mode >= 1 and S > 2:
        pad_end = int(rng.integers(1, S + 1))
        vl = rng.integers(1, pad_end + 1, B)
        for b in range(B):
            live[b, vl[b]:pad_end] = False
            vd[b, :, vl[b]:pad_end] = float("nan")      # padding rows are never read as data
        valid = torch.tensor(vl, dtype=torch.int32, device=dev)
    if mode == 2 and S > 2:
        n = int(rng.integers(max(1, pad_end or 1), S + 1))
        live[:, n:] = False
        kd[:, :, n:] = float("nan"); vd[:, :, n:] = float("nan")
        n_rows = torch.tensor([n], dtype=torch.int32, device=dev)
    if not live.any(dim=1).all():
        continue
    o = sdag_amd.decode_attention(q.to(dev), kd.to(dev), vd.to(dev), out_dtype=torch.float32, valid_len=valid,
                                  pad_end=pad_end or 0, n_rows=n_rows).cpu()
    qf, kf, vf = q.float(), kk.float().repeat_interleave(G, 1), v.float().repeat_interleave(G, 1)
    sc = torch.einsum("bhqd,bhsd->bhqs", qf, kf) * 128 ** -0.5
    sc = sc.masked_fill(~live[:, None, None, :], float("-inf"))
    ref = torch.einsum("bhqs,bhsd->bhqd", torch.softmax(sc, -1), vf).transpose(1, 2)
    err = float(((o - ref).abs().amax(dim=(0, 1, 3)) / ref.abs().amax(dim=(0, 1, 3)).clamp_min(1e-30)).max())
    if not err <= 1e-3:
        bad += 1
        print("DECODE MISMATCH", dict(B=B, Hq=Hq, Hkv=Hkv, S=S, mode=mode, pad_end=pad_end, err=err), flush=True)
    print(f"decode {c}: B={B} Hq={Hq} Hkv={Hkv} S={S} mode={mode} err {err:.1e}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
