"""GPU parity of the public entry points end to end, of the benchmark's head layout, and of the sharded search at
the size of one shard of the 21 M-passage corpus."""
import importlib
import os

import numpy as np
import pytest
import torch

from helpers import FakeRanker, load_e2e, load_e2e_model, synthetic_layout

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel_err(o, ref):
    num = (o - ref).abs().amax(dim=(0, 1, 3))
    den = ref.abs().amax(dim=(0, 1, 3))
    return float((num / den).max())


def test_run_rag_with_doc_isolation_end_to_end():
    """``run_rag_with_doc_isolation`` (the one function the reference's main.py imports) on the GPU against what the
    REFERENCE's own function returned on CPU for the same inputs (tests/golden/make_golden_e2e.py): token spans,
    first-step logits (tolerance: bf16 attention operands inside an fp32 model, 2e-2 of max |logit|; the reference's
    smallest winning margin in these cases is 2.9e-2) and the greedy text, with the ranker's embeddings on either
    device, neighbours 0 / 1 / 2, an injected document and an empty one."""
    SDAG = importlib.import_module("src.pipeline.sparse_attention_RAG.SDAG")
    from sdag_amd import hf_attention

    tok, g, path = load_e2e()
    model, first_logits = load_e2e_model(g, path, device=DEV)
    old_t = SDAG.TEMPERATURE
    SDAG.TEMPERATURE = 0
    calls = {"prefill": 0, "decode": 0}
    real_prefill, real_decode = hf_attention.sdag_attention, hf_attention.decode_attention

    def counting_prefill(*a, **kw):
        calls["prefill"] += 1
        return real_prefill(*a, **kw)

    def counting_decode(*a, **kw):
        calls["decode"] += 1
        return real_decode(*a, **kw)

    hf_attention.sdag_attention, hf_attention.decode_attention = counting_prefill, counting_decode
    try:
        for n, c in enumerate(g["e2e"]):
            ranker = FakeRanker(g["ranker_dim"], device=DEV if n % 2 else "cpu")
            chat, sys_len, spans, qa, ranked = SDAG.build_rag_chat_and_spans(
                tok, g["system_prompt"], g["user_template"], c["query"], c["mal"], c["docs"], c["rank"])
            assert (sys_len, [list(s) for s in spans], qa, ranked) == (c["sys_user_len"], c["spans"], c["qa_start"], c["ranked_docs"])
            if c["nbr_k"] > 0:
                assert SDAG.compute_doc_knn_for_docs(ranker, ranked, c["nbr_k"]) == c["neighbors"]
            seen = {}
            orig = model.forward

            def spy(*a, **kw):
                o = orig(*a, **kw)
                seen.setdefault("logits", o.logits[0, -1, :].detach().float().cpu())
                return o

            model.forward = spy
            calls["prefill"] = calls["decode"] = 0
            try:
                text = SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], DEV,
                                                       c["rank"], ranker, c["nbr_k"])
            finally:
                model.forward = orig
            nl = g["model_cfg"]["num_hidden_layers"]
            assert calls["prefill"] == nl and calls["decode"] > 0 and calls["decode"] % nl == 0   # both kernels ran
            ref = first_logits[c["name"]]
            assert float((seen["logits"] - ref).abs().max()) <= 2e-2 * float(ref.abs().max()), c["name"]
            assert text == c["text"], c["name"]
    finally:
        hf_attention.sdag_attention, hf_attention.decode_attention = real_prefill, real_decode
        SDAG.TEMPERATURE = old_t


def test_graph_replayed_decode_gives_the_reference_text(monkeypatch):
    """SDAG_DECODE_GRAPH=1: the single-token steps are replayed from one captured HIP graph over a preallocated cache.
    Greedy texts must equal what the REFERENCE's own function returned (tests/golden/make_golden_e2e.py), the decode
    kernel must have run through the padded entry point (lengths read from the device), and a second answer from the
    same model (new graph, new cache) must be right as well."""
    SDAG = importlib.import_module("src.pipeline.sparse_attention_RAG.SDAG")
    from sdag_amd import hf_attention

    tok, g, path = load_e2e()
    model, _ = load_e2e_model(g, path, device=DEV)
    monkeypatch.setattr(SDAG, "TEMPERATURE", 0)
    monkeypatch.setenv("SDAG_DECODE_GRAPH", "1")
    seen = {"prealloc": 0, "other": 0}
    real_decode = hf_attention.decode_attention

    def counting_decode(*a, **kw):
        seen["prealloc" if kw.get("n_rows") is not None else "other"] += 1
        return real_decode(*a, **kw)

    monkeypatch.setattr(hf_attention, "decode_attention", counting_decode)
    for c in g["e2e"]:
        ranker = FakeRanker(g["ranker_dim"], device=DEV)
        text = SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], DEV,
                                               c["rank"], ranker, c["nbr_k"])
        assert text == c["text"], c["name"]
    # python-level calls: the eager first step and the captured one, per layer and answer; replays run no Python
    nl = g["model_cfg"]["num_hidden_layers"]
    assert seen["other"] == 0 and seen["prealloc"] == 2 * nl * len(g["e2e"])
    # the batched entry point, all prompts decoded together from one captured step (attack rank 2, one neighbour for
    # every prompt: a combination whose first-token margins are not near-ties on this random model)
    batch_args = ([c["query"] for c in g["e2e"]], [c["mal"] for c in g["e2e"]], [c["docs"] for c in g["e2e"]], g["max_new_tokens"], DEV, 2)
    singles = [SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], DEV, 2,
                                               FakeRanker(g["ranker_dim"], device=DEV), 1) for c in g["e2e"]]
    seen["prealloc"] = 0
    texts = SDAG.run_rag_with_doc_isolation_batch(model, tok, *batch_args, FakeRanker(g["ranker_dim"], device=DEV), 1, decode="batched")
    assert texts == singles and seen["other"] == 0 and seen["prealloc"] == 2 * nl
    # sampling inside the captured step: runs, and the same seed gives the same answer twice
    monkeypatch.setattr(SDAG, "TEMPERATURE", 0.7)
    c = g["e2e"][0]
    sampled = []
    for _ in range(2):
        torch.manual_seed(1234)
        sampled.append(SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], DEV, c["rank"],
                                                       FakeRanker(g["ranker_dim"], device=DEV), c["nbr_k"]))
    assert isinstance(sampled[0], str) and sampled[0] == sampled[1]
    monkeypatch.setattr(SDAG, "TEMPERATURE", 0)
    monkeypatch.setenv("SDAG_DECODE_GRAPH", "0")
    eager = SDAG.run_rag_with_doc_isolation_batch(model, tok, *batch_args, FakeRanker(g["ranker_dim"], device=DEV), 1, decode="batched")
    assert eager == texts and seen["other"] > 0


def test_batched_entry_point_on_gpu():
    """run_rag_with_doc_isolation_batch: one right-padded prefill (one batched kernel launch per layer, a descriptor per
    prompt) and per-prompt decoding.  Grouped by the fixture's (attack rank, neighbour count) its texts equal the
    REFERENCE's; a mixed batch equals one call of the per-query entry point per prompt."""
    import json

    SDAG = importlib.import_module("src.pipeline.sparse_attention_RAG.SDAG")
    from sdag_amd import hf_attention

    tok, g, path = load_e2e()
    model, _ = load_e2e_model(g, path, device=DEV)
    old_t = SDAG.TEMPERATURE
    SDAG.TEMPERATURE = 0
    calls = {"batch": 0}
    real = hf_attention.sdag_attention_batch

    def counting(*a, **kw):
        calls["batch"] += 1
        return real(*a, **kw)

    hf_attention.sdag_attention_batch = counting
    try:
        ranker = FakeRanker(g["ranker_dim"], device=DEV)
        groups = {}
        for c in g["e2e"]:
            groups.setdefault((json.dumps(c["rank"]), c["nbr_k"]), []).append(c)
        for (rank, nbr_k), cs in groups.items():
            texts = SDAG.run_rag_with_doc_isolation_batch(model, tok, [c["query"] for c in cs], [c["mal"] for c in cs],
                                                          [c["docs"] for c in cs], g["max_new_tokens"], DEV, json.loads(rank),
                                                          ranker, nbr_k)
            assert texts == [c["text"] for c in cs]
        cs = g["e2e"]
        calls["batch"] = 0
        texts = SDAG.run_rag_with_doc_isolation_batch(model, tok, [c["query"] for c in cs], [c["mal"] for c in cs],
                                                      [c["docs"] for c in cs], g["max_new_tokens"], DEV, 2, ranker, 1)
        assert calls["batch"] == g["model_cfg"]["num_hidden_layers"]          # ONE launch per layer for all prompts
        singles = [SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], DEV, 2,
                                                   ranker, 1) for c in cs]
        assert texts == singles
        # all prompts decoded together (padded decode kernel, per-prompt positions): the same greedy answers
        together = SDAG.run_rag_with_doc_isolation_batch(model, tok, [c["query"] for c in cs], [c["mal"] for c in cs],
                                                         [c["docs"] for c in cs], g["max_new_tokens"], DEV, 2, ranker, 1,
                                                         decode="batched")
        assert together == singles
    finally:
        hf_attention.sdag_attention_batch = real
        SDAG.TEMPERATURE = old_t


def test_mask_edited_in_place_is_honoured():
    """A mask from build_blocked_causal_mask_full that the caller edits in place afterwards: the prefill must follow
    the tensor's contents (SDAG.py:167-171), not the descriptor the tensor was expanded from."""
    SDAG = importlib.import_module("src.pipeline.sparse_attention_RAG.SDAG")
    import sdag_amd

    spans, qa, L = synthetic_layout(9, [40, 30, 50], 1, 12)
    mask = SDAG.build_blocked_causal_mask_full(L, 500, 9, spans, qa, device=DEV)
    d0 = SDAG._descriptor_of(mask, torch.bfloat16)
    assert d0 is mask._sdag_desc                                      # untouched: the remembered descriptor is used
    mask[qa:, spans[1][0]:spans[1][1]] = False                        # the question no longer sees document 1
    d1 = SDAG._descriptor_of(mask, torch.bfloat16)
    assert d1 is not d0
    assert torch.equal(d1.materialize(DEV), mask)
    assert d1.allowed_pairs == d0.allowed_pairs - (L - qa) * (spans[1][1] - spans[1][0])
    assert isinstance(d1, sdag_amd.Descriptor)


def test_attention_benchmark_head_layout():
    """The benchmark's head layout (32 query heads, 8 key/value heads: eight head slots per work stream) at
    L = 3100: a single-sequence launch and a two-sequence batch launch (the second sequence ragged, document lengths
    200-400) against the fp32 oracle, 1e-3 of max |ref| per head; bf16 output = rounding of the fp32 output."""
    import sdag_amd
    from oracle import sdag_ref

    Hq, Hkv = 32, 8
    spans, qa, L = synthetic_layout(60, [300] * 10, 1, 30)
    assert L == 3100
    lens = [int(x) for x in np.random.default_rng(2).integers(200, 401, size=10)]
    spans2, qa2, L2 = synthetic_layout(60, lens, 1, 30)
    Lmax = max(L, L2)
    g = torch.Generator().manual_seed(404)
    q = torch.randn(2, Lmax, Hq, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    k = torch.randn(2, Lmax, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(2, Lmax, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    d1 = sdag_amd.from_spans(L, 60, spans, qa, None)
    d2 = sdag_amd.from_spans(L2, 60, spans2, qa2, None)
    refs = []
    for b, (Lb, sp, qb) in enumerate(((L, spans, qa), (L2, spans2, qa2))):
        mask = torch.from_numpy(sdag_ref.mask_from_spans(Lb, 60, sp, qb, None))
        refs.append(sdag_ref.masked_attention(q[b:b + 1, :, :Lb], k[b:b + 1, :, :Lb], v[b:b + 1, :, :Lb], mask, 128 ** -0.5))
    o1 = sdag_amd.sdag_attention(qd[:1, :, :L], kd[:1, :, :L], vd[:1, :, :L], d1, out_dtype=torch.float32).cpu()
    assert _rel_err(o1, refs[0]) <= 1e-3
    ob = sdag_amd.sdag_attention(qd[:1, :, :L], kd[:1, :, :L], vd[:1, :, :L], d1).cpu()
    assert torch.equal(ob, o1.to(torch.bfloat16))
    batch = sdag_amd.BatchDescriptor([sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=64),
                                      sdag_amd.from_spans(L2, 60, spans2, qa2, None, max_items_per_unit=64)])
    o2 = sdag_amd.sdag_attention_batch(qd, kd, vd, batch, out_dtype=torch.float32).cpu()
    assert _rel_err(o2[:1, :L], refs[0]) <= 1e-3 and _rel_err(o2[1:, :L2], refs[1]) <= 1e-3
    assert d2.n_uniform_rows == 10


def test_attention_cfg4_benchmark_head_layout_batch():
    """The long-context configuration (20 documents x 380 tokens, neighbours [(i+1)%K, (i+2)%K], L = 7710) at the
    benchmark's head layout (32 query heads, 8 key/value heads) as a BATCH of two prompts -- the launch shape of
    ``bench.py --config cfg4``: the second prompt ragged (document lengths 200-400) and with neighbour ids outside
    [0, K) that the reference skips (SDAG.py:119-120).  fp32 output against the fp32 oracle, 1e-3 of max |ref| per
    head; the bf16 output equals the rounded fp32 output; rows past a prompt's end stay untouched."""
    import sdag_amd
    from oracle import sdag_ref

    Hq, Hkv, K = 32, 8, 20
    spans, qa, L = synthetic_layout(60, [380] * K, 1, 30)
    assert L == 7710
    nbr = [[(i + 1) % K, (i + 2) % K] for i in range(K)]
    lens = [int(x) for x in np.random.default_rng(4).integers(200, 401, size=K)]
    spans2, qa2, L2 = synthetic_layout(60, lens, 1, 30)
    nbr2 = [[(i + 1) % K, (i + 2) % K] for i in range(K)]
    nbr2[3] = [K, 4]          # out of range: skipped
    nbr2[7] = [-1, 9, 99]
    nbr2[11] = []
    Lmax = max(L, L2)
    g = torch.Generator().manual_seed(4040)
    q = torch.randn(2, Lmax, Hq, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    k = torch.randn(2, Lmax, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(2, Lmax, Hkv, 128, generator=g).to(torch.bfloat16).transpose(1, 2)
    qd, kd, vd = q.to(DEV), k.to(DEV), v.to(DEV)
    layouts = ((L, spans, qa, nbr), (L2, spans2, qa2, nbr2))
    descs = [sdag_amd.from_spans(Lb, 60, sp, qb, nb, max_items_per_unit=64) for Lb, sp, qb, nb in layouts]
    batch = sdag_amd.BatchDescriptor(descs)
    o32 = torch.full((2, Lmax, Hq, 128), 7.0, dtype=torch.float32, device=DEV)
    from sdag_amd.attention import BoundBatchAttention
    BoundBatchAttention(qd, kd, vd, batch, o32)()
    o16 = sdag_amd.sdag_attention_batch(qd, kd, vd, batch).cpu()
    o32 = o32.cpu()
    for b, (Lb, sp, qb, nb) in enumerate(layouts):
        mask_np = sdag_ref.mask_from_spans(Lb, 60, sp, qb, nb)
        assert np.array_equal(descs[b].materialize(DEV).cpu().numpy(), mask_np)
        ref = sdag_ref.masked_attention(q[b:b + 1, :, :Lb], k[b:b + 1, :, :Lb], v[b:b + 1, :, :Lb],
                                        torch.from_numpy(mask_np), 128 ** -0.5)
        err = _rel_err(o32[b:b + 1, :Lb], ref)
        print(f"cfg4 batch prompt {b}: L {Lb}, rel err {err:.2e}")
        assert err <= 1e-3
        assert torch.equal(o16[b, :Lb], o32[b, :Lb].to(torch.bfloat16))
        assert bool((o32[b, Lb:] == 7.0).all())


def test_prefill_on_other_head_dims_is_masked_not_causal(monkeypatch):
    """Models whose head_dim is not 128, under an active descriptor / batch of descriptors.  Narrower heads (64, 96) run
    the kernel on zero-padded copies -- no L x L mask is built, stock attention is never reached; wider ones (256) run
    SDPA under the dense mask expanded from the descriptor.  Never plain causal attention."""
    import sdag_amd
    import transformers.integrations.sdpa_attention as sdpa_mod
    from oracle import sdag_ref
    from sdag_amd import hf_attention

    spans, qa, L = synthetic_layout(7, [30, 18, 41], 1, 9)
    nbr = [[1], [2], []]
    spans2, qa2, L2 = synthetic_layout(5, [22, 35], 2, 6)
    d1 = sdag_amd.from_spans(L, 7, spans, qa, nbr)
    d2 = sdag_amd.from_spans(L2, 5, spans2, qa2, None)
    m1 = torch.from_numpy(sdag_ref.mask_from_spans(L, 7, spans, qa, nbr))
    m2 = torch.from_numpy(sdag_ref.mask_from_spans(L2, 5, spans2, qa2, None))

    class Mod:  # what transformers' SDPA wrapper reads from the attention module
        num_key_value_groups = 2
        is_causal = True
        training = False

    stock = sdpa_mod.sdpa_attention_forward
    reached = []

    def spy(*a, **kw):
        reached.append(1)
        return stock(*a, **kw)

    monkeypatch.setattr(sdpa_mod, "sdpa_attention_forward", spy)
    for D, tol in ((64, 1e-3), (96, 1e-3), (256, 1e-4)):
        g = torch.Generator().manual_seed(D)
        q = torch.randn(2, 4, L, D, generator=g)
        k = torch.randn(2, 2, L, D, generator=g)
        v = torch.randn(2, 2, L, D, generator=g)
        scale = D ** -0.5
        if D < 128:      # the kernel computes on bf16 operands: compare on what it sees
            q, k, v = (t.to(torch.bfloat16).float() for t in (q, k, v))
        reached.clear()
        with hf_attention.active_descriptor(d1):
            o, _ = hf_attention.sdag_hip_attention_forward(Mod(), q[:1].to(DEV), k[:1].to(DEV), v[:1].to(DEV), None,
                                                           scaling=None if D == 64 else scale)
        assert o.shape == (1, L, 4, D) and o.dtype == torch.float32
        ref1 = sdag_ref.masked_attention(q[:1], k[:1], v[:1], m1, scale)
        # wide heads: rows that allow nothing are left out -- what a fused SDPA kernel returns for a row masked with
        # finfo.min everywhere is that kernel's business (the reference on this GPU would get the same); the kernel
        # route returns the reference's mean of V for them, so there every row is compared (to bf16 output rounding)
        r1 = m1.any(1) if D > 128 else torch.ones(L, dtype=torch.bool)
        r2 = m2.any(1) if D > 128 else torch.ones(L2, dtype=torch.bool)
        otol = tol if D > 128 else 6e-3          # the hook returns the kernel's bf16 output (2^-8 relative)
        assert _rel_err(o.cpu()[:, r1], ref1[:, r1]) <= otol
        causal = sdag_ref.masked_attention(q[:1], k[:1], v[:1], torch.ones(L, L).tril().bool(), scale)
        assert _rel_err(causal[:, r1], ref1[:, r1]) > 5e-2        # the two differ, so the comparison above means something
        with hf_attention.active_descriptor(sdag_amd.BatchDescriptor([d1, d2])):
            ob, _ = hf_attention.sdag_hip_attention_forward(Mod(), q.to(DEV), k.to(DEV), v.to(DEV), None, scaling=scale)
        ob = ob.cpu()
        assert _rel_err(ob[:1, r1], ref1[:, r1]) <= otol
        ref2 = sdag_ref.masked_attention(q[1:, :, :L2], k[1:, :, :L2], v[1:, :, :L2], m2, scale)
        assert _rel_err(ob[1:, :L2][:, r2], ref2[:, r2]) <= otol
        assert bool(torch.isfinite(ob).all())
        assert bool(reached) == (D > 128)                          # stock attention only for the wide heads, and masked


def test_pipeline_chain_sharded_hybrid_selection_generation():
    """The flow of the end-to-end configuration at small scale, through the modules that carry the reference's names:
    corpus sharded over the ranks of an RCCL group (one rank here) -> DenseRetriever -> HybridRetriever (a lexical
    stand-in for the JVM retriever) -> Near / Far selection for the batch -> batched generation.  Every stage against
    its own check: search against the oracle, selection against per-query calls, answers against per-query calls."""
    import torch.distributed as dist

    import sdag_amd
    from oracle import dense_ref
    dense = importlib.import_module("src.pipeline.retrieval.dense")
    hybrid = importlib.import_module("src.pipeline.retrieval.hybrid")
    sel = importlib.import_module("src.pipeline.attack.malicious_selection")
    SDAG = importlib.import_module("src.pipeline.sparse_attention_RAG.SDAG")
    from sdag_amd.datamodels import RetrievalBatch

    tok, g, path = load_e2e()
    model, _ = load_e2e_model(g, path, device=DEV)
    rk = FakeRanker(g["ranker_dim"], device=DEV)
    _, sentence = __import__("helpers").make_bpe_tokenizer(seed=3)
    passages = [sentence(8, 30) for _ in range(400)]
    emb = rk.encode(["passage: " + t for t in passages]).float()
    meta = [{"id": 5000 + i, "text": t} for i, t in enumerate(passages)]
    queries = [sentence(3, 7) + "?" for _ in range(3)]

    class Lexical(hybrid.Retriever):                       # shares words with the query: enough for a second ranking
        def retrieve_batch(self, qs, k, bs):
            out_t, out_i, out_s = [], [], []
            for q in qs:
                words = set(q.lower().split())
                sc = sorted(((len(words & set(t.lower().split())), -i) for i, t in enumerate(passages)), reverse=True)[:k]
                out_t.append([passages[-j] for _, j in sc])
                out_i.append([str(5000 - j) for _, j in sc])
                out_s.append([float(n) for n, _ in sc])
            return RetrievalBatch(q_embs=[None] * len(qs), docs_texts_full=out_t, ids_full=out_i, scores_full=out_s)

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29654")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    old_t = SDAG.TEMPERATURE
    SDAG.TEMPERATURE = 0
    try:
        index = sdag_amd.ShardedFlatIPIndex(g["ranker_dim"], len(passages), DEV)
        index.add_local(emb)
        dr = dense.DenseRetriever(rk, index, meta)
        rb = dr.retrieve_batch(queries, 6, 2)
        qe = rk.encode(["query: " + q for q in queries]).float().cpu().numpy()
        rs, ri = dense_ref.dense_topk(qe, emb.cpu().numpy(), 6)
        assert rb.ids_full == [[str(5000 + int(j)) for j in row] for row in ri]
        fused = hybrid.HybridRetriever(dr, Lexical(), seed=11).retrieve_batch(queries, 5, 2)
        assert all(len(x) == 5 for x in fused.docs_texts_full)
        cands = [[sentence(4, 10) for _ in range(4)] for _ in queries]
        chosen = sel.select_malicious_docs_for_batch(rk, fused.docs_texts_full, cands, "closest_to_centroid", 1)
        assert chosen == [sel._select_malicious_docs_for_query(rk, d_, c_, "closest_to_centroid", 1)
                          for d_, c_ in zip(fused.docs_texts_full, cands)]
        answers = SDAG.run_rag_with_doc_isolation_batch(model, tok, queries, chosen, fused.docs_texts_full, 4, DEV, 2, rk, 2,
                                                        decode="batched")
        singles = [SDAG.run_rag_with_doc_isolation(model, tok, q, m_, d_, 4, DEV, 2, rk, 2)
                   for q, m_, d_ in zip(queries, chosen, fused.docs_texts_full)]
        assert answers == singles and all(isinstance(a, str) for a in answers)
    finally:
        SDAG.TEMPERATURE = old_t
        dist.destroy_process_group()


def _unit_corpus(n, d, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    out = torch.empty((n, d), dtype=torch.float32, device=DEV)
    for lo in range(0, n, 1 << 18):
        hi = min(n, lo + (1 << 18))
        x = torch.randn((hi - lo, d), generator=g, device=DEV)
        out[lo:hi] = x / x.norm(dim=1, keepdim=True)
    return out


def test_sharded_index_on_rccl_and_8_way_split_of_21m_corpus():
    """The sharded search on the HIP path at the size of the 21 M-passage corpus (21,015,324 x 1024 fp32 = 86 GB
    resident, + 43 GB bf16 copy):
    (a) the 8-way split (shards of 2,626,916 rows, id_base = rank * ceil(N / 8)): every shard searched on its own, the
        eight partial lists merged by ``sdag_topk_merge`` -> ids and score bits equal to ONE search over the whole
        corpus; the scores of the returned rows are the oracle's, bit for bit;
    (b) ``ShardedFlatIPIndex`` under a one-rank ``nccl`` (= RCCL) process group: the real all_gather_into_tensor and
        merge kernel of the product path, against the single index and the oracle."""
    import torch.distributed as dist

    import sdag_amd
    from oracle import dense_ref

    d, nq, k = 1024, 32, 10
    N = 21_015_324
    per = -(-N // 8)
    assert per == 2_626_916
    gq = torch.Generator(device=DEV).manual_seed(5)
    Q = torch.randn((nq, d), generator=gq, device=DEV)
    Q = Q / Q.norm(dim=1, keepdim=True)
    Qh = Q.cpu().numpy()
    C = _unit_corpus(N, d, seed=1000)
    whole = sdag_amd.FlatIPIndex.from_tensor(C)
    s, i = whole.search_device(Q, k)
    assert not whole.last_search_fell_back()
    sh_, ih = s.cpu().numpy(), i.cpu().numpy()
    assert (ih >= 0).all() and (ih < N).all() and (np.diff(sh_, axis=1) <= 0).all()
    rows = np.unique(ih)
    sc = dense_ref.dense_scores(Qh, C[torch.from_numpy(rows).to(DEV)].cpu().numpy())               # [rows, nq]
    pos = {int(x): n for n, x in enumerate(rows)}
    exact = np.array([[sc[pos[int(x)], qi] for x in ih[qi]] for qi in range(nq)], dtype=np.float32)
    assert np.array_equal(exact.view(np.uint32), sh_.view(np.uint32))
    del whole
    parts = []
    for r in range(8):
        lo, hi = sdag_amd.shard_bounds(N, 8, r)
        assert lo == r * per and hi == min(N, lo + per)
        idx = sdag_amd.FlatIPIndex.from_tensor(C[lo:hi], id_base=lo)
        ps, pi = idx.search_device(Q, k)
        assert int(pi.min()) >= lo and int(pi.max()) < hi
        parts.append((ps, pi))
        del idx
    ms, mi = sdag_amd.merge_partial_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
    assert torch.equal(mi, i) and torch.equal(ms, s)

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29653")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        Csub = C[: 1 << 18]
        shd = sdag_amd.ShardedFlatIPIndex(d, Csub.shape[0], DEV)
        shd.add_local(Csub)
        s1, i1 = shd.search_device(Q, k)
        s0, i0 = sdag_amd.FlatIPIndex.from_tensor(Csub).search_device(Q, k)
        assert torch.equal(i1, i0) and torch.equal(s1, s0)
        gs, gi = dense_ref.dense_topk(Qh, Csub.cpu().numpy(), k)
        assert np.array_equal(i1.cpu().numpy(), gi) and np.array_equal(s1.cpu().numpy().view(np.uint32), gs.view(np.uint32))
        # more than 32 results per query: pages of 32 behind a global cursor, one gather + merge per page
        s2, i2 = shd.search_device(Q, 40)
        gs, gi = dense_ref.dense_topk(Qh, Csub.cpu().numpy(), 40)
        assert np.array_equal(i2.cpu().numpy(), gi) and np.array_equal(s2.cpu().numpy().view(np.uint32), gs.view(np.uint32))
        # the same second page from four shards of that corpus: the cursor is the GLOBAL 32nd entry on every shard
        a_s, a_i = s2[:, 31].contiguous(), i2[:, 31].contiguous()
        parts = []
        for r in range(4):
            lo, hi = sdag_amd.shard_bounds(Csub.shape[0], 4, r)
            parts.append(sdag_amd.FlatIPIndex.from_tensor(Csub[lo:hi], id_base=lo).search_after(Q, 8, a_s, a_i))
        ms, mi = sdag_amd.merge_partial_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), 8)
        assert torch.equal(mi, i2[:, 32:]) and torch.equal(ms, s2[:, 32:])
    finally:
        dist.destroy_process_group()
    del C
    torch.cuda.empty_cache()


def test_bench_two_ranks_rehearsed_on_one_gpu():
    """``bench.py --gpus 2 --config cfg5`` the way the round driver starts it (the launcher creates one process per
    rank), both ranks on GPU 0 with the collectives over gloo: the sharded search (local top-k, one gather, merge), the
    batched Near / Far kernel, the prefill launches and the decode steps all run for real at world size 2, and rank 0
    prints ONE line with the keys of the one-rank line.  Started through the helper process of conftest.py, which has
    never touched the GPU."""
    import json
    import os
    import sys

    from conftest import run_in_fresh_process

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_REHEARSE_ON_ONE_GPU"] = "1"
    lines = {}
    for n in (2, 1):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--config", "cfg5", "--corpus-n", "400000",
               "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-probe"]
        res = run_in_fresh_process(cmd, env, 420)
        if res is None:
            pytest.skip("no helper process to start the benchmark from")
        assert res["rc"] == 0, res["stderr"]
        out = [l for l in res["stdout"].splitlines() if l.strip()]
        assert len(out) == 1, res["stdout"]
        lines[n] = json.loads(out[0])
    two, one = lines[2], lines[1]
    assert two["n_gpus"] == 2 and two["steps"] == 2 and two["scaling"] == "weak"
    assert "sharded" in two["config"]["parallelism"] and two["config"]["parallelism"].startswith("dp2")
    assert set(two) == set(one) and "cpu_baseline" not in two
    assert set(two["stage"]) == set(one["stage"]) and set(two["roofline"]) == set(one["roofline"])
    assert two["value"] > 0 and two["stage"]["retrieval_ms_per_batch"] > 0
