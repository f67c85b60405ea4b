"""Host logic of the drop-in modules, on CPU.  Where a step needs the GPU kernel, the test substitutes the CPU
oracle for it (the oracle is the checker here; the product modules never import it)."""
import importlib
import json
import random

import numpy as np
import pytest
import torch

from helpers import GOLDEN, expand_descriptor
from oracle import sdag_ref


@pytest.fixture()
def SDAG():
    return importlib.import_module("src.pipeline.sparse_attention_RAG.SDAG")


def test_module_surface_matches_reference_names(SDAG):
    for name in ("compute_doc_knn_for_docs", "build_blocked_causal_mask_full", "generate_with_custom_mask",
                 "build_rag_chat_and_spans", "run_rag_with_doc_isolation"):
        assert callable(getattr(SDAG, name))
    alias = importlib.import_module("src.pipeline.sdag")
    assert alias.run_rag_with_doc_isolation is SDAG.run_rag_with_doc_isolation
    dense = importlib.import_module("src.pipeline.retrieval.dense")
    for name in ("load_faiss_index_and_meta", "embed_queries", "search_index", "materialize_dense_hits", "DenseRetriever"):
        assert hasattr(dense, name)
    retr = importlib.import_module("src.pipeline.retrieval.retriever")
    assert issubclass(dense.DenseRetriever, retr.Retriever)
    with pytest.raises(TypeError):
        retr.Retriever()  # abstract


class ToyTokenizer:
    """Whitespace-and-punctuation tokenizer with a chat template: enough for span extraction."""
    eos_token_id = 2

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
        s = "<s>"
        for m in messages:
            s += f"[{m['role']}]\n{m['content']}</e>\n"
        return s + "[assistant]\n"

    def _split(self, text):
        import re
        return re.findall(r"\w+|[^\w\s]|\s+", text)

    def __call__(self, text, return_tensors="pt"):
        ids = torch.tensor([[hash(t) % 200 + 3 for t in self._split(text)]])

        class Enc(dict):
            def to(self, device):
                return self
        e = Enc(input_ids=ids)
        e.input_ids = ids
        return e

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(int(i)) for i in ids)


def test_build_rag_chat_and_spans_semantics(SDAG):
    tok = ToyTokenizer()
    docs = ["alpha beta gamma", "", "delta epsilon", "  ", "zeta"]
    chat, sys_len, spans, qa, ranked = SDAG.build_rag_chat_and_spans(tok, "SYS", SDAG.USER_RAG_PROMPT, "what is x?", ["EVIL doc"], docs, 2)
    assert ranked == ["alpha beta gamma", "EVIL doc", "", "delta epsilon", "  ", "zeta"]   # injected at rank 2
    assert len(spans) == 4                                                                 # empty documents have no bullet
    n = lambda s: len(tok(s).input_ids[0])
    pos = 0
    for (s, e), text in zip(spans, ["- alpha beta gamma", "- EVIL doc", "- delta epsilon", "- zeta"]):
        at = chat.find(text, pos)
        assert (s, e) == (n(chat[:at]), n(chat[:at + len(text)]))
        pos = at + len(text)
    assert sys_len == spans[0][0] and qa == n(chat[:chat.find("- Question:")])
    # no documents at all: the system/user section runs up to the question marker
    chat, sys_len, spans, qa, _ = SDAG.build_rag_chat_and_spans(tok, "SYS", SDAG.USER_RAG_PROMPT, "q", [], [], 0)
    assert spans == [] and sys_len == qa


def _reference_method_spans(tok, chat, bullets):
    """The reference's own way (SDAG.py:274-302): one tokenisation of the text before every boundary."""
    n = lambda t: len(tok(t, return_tensors="pt").input_ids[0])
    starts, cursor = [], 0
    for text in bullets:
        at = chat.find(text, cursor)
        at = cursor if at == -1 else at
        starts.append(at)
        cursor = at + len(text)
    q_at = chat.find("- Question:")
    q_at = len(chat) if q_at == -1 else q_at
    sys_len = n(chat[: starts[0] if starts else q_at])
    spans = [(n(chat[:at]), n(chat[: at + len(t)])) for t, at in zip(bullets, starts)]
    return sys_len, spans, n(chat[:q_at])


def test_single_pass_spans_equal_prefix_tokenisations(SDAG):
    """Token spans from ONE tokenisation of the prompt against the reference's 2K + 2 prefix tokenisations, on a
    byte-level BPE tokenizer of the Llama-3 kind: documents that end in punctuation (which the split pattern glues
    to the following blank line), contractions, digit runs, accents, inner blank lines; and every cut position of
    a prompt, not only the ones the pipeline uses."""
    import time

    from helpers import make_bpe_tokenizer

    tok, sentence = make_bpe_tokenizer(seed=3)
    assert SDAG._tokenizer_is_local(tok)
    rng = random.Random(11)
    t_fast = t_ref = 0.0
    for trial in range(25):
        docs = [sentence(5, 60) for _ in range(rng.randint(1, 8))]
        if trial % 5 == 0:
            docs.insert(1, "")
        mal = [sentence(3, 10)] if trial % 3 == 0 else []
        query = sentence(2, 6) + "?"
        t0 = time.perf_counter()
        chat, sys_len, spans, qa, ranked = SDAG.build_rag_chat_and_spans(tok, "SYS " + sentence(2, 5), SDAG.USER_RAG_PROMPT,
                                                                         query, mal, docs, 1)
        t_fast += time.perf_counter() - t0
        bullets = [f"- {d.strip()}" for d in ranked if d and d.strip()]
        t0 = time.perf_counter()
        want = _reference_method_spans(tok, chat, bullets)
        t_ref += time.perf_counter() - t0
        assert (sys_len, spans, qa) == want, trial
        assert qa <= len(tok(chat).input_ids)
    print(f"spans: single pass {t_fast * 1e3:.1f} ms, prefix tokenisations {t_ref * 1e3:.1f} ms over 25 prompts")
    # every cut of one prompt (the pipeline only cuts at bullets, but the count must hold anywhere)
    chat = tok.apply_chat_template([{"role": "user", "content": "It's 12345 units!\n\n- we're here...\n\n\n- x'r  y  \n z"}],
                                   tokenize=False, add_generation_prompt=True)
    cuts = list(range(len(chat) + 1))
    got = SDAG._prefix_token_counts(tok, chat, cuts)
    assert got == [len(tok(chat[:c]).input_ids) for c in cuts]
    # a tokenizer that is not of that kind takes the reference's route (and still agrees with it)
    toy = ToyTokenizer()
    assert not SDAG._tokenizer_is_local(toy)
    assert SDAG._prefix_token_counts(toy, "a b c d", [0, 3, 7]) == [len(toy("a b c d"[:c]).input_ids[0]) for c in (0, 3, 7)]


def test_spans_equal_reference_function_outputs(SDAG):
    """build_rag_chat_and_spans against what the REFERENCE's own function returned for the same inputs
    (tests/golden/make_golden_e2e.py ran SDAG.py:216-304 with this tokenizer): prompt text, ranked documents and every
    token position, over empty / whitespace / repeated documents, injected documents at one rank, past the end and at
    two ranks, no documents, and a document that contains the question marker."""
    import hashlib

    from helpers import load_e2e

    tok, g, _ = load_e2e()
    assert SDAG._tokenizer_is_local(tok)                       # the single-tokenisation route is the one under test
    for c in g["spans"] + g["e2e"]:
        chat, sys_len, spans, qa, ranked = SDAG.build_rag_chat_and_spans(tok, g["system_prompt"], g["user_template"],
                                                                         c["query"], c["mal"], c["docs"], c["rank"])
        assert hashlib.sha256(chat.encode()).hexdigest() == c["chat_sha256"]
        assert ranked == c["ranked_docs"]
        assert (sys_len, [list(s) for s in spans], qa) == (c["sys_user_len"], c["spans"], c["qa_start"])
    assert SDAG.SYSTEM_PROMPT_RAG == g["system_prompt"] and SDAG.USER_RAG_PROMPT == g["user_template"]


def test_run_rag_with_doc_isolation_matches_reference_text(SDAG, monkeypatch):
    """The public entry point end to end (spans -> document kNN -> descriptor -> prefill through the transformers
    hook -> decode loop) against the text and first-step logits the REFERENCE's run_rag_with_doc_isolation
    (SDAG.py:307-375) produced; on CPU the oracle stands in for the two GPU kernels (attention, Gram matrix)."""
    from helpers import FakeRanker, load_e2e, load_e2e_model
    from sdag_amd import hf_attention
    import sdag_amd

    tok, g, path = load_e2e()
    model, first_logits = load_e2e_model(g, path)

    def oracle_attention(q, k, v, desc, scaling=None, out_dtype=torch.bfloat16, out=None):
        mask = torch.from_numpy(expand_descriptor(desc.host))
        return sdag_ref.masked_attention(q.float(), k.float(), v.float(), mask, scaling)

    def oracle_knn(emb, kept, n, k, device=None):
        return sdag_ref.doc_knn_from_embeddings(emb.cpu().numpy(), kept, n, k)

    monkeypatch.setattr(hf_attention, "sdag_attention", oracle_attention)
    monkeypatch.setattr(hf_attention.torch, "bfloat16", torch.float32, raising=False)
    monkeypatch.setattr(sdag_amd, "doc_knn", oracle_knn)
    monkeypatch.setattr(SDAG, "TEMPERATURE", 0)
    ranker = FakeRanker(g["ranker_dim"])
    for c in g["e2e"]:
        seen = {}
        orig = model.forward

        def spy(*a, **kw):
            o = orig(*a, **kw)
            seen.setdefault("logits", o.logits[0, -1, :].detach().clone())
            return o

        model.forward = spy
        try:
            text = SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], "cpu",
                                                   c["rank"], ranker, c["nbr_k"])
        finally:
            model.forward = orig
        assert torch.allclose(seen["logits"], first_logits[c["name"]], atol=1e-4, rtol=1e-4), c["name"]
        assert text == c["text"], c["name"]


def test_batched_entry_point_matches_reference_texts(SDAG, monkeypatch):
    """run_rag_with_doc_isolation_batch (one right-padded prefill for all prompts, per-prompt decoding): the same
    texts the REFERENCE's per-query entry point produced, with the oracle standing in for the batched kernel."""
    from helpers import FakeRanker, load_e2e, load_e2e_model
    from sdag_amd import hf_attention
    import sdag_amd

    tok, g, path = load_e2e()
    model, _ = load_e2e_model(g, path)
    calls = {"batch": 0}

    def oracle_batch(q, k, v, batch, scaling=None, out_dtype=torch.bfloat16):
        calls["batch"] += 1
        out = torch.zeros((q.shape[0], q.shape[2], q.shape[1], q.shape[3]), dtype=torch.float32)
        for b, d in enumerate(batch.descs):
            mask = torch.from_numpy(expand_descriptor(d.host))
            out[b:b + 1, :d.L] = sdag_ref.masked_attention(q[b:b + 1, :, :d.L].float(), k[b:b + 1, :, :d.L].float(),
                                                           v[b:b + 1, :, :d.L].float(), mask, scaling)
        return out

    def oracle_knn(emb, kept, n, k, device=None):
        return sdag_ref.doc_knn_from_embeddings(emb.cpu().numpy(), kept, n, k)

    monkeypatch.setattr(hf_attention, "sdag_attention_batch", oracle_batch)
    monkeypatch.setattr(hf_attention.torch, "bfloat16", torch.float32, raising=False)
    monkeypatch.setattr(sdag_amd, "doc_knn", oracle_knn)
    monkeypatch.setattr(SDAG, "TEMPERATURE", 0)
    # the batch call takes one attack rank and one neighbour count for all queries: group the fixture's cases by them
    groups = {}
    for c in g["e2e"]:
        groups.setdefault((json.dumps(c["rank"]), c["nbr_k"]), []).append(c)
    ranker = FakeRanker(g["ranker_dim"])
    for (rank, nbr_k), cs in groups.items():
        texts = SDAG.run_rag_with_doc_isolation_batch(model, tok, [c["query"] for c in cs], [c["mal"] for c in cs],
                                                      [c["docs"] for c in cs], g["max_new_tokens"], "cpu", json.loads(rank),
                                                      ranker, nbr_k)
        assert texts == [c["text"] for c in cs]
    # and a mixed batch: different prompts, one call (same rank and neighbour count passed for all)
    cs = g["e2e"]
    texts = SDAG.run_rag_with_doc_isolation_batch(model, tok, [c["query"] for c in cs], [c["mal"] for c in cs],
                                                  [c["docs"] for c in cs], g["max_new_tokens"], "cpu", cs[0]["rank"], ranker, 0)
    singles = [SDAG_single(SDAG, model, tok, c, g, cs[0]["rank"], ranker, monkeypatch) for c in cs]
    assert texts == singles
    assert calls["batch"] >= g["model_cfg"]["num_hidden_layers"]

    # decode="batched": one model call per step for all prompts; the padding rows of the cache are left out by the
    # decode kernel (here: its fp32 restatement), every prompt keeps its own positions and its own end
    def oracle_decode(q, k, v, scaling=None, out_dtype=torch.bfloat16, valid_len=None, pad_end=0, n_rows=None):
        assert n_rows is None  # (a preallocated cache belongs to the graph-replayed loop, a GPU-only path)
        B, Hq, _, D = q.shape
        G = Hq // k.shape[1]
        out = torch.empty((B, 1, Hq, D), dtype=torch.float32)
        for b in range(B):
            keep = torch.ones(k.shape[2], dtype=torch.bool)
            if valid_len is not None:
                keep[int(valid_len[b]):pad_end] = False
            kk = k[b].float().repeat_interleave(G, dim=0)[:, keep]
            vv = v[b].float().repeat_interleave(G, dim=0)[:, keep]
            w = torch.softmax(q[b].float() @ kk.transpose(1, 2) * scaling, dim=-1)
            out[b, 0] = (w @ vv)[:, 0]
        return out

    decode_calls = {"padded": 0}

    def counting_decode(*a, **kw):
        decode_calls["padded"] += kw.get("valid_len") is not None
        return oracle_decode(*a, **kw)

    monkeypatch.setattr(hf_attention, "decode_attention", counting_decode)
    monkeypatch.setattr(hf_attention, "_on_gpu", lambda t: True)
    texts = SDAG.run_rag_with_doc_isolation_batch(model, tok, [c["query"] for c in cs], [c["mal"] for c in cs],
                                                  [c["docs"] for c in cs], g["max_new_tokens"], "cpu", cs[0]["rank"], ranker, 0,
                                                  decode="batched")
    assert texts == singles and decode_calls["padded"] > 0

    # prompts that end early: with the end-of-sequence id set to a token that some answers contain, both decode modes
    # must cut exactly those answers at it (and a first token equal to it ends nothing, as in the reference)
    class TokProxy:
        def __init__(self, inner, eos):
            self.inner, self.eos_token_id, self.seen = inner, eos, []

        def __getattr__(self, name):
            return getattr(self.inner, name)

        def __call__(self, *a, **kw):
            return self.inner(*a, **kw)

        def decode(self, ids_, **kw):
            self.seen.append([int(x) for x in ids_])
            return self.inner.decode(ids_, **kw)

    probe = TokProxy(tok, -1)
    args = ([c["query"] for c in cs], [c["mal"] for c in cs], [c["docs"] for c in cs], g["max_new_tokens"], "cpu", cs[0]["rank"], ranker, 0)
    SDAG.run_rag_with_doc_isolation_batch(model, probe, *args, decode="batched")
    full = probe.seen
    assert all(len(x) == g["max_new_tokens"] + 1 for x in full)
    for eos in {full[0][2], full[1][0], full[2][g["max_new_tokens"]], full[3][1]}:
        pa, pb = TokProxy(tok, eos), TokProxy(tok, eos)
        ta = SDAG.run_rag_with_doc_isolation_batch(model, pa, *args, decode="per_prompt")
        tb = SDAG.run_rag_with_doc_isolation_batch(model, pb, *args, decode="batched")
        assert ta == tb and pa.seen == pb.seen
        for ids_, whole in zip(pa.seen, full):
            cut = next((i for i in range(1, len(whole)) if whole[i] == eos), len(whole) - 1)
            assert ids_ == whole[:cut + 1]


def SDAG_single(SDAG, model, tok, c, g, rank, ranker, monkeypatch):
    """One query through the per-query entry point with the oracle standing in for the single-sequence kernel."""
    from sdag_amd import hf_attention

    def oracle_attention(q, k, v, desc, scaling=None, out_dtype=torch.bfloat16, out=None):
        mask = torch.from_numpy(expand_descriptor(desc.host))
        return sdag_ref.masked_attention(q.float(), k.float(), v.float(), mask, scaling)

    monkeypatch.setattr(hf_attention, "sdag_attention", oracle_attention)
    return SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], g["max_new_tokens"], "cpu", rank, ranker, 0)


def test_ranked_list_injection_rules():
    from sdag_amd.ranked_list import inject_malicious_docs_into_ranked_list as inj

    base = ["a", "b", "c"]
    assert inj(base, [], 1) == base and inj(base, ["x"], 0) == base
    assert inj(base, ["x", "y"], 2) == ["a", "x", "y", "b", "c"]
    assert inj(base, ["x"], 99) == ["a", "b", "c", "x"]
    assert inj(base, ["x"], -5) == base
    assert inj(base, ["x", "y"], [3, 1]) == ["y", "a", "b", "x", "c"]
    assert inj(base, ["x", "y"], [None, 2]) == ["a", "y", "b", "c"]
    random.seed(0)
    out = inj(base, ["x", "y"], [1])            # the missing position is padded with "random"
    assert out[0] == "x" and sorted(out) == ["a", "b", "c", "x", "y"]
    random.seed(1)
    assert sorted(inj(base, ["x"], -1)) == ["a", "b", "c", "x"]


def test_materialize_and_search_order():
    dense = importlib.import_module("src.pipeline.retrieval.dense")
    with open(f"{GOLDEN}/small_ops.json") as f:
        h = json.load(f)["hits"]
    out = dense.materialize_dense_hits(h["idx"], h["sc"], h["meta"])
    assert [list(x) for x in out] == h["out"]

    class FakeIndex:
        def search(self, q, k):
            assert q.dtype == np.float32
            return np.full((len(q), k), 0.5, np.float32), np.arange(len(q) * k, dtype=np.int64).reshape(len(q), k)

    ind, sc = dense.search_index(FakeIndex(), np.ones((2, 4), np.float64), 3)
    assert ind.dtype == np.int64 and sc.dtype == np.float32

    class FakeRanker:
        def encode(self, texts, convert_to_tensor=True, batch_size=32, normalize_embeddings=True):
            assert all(t.startswith("query: ") for t in texts)     # e5 prefix (dense.py:59)
            return torch.ones(len(texts), 4)

    r = dense.DenseRetriever(FakeRanker(), FakeIndex(), [{"id": i, "text": f"t{i}"} for i in range(4)])
    rb = r.retrieve_batch(["q1", "q2", "q3"], 3, 2)
    assert len(rb.q_embs) == 3 and rb.ids_full[0] == ["0", "1", "2"] and rb.ids_full[1] == ["3", "NA", "NA"]
    assert rb.docs_texts_full[1] == ["t3", "", ""] and rb.scores_full[2] == [0.5, 0.5, 0.5]


def test_flat_index_file_round_trip(tmp_path):
    from sdag_amd import index_io

    x = np.random.default_rng(0).standard_normal((37, 64)).astype(np.float32)
    p = str(tmp_path / "flat.index")
    index_io.write_flat_ip(p, x)
    assert np.array_equal(np.asarray(index_io.read_flat_vectors(p)), x)
    np.save(str(tmp_path / "c.npy"), x)
    assert np.array_equal(np.asarray(index_io.read_flat_vectors(str(tmp_path / "c.npy"))), x)
    dense = importlib.import_module("src.pipeline.retrieval.dense")
    with pytest.raises(FileNotFoundError, match="FAISS index not found"):
        dense.load_faiss_index_and_meta(str(tmp_path / "missing"), str(tmp_path / "c.npy"))
    with pytest.raises(FileNotFoundError, match="Metadata jsonl not found"):
        dense.load_faiss_index_and_meta(p, str(tmp_path / "missing.jsonl"))
    (tmp_path / "bad.index").write_bytes(b"IxF2" + b"\0" * 64)
    with pytest.raises(NotImplementedError):
        index_io.read_flat_vectors(str(tmp_path / "bad.index"))


def _tiny_llama():
    from transformers import LlamaConfig, LlamaForCausalLM

    z = np.load(f"{GOLDEN}/tiny_llama.npz")
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = LlamaConfig(**meta["cfg"])
    model = LlamaForCausalLM(cfg).eval()
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    model.load_state_dict(sd)
    return model, z, meta


def test_generate_through_hf_hook_matches_reference_greedy_text(SDAG, monkeypatch):
    """Prompt -> descriptor -> transformers attention hook -> decode loop, with the oracle standing in for the
    HIP kernel: prefill logits and greedy text equal what the reference's generate_with_custom_mask produced."""
    from sdag_amd import hf_attention
    import sdag_amd

    model, z, meta = _tiny_llama()
    c = meta["case"]
    calls = {"prefill": 0}

    def oracle_attention(q, k, v, desc, scaling=None, out_dtype=torch.bfloat16, out=None):
        calls["prefill"] += 1
        mask = torch.from_numpy(expand_descriptor(desc.host))
        return sdag_ref.masked_attention(q.float(), k.float(), v.float(), mask, scaling)

    monkeypatch.setattr(hf_attention, "sdag_attention", oracle_attention)
    monkeypatch.setattr(hf_attention.torch, "bfloat16", torch.float32, raising=False)  # keep the fp32 model in fp32
    monkeypatch.setattr(SDAG, "TEMPERATURE", 0)

    class Tok:
        eos_token_id = 2

        def decode(self, t, skip_special_tokens=True):
            return " ".join(str(int(i)) for i in t)

    ids = torch.from_numpy(z["input_ids"])
    desc = sdag_amd.from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
    seen = {}
    orig = model.forward

    def spy(*a, **kw):
        o = orig(*a, **kw)
        seen.setdefault("logits", o.logits[:, -1, :].detach().clone())
        return o

    model.forward = spy
    before = model.config._attn_implementation
    text = SDAG._generate_with_descriptor(model, Tok(), ids, desc, max_new_tokens=meta["max_new_tokens"])
    assert model.config._attn_implementation == before            # restored
    assert calls["prefill"] == meta["cfg"]["num_hidden_layers"]   # the kernel path ran once per layer, decode did not use it
    ref_logits = torch.from_numpy(z["prefill_last_logits"])
    assert torch.allclose(seen["logits"], ref_logits, atol=1e-4, rtol=1e-4)
    assert text == bytes(z["greedy_text"]).decode()
    assert len(text.split()) == meta["max_new_tokens"] + 1       # the reference's off-by-one is kept
    # a bool mask without an attached descriptor goes through from_dense and gives the same text
    mask = torch.from_numpy(sdag_ref.mask_from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"]))
    model.forward = orig
    assert SDAG.generate_with_custom_mask(model, Tok(), ids, mask, max_new_tokens=meta["max_new_tokens"]) == text


def test_hook_never_runs_a_prefill_unmasked_while_a_descriptor_is_active(monkeypatch):
    """The reference masks every prefill (SDAG.py:167-176).  While a descriptor (or a batch of them) is active, a
    call with more than one query row that does not fit it raises instead of reaching stock causal attention; a call
    the kernel cannot cover (head_dim 256) is refused on the CPU and masked through SDPA on a GPU; narrower heads (64)
    go to the kernel zero-padded, which refuses CPU tensors as well."""
    import sdag_amd
    from sdag_amd import hf_attention
    import transformers.integrations.sdpa_attention as sdpa_mod

    def forbidden(*a, **kw):
        raise AssertionError("stock attention reached while a descriptor was active")

    monkeypatch.setattr(sdpa_mod, "sdpa_attention_forward", forbidden)
    d40 = sdag_amd.from_spans(40, 5, [(5, 15), (16, 30)], 31)
    d33 = sdag_amd.from_spans(33, 4, [(4, 12), (13, 25)], 26)
    batch = sdag_amd.BatchDescriptor([d40, d33])

    def qkv(B, L, S=None, D=128, Hq=4, Hkv=2):
        S = L if S is None else S
        return torch.zeros(B, Hq, L, D), torch.zeros(B, Hkv, S, D), torch.zeros(B, Hkv, S, D)

    fwd = hf_attention.sdag_hip_attention_forward
    with hf_attention.active_descriptor(d40):
        for q, k, v in (qkv(1, 39), qkv(1, 41), qkv(1, 8, S=48)):       # shorter, longer, a chunk over a cache
            with pytest.raises(RuntimeError, match="SDAG prefill"):
                fwd(None, q, k, v, None, scaling=1.0)
        for D in (256, 64):                                               # the dense-mask route / the kernel: never unmasked
            with pytest.raises(RuntimeError, match="GPU only"):
                fwd(None, *qkv(1, 40, D=D), None, scaling=1.0)
    with hf_attention.active_descriptor(batch):
        for q, k, v in (qkv(1, 40), qkv(3, 40), qkv(2, 39), qkv(2, 6, S=46)):  # wrong batch, too short, chunked
            with pytest.raises(RuntimeError, match="SDAG prefill"):
                fwd(None, q, k, v, None, scaling=1.0)
        for D in (256, 64):
            with pytest.raises(RuntimeError, match="GPU only"):
                fwd(None, *qkv(2, 40, D=D), None, scaling=1.0)
        with pytest.raises(NotImplementedError):
            fwd(None, *qkv(2, 40), None, dropout=0.1, scaling=1.0)


def test_overlay_on_reference_tree_when_present():
    """With this package in front of the reference on the path, the two hot-path modules resolve here and
    everything else (config, data models, utils) stays the reference's.  Skipped where the reference is absent."""
    import os
    import subprocess
    import sys

    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "src", "pipeline")):
        pytest.skip("reference tree not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "sparse-document-attention-rag-_amd")
    code = (
        "import src.pipeline.retrieval.dense as d, src.pipeline.config as c, src.pipeline.sparse_attention_RAG.SDAG as S;"
        "import src.pipeline.models.datamodels as dm; from sdag_amd.datamodels import RetrievalBatch;"
        "import src.pipeline.attack.malicious_selection as M, src.pipeline.attack.doc_corruption as DC;"
        "print(d.__file__); print(S.__file__); print(c.__file__); print(RetrievalBatch is dm.RetrievalBatch);"
        "print(M.__file__); print(DC.__file__)")
    env = dict(os.environ, PYTHONPATH=f"{pkg}:{ref}", PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp", timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith(pkg) and lines[1].startswith(pkg) and lines[2].startswith(ref) and lines[3] == "True"
    assert lines[4].startswith(pkg) and lines[5].startswith(ref)   # Near / Far selection here, the rest of attack/ there


def _tie_groups(scores, payload):
    """[(score, sorted payloads with that score)] in ranking order."""
    out = []
    for s, p in zip(scores, payload):
        if out and out[-1][0] == s:
            out[-1][1].append(p)
        else:
            out.append((s, [p]))
    return [(s, sorted(map(repr, ps))) for s, ps in out]


def test_hybrid_rrf_fusion_matches_reference_outputs():
    """Fixtures produced by the reference's hybrid.py (tests/golden/make_golden_hybrid.py): scores identical float
    for float; documents identical per group of equal score (inside such a group the reference's own order depends
    on string hashing); this implementation orders a group by first appearance, sparse ranking first."""
    import random

    hyb = importlib.import_module("src.pipeline.retrieval.hybrid")
    with open(f"{GOLDEN}/hybrid.json") as f:
        g = json.load(f)
    for c in g["one"]:
        docs, ids, scores = hyb.rrf_fuse_one_query(c["sparse_docs"], c["sparse_ids"], c["dense_docs"], c["dense_ids"], k0=c["k0"])
        assert scores == c["scores"]
        assert _tie_groups(scores, zip(docs, ids)) == _tie_groups(c["scores"], zip(c["docs"], c["ids"]))
        first_seen = {}
        for t, i in list(zip(c["sparse_docs"], c["sparse_ids"])) + list(zip(c["dense_docs"], c["dense_ids"])):
            first_seen.setdefault(t if i in ("", "NA") else i, len(first_seen))
        order = [first_seen[t if i in ("", "NA") else i] for t, i in zip(docs, ids)]
        for a in range(len(order) - 1):
            assert scores[a] > scores[a + 1] or order[a] < order[a + 1]
    for c in g["batch"]:
        t, i, s = hyb.fuse_sparse_and_dense_batch(c["sparse_texts"], c["sparse_ids"], c["dense_texts"], c["dense_ids"],
                                                  top_k=c["top_k"], seed=c["seed"])
        assert s == c["scores"] and [len(x) for x in t] == [len(x) for x in c["texts"]]
        for q in range(len(s)):
            mine, ref = _tie_groups(s[q], zip(t[q], i[q])), _tie_groups(c["scores"][q], zip(c["texts"][q], c["ids"][q]))
            assert mine[:-1] == ref[:-1]              # the last group may be cut differently by the top_k limit
    for c in g["splits"]:
        r = random.Random(c["seed"])
        assert [list(hyb.split_k_between_sparse_and_dense(c["top_k"], r)) for _ in range(6)] == c["out"]

    class Fixed(hyb.Retriever):
        def __init__(self, tag):
            self.tag = tag

        def retrieve_batch(self, queries, k, bs):
            from sdag_amd.datamodels import RetrievalBatch

            return RetrievalBatch(q_embs=[self.tag] * len(queries), docs_texts_full=[[f"{self.tag}{j}" for j in range(k)] for _ in queries],
                                  ids_full=[[f"{self.tag}{j}" for j in range(k)] for _ in queries], scores_full=[[0.0] * k for _ in queries])

    rb = hyb.HybridRetriever(Fixed("d"), Fixed("s"), seed=3).retrieve_batch(["q1", "q2"], 5, 8)
    assert rb.q_embs == ["d", "d"] and all(len(x) == 5 for x in rb.ids_full)
    assert rb.ids_full[0][0] == "s0" and rb.ids_full[0][1] == "d0"      # equal scores: sparse ranking first


def test_malicious_selection_overlay_matches_reference_outputs(monkeypatch):
    """The Near / Far overlay module against what the REFERENCE's select_malicious_docs_for_batch and
    _select_malicious_docs_for_query returned (tests/golden/make_golden.py), including the queries that fall back to
    the ``random`` module: the picks and the state of the random generator afterwards must be the same.  On CPU the
    oracle's centroid arithmetic stands in for the GPU kernel."""
    import random

    import sdag_amd
    from helpers import FakeRanker

    sel = importlib.import_module("src.pipeline.attack.malicious_selection")
    with open(f"{GOLDEN}/small_ops.json") as f:
        g = json.load(f)

    def oracle_batch(rs, cs, device=None):
        out = []
        for r, c in zip(rs, cs):
            cen = np.asarray(r, np.float32).mean(axis=0, keepdims=True)
            out.append(np.matmul(np.asarray(c, np.float32), cen.T).reshape(-1))
        return out

    monkeypatch.setattr(sdag_amd, "centroid_sims_batch", oracle_batch)
    rk = FakeRanker(g["ranker_dim"])
    for c in g["select"]:
        assert sel._select_malicious_docs_for_query(rk, c["retrieved"], c["cands"], c["strategy"], c["max_docs"]) == c["out"]
    for c in g["select_batch"]:
        random.seed(c["seed"])
        out = sel.select_malicious_docs_for_batch(rk, c["retrieved"], c["cands"], c["strategy"], c["max_docs"])
        assert out == c["out"], (c["strategy"], c["max_docs"])
        assert random.random() == c["next_random"]
    assert sel.encode_texts_with_ranker(rk, []).shape == (0, g["ranker_dim"])
    assert sel.encode_texts_with_ranker(rk, ["a", "b"]).dtype == np.float32
