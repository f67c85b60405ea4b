"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE's own Python.

Run once in the development container (the reference tree is not available on the GPU
box, and is never copied into this repository):

    cd /tmp && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 \
        python /root/repo/tests/golden/make_golden.py

Only inputs and reference outputs are stored (npz / json).  ``sentence_transformers`` and
``faiss`` are absent from this image; the reference modules import them only for type
annotations (SDAG.py:7, malicious_selection.py:7) or for calls this script never reaches
(dense.py:36 ``faiss.read_index``), so empty stand-in modules are registered before import.
While generating, every case is also checked against the CPU oracle in ``oracle/`` so a
disagreement stops the script.
"""
from __future__ import annotations

import hashlib
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

st = types.ModuleType("sentence_transformers")
st.SentenceTransformer = type("SentenceTransformer", (), {})
sys.modules["sentence_transformers"] = st
sys.modules["faiss"] = types.ModuleType("faiss")

from src.pipeline.sparse_attention_RAG import SDAG as REF_SDAG  # noqa: E402
from src.pipeline.attack import malicious_selection as REF_SEL  # noqa: E402
from src.pipeline.retrieval import dense as REF_DENSE  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402
from transformers.models.llama.modeling_llama import eager_attention_forward  # noqa: E402

from oracle import sdag_ref  # noqa: E402


# ----------------------------------------------------------------------- mask cases
def synthetic_layout(sys_len, doc_lens, gap, qa_len):
    spans, pos = [], sys_len
    for n in doc_lens:
        spans.append((pos, pos + n))
        pos += n + gap
    qa_start = pos
    return spans, qa_start, qa_start + qa_len


def mask_cases():
    rng = random.Random(1234)
    cases = []
    # a toy case (two documents, a few tokens each), with and without neighbours
    cases.append(dict(L=12, sys=3, spans=[(3, 5), (6, 9)], qa=10, nbr=None))
    cases.append(dict(L=12, sys=3, spans=[(3, 5), (6, 9)], qa=10, nbr=[[1], [0]]))
    # neighbour list of the wrong length is ignored wholesale
    cases.append(dict(L=12, sys=3, spans=[(3, 5), (6, 9)], qa=10, nbr=[[1]]))
    # out-of-range neighbour ids are skipped
    cases.append(dict(L=40, sys=4, spans=[(4, 10), (11, 20), (21, 30)], qa=31, nbr=[[7, -1, 2], [0], [1, 99]]))
    # zero-length document, adjacent documents (no gap), qa_start == L, no documents
    cases.append(dict(L=30, sys=5, spans=[(5, 5), (5, 12), (12, 20)], qa=22, nbr=None))
    cases.append(dict(L=30, sys=5, spans=[(5, 12), (12, 20)], qa=30, nbr=None))
    cases.append(dict(L=17, sys=9, spans=[], qa=11, nbr=None))
    cases.append(dict(L=17, sys=0, spans=[(0, 8)], qa=9, nbr=[[0]]))
    # overlapping documents and a document overlapping the QA tail
    cases.append(dict(L=50, sys=6, spans=[(6, 20), (15, 30), (28, 45)], qa=40, nbr=[[2], [], [0, 1]]))
    # randomised
    for _ in range(18):
        K = rng.randint(1, 7)
        sys_len = rng.randint(0, 70)
        doc_lens = [rng.randint(0, 90) for _ in range(K)]
        gap = rng.randint(0, 3)
        spans, qa, L = synthetic_layout(sys_len, doc_lens, gap, rng.randint(0, 40))
        if L == 0:
            L = 1
        nbr = None
        if rng.random() < 0.6:
            nbr = [[rng.randint(-1, K) for _ in range(rng.randint(0, 3))] for _ in range(K)]
        cases.append(dict(L=L, sys=sys_len, spans=spans, qa=qa, nbr=nbr))
    return cases


def big_mask_cases():
    out = []
    spans, qa, L = synthetic_layout(60, [300] * 10, 1, 30)
    out.append(dict(name="cfg3_L3100", L=L, sys=60, spans=spans, qa=qa, nbr=None))
    spans, qa, L = synthetic_layout(60, [380] * 10, 1, 30)
    out.append(dict(name="cfg3_L3900", L=L, sys=60, spans=spans, qa=qa, nbr=None))
    K = 20
    spans, qa, L = synthetic_layout(60, [380] * K, 1, 30)
    out.append(dict(name="cfg4_L7710_nbr2", L=L, sys=60, spans=spans, qa=qa,
                    nbr=[[(i + 1) % K, (i + 2) % K] for i in range(K)]))
    rr = np.random.default_rng(2)
    lens = [int(x) for x in rr.integers(200, 401, size=10)]
    spans, qa, L = synthetic_layout(57, lens, 1, 33)
    out.append(dict(name="ragged_K10", L=L, sys=57, spans=spans, qa=qa, nbr=None))
    return out


def ref_mask(c):
    return REF_SDAG.build_blocked_causal_mask_full(c["L"], 500, c["sys"], [tuple(s) for s in c["spans"]], c["qa"],
                                                   device="cpu", doc_neighbors=c["nbr"]).numpy()


def gen_masks():
    small = []
    for c in mask_cases():
        m = ref_mask(c)
        o = sdag_ref.mask_from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        assert np.array_equal(m, o), c
        small.append(dict(c, bits=np.packbits(m).tobytes().hex()))
    big = []
    for c in big_mask_cases():
        m = ref_mask(c)
        o = sdag_ref.mask_from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        assert np.array_equal(m, o), c["name"]
        big.append(dict(c, sha256=hashlib.sha256(np.packbits(m).tobytes()).hexdigest(),
                        true_count=int(m.sum()), empty_rows=int((m.sum(1) == 0).sum()),
                        row_counts_sha256=hashlib.sha256(m.sum(1).astype(np.int32).tobytes()).hexdigest()))
    with open(os.path.join(HERE, "masks.json"), "w") as f:
        json.dump(dict(small=small, big=big), f)
    # a wider randomised sweep of oracle == reference that is not stored
    rng = random.Random(99)
    for _ in range(300):
        K = rng.randint(0, 6)
        sys_len = rng.randint(0, 40)
        lens = [rng.randint(0, 50) for _ in range(K)]
        spans, qa, L = synthetic_layout(sys_len, lens, rng.randint(0, 2), rng.randint(0, 20))
        L = max(L, 1)
        if rng.random() < 0.3 and K >= 2:  # shuffle into overlapping spans
            spans = [(max(0, s - rng.randint(0, 10)), min(L, e + rng.randint(0, 10))) for s, e in spans]
        nbr = [[rng.randint(-2, K + 1) for _ in range(rng.randint(0, 4))] for _ in range(K)] if rng.random() < 0.7 else None
        c = dict(L=L, sys=min(sys_len, L), spans=spans, qa=qa, nbr=nbr)
        assert np.array_equal(ref_mask(c), sdag_ref.mask_from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])), c
    print("masks ok:", len(small), "small,", len(big), "big, 300 random checks")


# ------------------------------------------------------------------ attention cases
class _Mod:
    def __init__(self, g):
        self.num_key_value_groups = g
        self.training = False


def gen_attention():
    out = {}
    cases = [
        dict(name="toy", L=12, sys=3, spans=[(3, 5), (6, 9)], qa=10, nbr=[[1], [0]], Hq=2, Hkv=1),
        dict(name="mid", L=200, sys=21, spans=[(21, 70), (71, 133), (134, 180)], qa=181, nbr=None, Hq=4, Hkv=1),
        dict(name="nbr", L=256, sys=17, spans=[(17, 80), (81, 150), (151, 222)], qa=223, nbr=[[2], [0], [1]], Hq=8, Hkv=2),
    ]
    for c in cases:
        g = torch.Generator().manual_seed(7 + len(out))
        L, D = c["L"], 128
        q = torch.randn(1, c["Hq"], L, D, generator=g).to(torch.bfloat16)
        k = torch.randn(1, c["Hkv"], L, D, generator=g).to(torch.bfloat16)
        v = torch.randn(1, c["Hkv"], L, D, generator=g).to(torch.bfloat16)
        mask = torch.from_numpy(ref_mask(c))
        for dt, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            # the additive mask exactly as the reference builds it (SDAG.py:164-173)
            neg = torch.finfo(dt).min
            attn = torch.zeros_like(mask, dtype=dt).masked_fill(~mask, neg).unsqueeze(0).unsqueeze(1)
            o, _ = eager_attention_forward(_Mod(c["Hq"] // c["Hkv"]), q.to(dt), k.to(dt), v.to(dt), attn,
                                           scaling=D ** -0.5)
            out[f"{c['name']}_o_{tag}"] = o.float().numpy()
        o_or = sdag_ref.masked_attention(q, k, v, mask, D ** -0.5)
        assert torch.allclose(o_or, torch.from_numpy(out[f"{c['name']}_o_f32"]), atol=2e-6, rtol=1e-5), c["name"]
        out[f"{c['name']}_q"] = q.view(torch.int16).numpy()
        out[f"{c['name']}_k"] = k.view(torch.int16).numpy()
        out[f"{c['name']}_v"] = v.view(torch.int16).numpy()
        out[f"{c['name']}_meta"] = np.frombuffer(json.dumps({kk: c[kk] for kk in ("L", "sys", "spans", "qa", "nbr", "Hq", "Hkv")}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **out)
    print("attention ok:", [c["name"] for c in cases])


# ------------------------------------------------------------- tiny-Llama prefill
TINY = dict(vocab_size=257, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
            num_key_value_heads=1, head_dim=128, max_position_embeddings=512)


def gen_tiny_llama():
    cfg = LlamaConfig(**TINY)
    cfg._attn_implementation = "eager"
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).eval()
    L = 96
    c = dict(L=L, sys=9, spans=[(9, 30), (31, 55), (56, 80)], qa=81, nbr=[[1], [2], [0]])
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, 257, (1, L), generator=g)
    mask = torch.from_numpy(ref_mask(c))
    captured = {}

    orig = model.forward

    def spy(*a, **kw):
        o = orig(*a, **kw)
        if "logits" not in captured:
            captured["logits"] = o.logits[:, -1, :].detach().clone()
        return o

    model.forward = spy

    class Tok:
        eos_token_id = 2

        def decode(self, t, skip_special_tokens=True):
            return " ".join(str(int(i)) for i in t)

    REF_SDAG.TEMPERATURE = 0  # greedy, so the text is reproducible without the RNG
    text = REF_SDAG.generate_with_custom_mask(model, Tok(), ids, mask, max_new_tokens=4)
    sd = {k_: v_.numpy() for k_, v_ in model.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "tiny_llama.npz"), input_ids=ids.numpy(), prefill_last_logits=captured["logits"].numpy(),
                        greedy_text=np.frombuffer(text.encode(), dtype=np.uint8),
                        meta=np.frombuffer(json.dumps(dict(cfg=TINY, case=c, max_new_tokens=4)).encode(), dtype=np.uint8),
                        **{"w::" + k_: v_ for k_, v_ in sd.items()})
    print("tiny llama ok: text=", text)


# ------------------------------------------------------- doc kNN / centroid / hits
class FakeRanker:
    """Deterministic stand-in for SentenceTransformer: unit vectors seeded by the text."""

    def __init__(self, d=64):
        self.d = d

    def get_sentence_embedding_dimension(self):
        return self.d

    def encode(self, texts, convert_to_tensor=True, normalize_embeddings=True, batch_size=32):
        rows = []
        for t in texts:
            seed = int.from_bytes(hashlib.sha256(t.encode()).digest()[:4], "little")
            v = np.random.default_rng(seed).standard_normal(self.d).astype(np.float32)
            rows.append(v / np.linalg.norm(v))
        return torch.from_numpy(np.stack(rows))


def gen_small_ops():
    rk = FakeRanker()
    docs_sets = [
        ["alpha one", "beta two", "", "gamma three", "   ", "delta four", "epsilon"],
        ["only one"],
        ["a", "b"],
        [f"document number {i} about topic {i % 3}" for i in range(12)],
    ]
    knn = []
    for docs in docs_sets:
        for kk in (0, 1, 2, 5):
            ref = REF_SDAG.compute_doc_knn_for_docs(rk, docs, kk)
            nonempty = [i for i, d in enumerate(docs) if d and d.strip()]
            emb = rk.encode(["passage: " + docs[i] for i in nonempty]).numpy() if nonempty else np.zeros((0, rk.d), np.float32)
            mine = sdag_ref.doc_knn_from_embeddings(emb, nonempty, len(docs), kk)
            assert ref == mine, (docs, kk, ref, mine)
            knn.append(dict(docs=docs, k=kk, out=ref))
    sel = []
    retrieved = [f"retrieved passage {i}" for i in range(10)] + ["", "  "]
    cands = [f"malicious candidate {i}" for i in range(5)]
    for strategy in ("closest_to_centroid", "furthest_from_centroid"):
        for n in (1, 2, 3, 5, 9):
            ref = REF_SEL._select_malicious_docs_for_query(rk, retrieved, cands, strategy, n)
            sel.append(dict(retrieved=retrieved, cands=cands, strategy=strategy, max_docs=n, out=ref))
    # the batch routine (malicious_selection.py:140-180) with queries that fall back to the random module in between
    sel_batch = []
    r_batch = [retrieved, [], [f"other passage {i}" for i in range(4)], ["", "  "], [f"third set {i}" for i in range(7)]]
    c_batch = [cands, cands[:3], [], cands[1:], [f"alt candidate {i}" for i in range(6)]]
    for strategy in ("closest_to_centroid", "furthest_from_centroid", "random", "no_such_strategy"):
        for n in (1, 2, 4, -1):
            random.seed(1000 + n)
            ref = REF_SEL.select_malicious_docs_for_batch(rk, r_batch, c_batch, strategy, n)
            sel_batch.append(dict(seed=1000 + n, retrieved=r_batch, cands=c_batch, strategy=strategy, max_docs=n, out=ref,
                                  next_random=random.random()))
    meta = [{"id": 10 + i, "text": f"text {i}"} for i in range(5)] + [{"text": "no id"}, {"id": "x"}]
    idx = [[0, 3, -1, 6], [5, 7, 100, 2]]
    sc = [[0.9, 0.5, -3.4e38, 0.1], [0.8, 0.7, 0.6, 0.5]]
    hits = REF_DENSE.materialize_dense_hits(idx, sc, meta)
    assert hits == sdag_ref.materialize_hits(idx, sc, meta)
    # search_index output order with a fake index (dense.py:82-84)
    class FakeIndex:
        def search(self, q, k):
            assert q.dtype == np.float32
            return np.full((len(q), k), 0.5, np.float32), np.arange(len(q) * k, dtype=np.int64).reshape(len(q), k)

    ind, scs = REF_DENSE.search_index(FakeIndex(), np.ones((2, 4), np.float64), 3)
    with open(os.path.join(HERE, "small_ops.json"), "w") as f:
        json.dump(dict(ranker_dim=rk.d, knn=knn, select=sel, select_batch=sel_batch,
                       hits=dict(meta=meta, idx=idx, sc=sc, out=[list(x) for x in hits]),
                       search_index_order=dict(first_is_indices=bool(ind.dtype == np.int64), second_is_scores=bool(scs.dtype == np.float32))), f)
    print("small ops ok")


if __name__ == "__main__":
    gen_masks()
    gen_attention()
    gen_tiny_llama()
    gen_small_ops()
