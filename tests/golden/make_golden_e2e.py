"""Generate e2e.json + e2e_model.npz by RUNNING THE REFERENCE's own entry points
``build_rag_chat_and_spans`` (SDAG.py:216-304) and ``run_rag_with_doc_isolation`` (SDAG.py:307-375).

    cd /tmp && PYTHONPATH=/root/reference:/root/repo:/root/repo/tests PYTHONDONTWRITEBYTECODE=1 \
        python /root/repo/tests/golden/make_golden_e2e.py

Stand-ins, all synthetic (no file of the reference is stored, nothing is fetched):
  * tokenizer: the in-memory byte-level BPE of the Llama-3 kind from tests/helpers.py; its serialised form is
    stored in the fixture so that the test runs the very same tokenizer;
  * model: a randomly initialised 2-layer Llama (head_dim 128, 2 query heads, 1 KV head) whose weights are stored;
  * ranker: unit vectors seeded by a hash of the text (``FakeRanker``).
The reference runs on CPU in fp32 with TEMPERATURE = 0 (greedy), so the text does not depend on an RNG.  Every
generated token's margin over the runner-up is recorded; model seeds are tried until every case keeps a margin that
bf16 attention operands cannot flip.
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
sys.path.insert(0, os.path.dirname(HERE))

st = types.ModuleType("sentence_transformers")
st.SentenceTransformer = type("SentenceTransformer", (), {})
sys.modules["sentence_transformers"] = st

from src.pipeline.sparse_attention_RAG import SDAG as REF_SDAG  # noqa: E402
from src.pipeline.utils.prompts import SYSTEM_PROMPT_RAG, USER_RAG_PROMPT  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

from helpers import FakeRanker, make_bpe_tokenizer  # noqa: E402

MAX_NEW = 6
MIN_MARGIN = 0.025  # of max |logit| at that step (bf16 attention operands move the logits by about 3e-3 of it)


def span_cases(tok, sentence):
    """Inputs for build_rag_chat_and_spans: ordinary, empty / whitespace documents, injected documents at a rank,
    past the end, at two ranks, no documents, a document repeated, a document that contains the question marker."""
    rng = random.Random(5)
    cases = []

    def docs(n, lo=5, hi=40):
        return [sentence(lo, hi) for _ in range(n)]

    cases.append(dict(query=sentence(2, 6) + "?", mal=[], docs=docs(4), rank=0))
    cases.append(dict(query=sentence(2, 6) + "?", mal=[sentence(4, 12)], docs=docs(5), rank=1))
    cases.append(dict(query=sentence(2, 6) + "?", mal=[sentence(4, 12), sentence(4, 12)], docs=docs(6), rank=3))
    cases.append(dict(query=sentence(2, 6) + "?", mal=[sentence(4, 12)], docs=docs(3), rank=99))
    d = docs(5)
    d.insert(1, "")
    d.insert(4, "   ")
    cases.append(dict(query=sentence(2, 6) + "?", mal=[sentence(3, 9)], docs=d, rank=2))
    cases.append(dict(query="why?", mal=[], docs=[], rank=0))
    d = docs(3)
    cases.append(dict(query=sentence(2, 6) + "?", mal=[], docs=[d[0], d[1], d[0], d[2]], rank=0))
    cases.append(dict(query=sentence(2, 6) + "?", mal=["see - Question: inside a document"], docs=docs(2), rank=2))
    cases.append(dict(query=sentence(2, 6) + "?", mal=[sentence(3, 9), sentence(3, 9)], docs=docs(4), rank=[4, 1]))
    for _ in range(6):
        n = rng.randint(1, 10)
        cases.append(dict(query=sentence(2, 8) + "?", mal=[sentence(3, 15)] if rng.random() < 0.5 else [],
                          docs=docs(n, 3, 70), rank=rng.randint(0, n + 1)))
    out = []
    for c in cases:
        chat, sys_len, spans, qa, ranked = REF_SDAG.build_rag_chat_and_spans(
            tok, SYSTEM_PROMPT_RAG, USER_RAG_PROMPT, c["query"], c["mal"], c["docs"], c["rank"])
        out.append(dict(c, chat_sha256=hashlib.sha256(chat.encode()).hexdigest(), n_tokens=len(tok(chat).input_ids),
                        sys_user_len=int(sys_len), spans=[[int(a), int(b)] for a, b in spans], qa_start=int(qa),
                        ranked_docs=ranked))
    return out


def e2e_inputs(sentence):
    def docs(n, lo=8, hi=40):
        return [sentence(lo, hi) for _ in range(n)]

    d_empty = docs(4)
    d_empty.insert(2, "")
    return [
        dict(name="iso_k0", query=sentence(3, 7) + "?", mal=[sentence(5, 14)], docs=docs(5), rank=2, nbr_k=0),
        dict(name="nbr_k2", query=sentence(3, 7) + "?", mal=[sentence(5, 14)], docs=docs(5), rank=1, nbr_k=2),
        # an empty document: doc_neighbors then has one list more than there are spans and the reference drops the
        # neighbours wholesale (SDAG.py:111)
        dict(name="empty_doc_k2", query=sentence(3, 7) + "?", mal=[sentence(5, 14)], docs=d_empty, rank=3, nbr_k=2),
        dict(name="no_attack_k1", query=sentence(3, 7) + "?", mal=[], docs=docs(3, 20, 60), rank=0, nbr_k=1),
    ]


def run_reference(model, tok, ranker, c):
    margins, steps = [], []
    orig = model.forward

    def spy(*a, **kw):
        o = orig(*a, **kw)
        lg = o.logits[0, -1, :].detach()
        top = torch.topk(lg, 2).values
        margins.append(float((top[0] - top[1]) / lg.abs().max()))
        steps.append(lg.clone())
        return o

    model.forward = spy
    try:
        text = REF_SDAG.run_rag_with_doc_isolation(model, tok, c["query"], c["mal"], c["docs"], MAX_NEW, "cpu", c["rank"],
                                                   ranker, c["nbr_k"])
    finally:
        model.forward = orig
    return text, margins, steps[0]


def main():
    tok, sentence = make_bpe_tokenizer(seed=3)
    REF_SDAG.TEMPERATURE = 0
    spans = span_cases(tok, sentence)
    inputs = e2e_inputs(sentence)
    ranker = FakeRanker(64)
    cfg = dict(vocab_size=len(tok), hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=1, head_dim=128, max_position_embeddings=4096)
    best = None
    for seed in range(24, 64):  # seeds 0-23 were tried: none keeps the margin
        lc = LlamaConfig(**cfg)
        lc._attn_implementation = "eager"
        torch.manual_seed(seed)
        model = LlamaForCausalLM(lc).eval()
        runs = [run_reference(model, tok, ranker, c) for c in inputs]
        worst = min(min(m) for _, m, _ in runs)
        print(f"seed {seed}: worst margin {worst:.4f}", [t for t, _, _ in runs])
        if best is None or worst > best[0]:
            best = (worst, seed, model, runs)
        if worst >= MIN_MARGIN:
            break
    worst, seed, model, runs = best
    assert worst >= MIN_MARGIN, f"no seed with margin >= {MIN_MARGIN} (best {worst})"
    e2e = []
    logits = {}
    for c, (text, margins, first) in zip(inputs, runs):
        chat, sys_len, sp, qa, ranked = REF_SDAG.build_rag_chat_and_spans(
            tok, SYSTEM_PROMPT_RAG, USER_RAG_PROMPT, c["query"], c["mal"], c["docs"], c["rank"])
        nbr = REF_SDAG.compute_doc_knn_for_docs(ranker, ranked, c["nbr_k"]) if c["nbr_k"] > 0 else None
        e2e.append(dict(c, text=text, margins=margins, chat_sha256=hashlib.sha256(chat.encode()).hexdigest(),
                        sys_user_len=int(sys_len), spans=[[int(a), int(b)] for a, b in sp], qa_start=int(qa),
                        ranked_docs=ranked, neighbors=nbr, n_tokens=len(tok(chat).input_ids)))
        logits["logits::" + c["name"]] = first.numpy()
    with open(os.path.join(HERE, "e2e.json"), "w") as f:
        json.dump(dict(tokenizer_json=tok.backend_tokenizer.to_str(), chat_template=tok.chat_template,
                       bos_token=tok.bos_token, eos_token=tok.eos_token, system_prompt=SYSTEM_PROMPT_RAG,
                       user_template=USER_RAG_PROMPT, model_cfg=cfg, model_seed=seed, max_new_tokens=MAX_NEW,
                       ranker_dim=64, spans=spans, e2e=e2e), f)
    np.savez_compressed(os.path.join(HERE, "e2e_model.npz"),
                        **{"w::" + k: v.numpy() for k, v in model.state_dict().items()}, **logits)
    print("e2e ok: seed", seed, "worst margin", worst, "L =", [c["n_tokens"] for c in e2e])


if __name__ == "__main__":
    main()
