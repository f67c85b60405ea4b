"""The CPU oracle against the fixtures produced by the reference's own Python (tests/golden/make_golden.py)."""
import json

import numpy as np
import torch

from helpers import GOLDEN, golden_bits, load_masks, mask_sha
from oracle import dense_ref, sdag_ref


def test_mask_restatement_matches_reference_masks():
    g = load_masks()
    for c in g["small"]:
        m = sdag_ref.mask_from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        assert np.array_equal(m, golden_bits(c)), c
    for c in g["big"]:
        m = sdag_ref.mask_from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        assert mask_sha(m) == c["sha256"], c["name"]
        assert int(m.sum()) == c["true_count"]
        assert int((m.sum(1) == 0).sum()) == c["empty_rows"]


def test_survey_properties_of_the_mask():
    m = sdag_ref.mask_from_spans(12, 3, [(3, 5), (6, 9)], 10, [[1], [0]])
    assert not m[5].any() and not m[9].any()          # separator rows allow nothing
    assert m[3, 6:9].all()                            # neighbour span to the right of the row is visible
    assert m[10, :11].all() and not m[10, 11]         # question rows are causal


def test_attention_restatement_matches_reference_eager_attention():
    z = np.load(f"{GOLDEN}/attention.npz")
    for name in ("toy", "mid", "nbr"):
        meta = json.loads(bytes(z[f"{name}_meta"]).decode())
        q = torch.from_numpy(z[f"{name}_q"]).view(torch.bfloat16)
        k = torch.from_numpy(z[f"{name}_k"]).view(torch.bfloat16)
        v = torch.from_numpy(z[f"{name}_v"]).view(torch.bfloat16)
        mask = torch.from_numpy(sdag_ref.mask_from_spans(meta["L"], meta["sys"], meta["spans"], meta["qa"], meta["nbr"]))
        o = sdag_ref.masked_attention(q, k, v, mask, 128 ** -0.5)
        ref = torch.from_numpy(z[f"{name}_o_f32"])
        assert torch.allclose(o, ref, atol=2e-6, rtol=1e-5)
        # separator rows: uniform softmax over every key = mean of V
        empty = ~mask.any(dim=1)
        if empty.any():
            mean_v = v.float().mean(dim=2)  # [B, Hkv, D]
            G = q.shape[1] // k.shape[1]
            for h in range(q.shape[1]):
                assert torch.allclose(ref[0, empty, h, :], mean_v[0, h // G].expand(int(empty.sum()), -1), atol=1e-5)


def test_small_ops_match_reference():
    with open(f"{GOLDEN}/small_ops.json") as f:
        g = json.load(f)
    import hashlib

    d = g["ranker_dim"]

    def enc(texts):
        rows = []
        for t in texts:
            seed = int.from_bytes(hashlib.sha256(t.encode()).digest()[:4], "little")
            v = np.random.default_rng(seed).standard_normal(d).astype(np.float32)
            rows.append(v / np.linalg.norm(v))
        return np.stack(rows) if rows else np.zeros((0, d), np.float32)

    for c in g["knn"]:
        docs = c["docs"]
        kept = [i for i, t in enumerate(docs) if t and t.strip()]
        emb = enc(["passage: " + docs[i] for i in kept])
        assert sdag_ref.doc_knn_from_embeddings(emb, kept, len(docs), c["k"]) == c["out"]
    for c in g["select"]:
        kept = [t for t in c["retrieved"] if t and t.strip()]
        r = enc(["passage: " + t for t in kept])
        cand = enc(["passage: " + t for t in c["cands"]])
        n = min(c["max_docs"], len(c["cands"])) if c["max_docs"] < len(c["cands"]) else len(c["cands"])
        if n == 1:
            picked = [c["cands"][sdag_ref.centroid_pick_one(r, cand, c["strategy"])]]
        else:
            picked = [c["cands"][int(i)] for i in sdag_ref.centroid_order(r, cand, c["strategy"])[:n]]
        assert picked == c["out"]
    h = g["hits"]
    out = sdag_ref.materialize_hits(h["idx"], h["sc"], h["meta"])
    assert [list(x) for x in out] == h["out"]
    assert g["search_index_order"] == {"first_is_indices": True, "second_is_scores": True}


def test_dense_oracle_definition():
    rng = np.random.default_rng(0)
    C = rng.standard_normal((3000, 96), dtype=np.float32)
    Q = rng.standard_normal((5, 96), dtype=np.float32)
    C[1500] = C[7]
    s, i = dense_ref.dense_topk(Q, C, 8, id_base=100)
    # definition: fp32 FMA chain in dimension order, (score desc, id asc)
    sc = dense_ref.dense_scores(Q, C)  # [N, nq]
    for q in range(5):
        order = np.lexsort((np.arange(3000), -sc[:, q].astype(np.float64)))[:8]
        assert np.array_equal(i[q], order + 100)
        assert np.array_equal(s[q], sc[order, q])
    # close to the sgemm the reference's index computes
    assert np.abs(sc.T - Q @ C.T).max() < 1e-4
    # fewer rows than k: faiss-style fill
    s, i = dense_ref.dense_topk(Q, C[:3], 5)
    assert (i[:, 3:] == -1).all() and (s[:, 3:] == -np.finfo(np.float32).max).all()
    # merge of shard results equals the global search
    parts = [dense_ref.dense_topk(Q, C[lo:lo + 1000], 8, id_base=lo) for lo in (0, 1000, 2000)]
    ms, mi = dense_ref.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), 8)
    gs, gi = dense_ref.dense_topk(Q, C, 8)
    assert np.array_equal(mi, gi) and np.array_equal(ms, gs)
