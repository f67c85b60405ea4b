"""The CPU oracle against the fixtures produced by the reference's own Python (tests/golden/make_golden.py)."""
import json

import numpy as np
import pytest
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


def test_structural_properties_of_the_mask():
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


@pytest.mark.parametrize("d,N", [(768, 200_000), (1024, 200_000)])
def test_search_oracle_against_fp64_ranking(d, N):
    """Independent cross-check of the search oracle's arithmetic (faiss, whose arithmetic the reference uses, is not
    available to pin it): on random unit corpora the C oracle's fp32 FMA-chain scores must lie within the chain's
    rounding bound d * 2^-24 * |q||c| of numpy's fp64 scores, its top-k SET must equal the fp64 top-k whenever the
    fp64 gap between ranks k and k+1 exceeds twice that bound, and every position whose fp64 neighbours are both
    further away than that must hold the same id.  The number of queries with a boundary gap under the bound is
    printed: only for those could a different fp32 summation order (BLAS blocking) return a different set."""
    from oracle import dense_ref

    rng = np.random.default_rng(d)
    C = rng.standard_normal((N, d), dtype=np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((32, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    k = 10
    s, i = dense_ref.dense_topk(Q, C, k)
    S64 = Q.astype(np.float64) @ C.astype(np.float64).T
    bound = d * 2.0 ** -24 * 1.0001          # |q| = |c| = 1 up to fp32 rounding of the normalisation
    under = 0
    for q in range(Q.shape[0]):
        order = np.argsort(-S64[q], kind="stable")[: k + 1]
        top = S64[q][order]
        assert np.abs(s[q].astype(np.float64) - S64[q][i[q]]).max() <= bound
        gaps = top[:-1] - top[1:]             # gaps[j] = score(rank j) - score(rank j + 1)
        if gaps[k - 1] > 2 * bound:
            assert set(i[q].tolist()) == set(order[:k].tolist())
        else:
            under += 1
        for j in range(k):
            if (j == 0 or gaps[j - 1] > 2 * bound) and gaps[j] > 2 * bound:
                assert i[q][j] == order[j]
    print(f"d={d} N={N}: {under} of {Q.shape[0]} queries have a k/(k+1) fp64 gap under {2 * bound:.2e}")
