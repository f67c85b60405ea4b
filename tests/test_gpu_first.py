"""First GPU parity checks: descriptor expansion, attention vs the fp32 oracle, exact search vs the C oracle."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN, golden_bits, load_masks, mask_sha, synthetic_layout

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_mask_materialize_matches_golden():
    import sdag_amd

    g = load_masks()
    for c in g["small"]:
        d = sdag_amd.from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        m = d.materialize(_dev()).cpu().numpy()
        assert np.array_equal(m, golden_bits(c)), c
    for c in g["big"]:
        d = sdag_amd.from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        m = d.materialize(_dev()).cpu().numpy()
        assert mask_sha(m) == c["sha256"], c["name"]


def _rel_err(o, ref):
    # per head: max |o - ref| / max |ref|
    num = (o - ref).abs().amax(dim=(0, 1, 3))
    den = ref.abs().amax(dim=(0, 1, 3))
    return float((num / den).max())


@pytest.mark.parametrize("name", ["toy", "mid", "nbr"])
def test_attention_golden(name):
    import json

    import sdag_amd
    from oracle import sdag_ref

    z = np.load(f"{GOLDEN}/attention.npz")
    meta = json.loads(bytes(z[f"{name}_meta"]).decode())
    q = torch.from_numpy(z[f"{name}_q"]).view(torch.bfloat16)
    k = torch.from_numpy(z[f"{name}_k"]).view(torch.bfloat16)
    v = torch.from_numpy(z[f"{name}_v"]).view(torch.bfloat16)
    ref = torch.from_numpy(z[f"{name}_o_f32"])
    d = sdag_amd.from_spans(meta["L"], meta["sys"], meta["spans"], meta["qa"], meta["nbr"])
    o = sdag_amd.sdag_attention(q.to(_dev()), k.to(_dev()), v.to(_dev()), d, out_dtype=torch.float32).cpu()
    assert torch.isfinite(o).all()
    err = _rel_err(o, ref)
    print(name, "rel err fp32-out", err)
    assert err <= 1e-3
    ob = sdag_amd.sdag_attention(q.to(_dev()), k.to(_dev()), v.to(_dev()), d, out_dtype=torch.bfloat16).cpu()
    assert torch.equal(ob, o.to(torch.bfloat16))


def test_attention_cfg3_full_size_two_kv_heads():
    import sdag_amd
    from oracle import sdag_ref

    spans, qa, L = synthetic_layout(60, [300] * 10, 1, 30)
    Hq, Hkv, D = 8, 2, 128
    g = torch.Generator().manual_seed(0)
    q = torch.randn(1, Hq, L, D, generator=g).to(torch.bfloat16)
    k = torch.randn(1, Hkv, L, D, generator=g).to(torch.bfloat16)
    v = torch.randn(1, Hkv, L, D, generator=g).to(torch.bfloat16)
    mask = torch.from_numpy(sdag_ref.mask_from_spans(L, 60, spans, qa, None))
    ref = sdag_ref.masked_attention(q, k, v, mask, D ** -0.5)
    d = sdag_amd.from_spans(L, 60, spans, qa, None)
    o = sdag_amd.sdag_attention(q.to(_dev()), k.to(_dev()), v.to(_dev()), d, out_dtype=torch.float32).cpu()
    err = _rel_err(o, ref)
    print("cfg3 rel err", err)
    assert err <= 1e-3


@pytest.mark.parametrize("N,d,nq,k", [(1000, 64, 7, 5), (4096, 768, 32, 5), (20000, 1024, 32, 10), (333, 128, 40, 21),
                                      (7, 32, 3, 10)])
def test_dense_topk_bit_exact(N, d, nq, k):
    import sdag_amd
    from oracle import dense_ref

    rng = np.random.default_rng(N + d)
    C = rng.standard_normal((N, d), dtype=np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if N >= 1000:
        C[N // 2] = C[N // 3]  # an exact tie: the lower id must come first
    rs, ri = dense_ref.dense_topk(Q, C, k, id_base=5)
    for two_stage in (True, False):      # bf16 candidate scan + exact rescoring (d % 64 == 0), and the plain exact scan
        idx = sdag_amd.FlatIPIndex(d, device=_dev(), id_base=5, two_stage=two_stage)
        assert idx.two_stage == (two_stage and d % 64 == 0)
        idx.add(C)
        s, i = idx.search(Q, k)
        assert np.array_equal(i, ri)
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32))
        if N >= 20000:
            assert not idx.last_search_fell_back()   # a large ordinary corpus is answered from the candidates


@pytest.mark.parametrize("N,d,nq,k", [(30000, 128, 33, 10), (50000, 256, 100, 10), (20000, 64, 256, 5), (9000, 128, 1024, 10),
                                      (30000, 128, 100, 20), (5000, 64, 300, 32), (70000, 1024, 128, 13), (3, 64, 130, 4)])
def test_dense_topk_more_than_32_queries_per_pass(N, d, nq, k):
    """The reference hands the whole query matrix to one index.search (dense.py:82-84).  Beyond 32 queries the scan
    multiplies every staged corpus tile against 64 or 128 queries (128 only while the stripe lists hold <= 16
    entries, else 64): ids and score bits equal to the oracle for both the two-stage and the exact search, with exact
    ties, query counts that do not fill the last pass, and a corpus smaller than k."""
    import sdag_amd
    from oracle import dense_ref

    rng = np.random.default_rng(N + nq)
    C = rng.standard_normal((N, d), dtype=np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if N >= 1000:
        C[N // 2] = C[N // 3]
        C[N - 1] = C[0]
        Q[nq - 1] = C[N // 3]       # the tied rows lead the last query's list
    rs, ri = dense_ref.dense_topk(Q, C, k, id_base=11)
    for two_stage in (True, False):
        idx = sdag_amd.FlatIPIndex(d, device=_dev(), id_base=11, two_stage=two_stage)
        idx.add(C)
        s, i = idx.search(Q, k)
        assert np.array_equal(i, ri), (two_stage, np.argwhere(i != ri)[:5])
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)), two_stage
        # the same queries 32 at a time (the narrow scan) give the same bits
        s32 = np.concatenate([idx.search(Q[a:a + 32], k)[0] for a in range(0, nq, 32)])
        assert np.array_equal(s32.view(np.uint32), s.view(np.uint32))


@pytest.mark.parametrize("N,d,nq,k", [(30000, 128, 9, 33), (20000, 64, 40, 64), (6000, 256, 32, 100), (50, 64, 5, 70),
                                      (9000, 96, 70, 45)])
def test_dense_topk_any_k_page_by_page(N, d, nq, k):
    """faiss answers index.search(q, top_k) for any top_k (dense.py:82-84); one kernel call here returns 32 entries per
    query, the rest comes page by page from the exact scan with the previous page's last entry as the cursor: ids and
    score bits equal to the oracle's top k, across exact ties that straddle a page boundary, for a corpus smaller than
    k (fill), and for a cursor given directly."""
    import sdag_amd
    from oracle import dense_ref

    rng = np.random.default_rng(N + k)
    C = rng.standard_normal((N, d), dtype=np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    if N >= 1000:
        # query 0: rows 500..539 are 40 exact copies of one vector and its best matches -> ties across the 32 | 33 boundary
        C[500:540] = Q[0]
        C[N - 1] = Q[0]
    rs, ri = dense_ref.dense_topk(Q, C, k, id_base=7)
    for two_stage in (True, False):
        idx = sdag_amd.FlatIPIndex(d, device=_dev(), id_base=7, two_stage=two_stage)
        idx.add(C)
        s, i = idx.search(Q, k)
        assert s.shape == (nq, k) and i.shape == (nq, k)
        assert np.array_equal(i, ri), (two_stage, np.argwhere(i != ri)[:5])
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)), two_stage
    # a page from a cursor in the middle of the list: ranks 6 .. 6 + 20 after entry 5
    if N >= 1000:
        dev = _dev()
        ps, pi = idx.search_after(torch.from_numpy(Q).to(dev), 20, torch.from_numpy(rs[:, 5].copy()).to(dev),
                                  torch.from_numpy(ri[:, 5].copy()).to(dev))
        assert np.array_equal(pi.cpu().numpy(), ri[:, 6:26])
        assert np.array_equal(ps.cpu().numpy().view(np.uint32), rs[:, 6:26].view(np.uint32))


def test_threshold_scan_on_hostile_corpora():
    """More than 32 queries over a corpus large enough for the threshold scan (thresholds from a sample, every row at
    or above them a candidate, exact rescoring of the ones near the k-th best): corpora built against it -- thousands of
    rows within rounding distance of a query's best score (the candidate lists overflow), a block of exact duplicates
    that the sample never sees (tiles the pilot skips), rows with wildly different norms, a query equal to a corpus row,
    all-zero queries -- must still give the oracle's ids and score bits, through the exact scan where the candidate set
    cannot be vouched for."""
    import sdag_amd
    from oracle import dense_ref

    rng = np.random.default_rng(23)
    d, nq, k, N = 128, 70, 10, 60000
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    fell = []
    for case in range(3):
        C = rng.standard_normal((N, d), dtype=np.float32)
        C /= np.linalg.norm(C, axis=1, keepdims=True)
        if case == 0:
            C[130:160] = Q[1]                        # 30 exact duplicates inside a tile the pilot does not sample
            C[40000] = Q[2]
            Q[3] = C[777]
        elif case == 1:
            C[1000:7000] = Q[0] * 0.9 + 1e-4 * rng.standard_normal((6000, d), dtype=np.float32)   # 6000 near-ties
        else:
            C *= np.exp(rng.uniform(-5, 5, (N, 1))).astype(np.float32)
            Q[5] = 0.0                               # every row scores 0: all of them are candidates
        rs, ri = dense_ref.dense_topk(Q, C, k, id_base=3)
        idx = sdag_amd.FlatIPIndex(d, device=_dev(), id_base=3)
        idx.add(C)
        s, i = idx.search(Q, k)
        assert np.array_equal(i, ri), (case, np.argwhere(i != ri)[:5])
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32)), case
        fell.append(idx.last_search_fell_back())
    assert not fell[0] and fell[1]            # the flood of near-ties is answered by the exact scan


def test_two_stage_search_when_candidates_abound():
    """Corpora built against the candidate scan: hundreds of rows within rounding distance of the best score (more
    candidates than the rescoring kernel holds, and whole stripes full of them), exact duplicates, a corpus smaller
    than k, and rows with wildly different norms.  Results must still equal the exact oracle bit for bit."""
    import sdag_amd
    from oracle import dense_ref

    rng = np.random.default_rng(11)
    d, nq, k = 128, 9, 10
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    Q /= np.linalg.norm(Q, axis=1, keepdims=True)
    cases = []
    C = rng.standard_normal((6000, d), dtype=np.float32)
    C /= np.linalg.norm(C, axis=1, keepdims=True)
    C[100:900] = Q[0] * 0.9 + 1e-4 * rng.standard_normal((800, d), dtype=np.float32)     # 800 near-ties for query 0
    C[2000:2040] = Q[1]                                                                  # 40 exact duplicates
    cases.append(C)
    cases.append(C[:6].copy())                                                           # fewer rows than k
    C2 = rng.standard_normal((3000, d), dtype=np.float32) * np.exp(rng.uniform(-6, 6, (3000, 1))).astype(np.float32)
    cases.append(C2)
    fell_back = []
    for C in cases:
        rs, ri = dense_ref.dense_topk(Q, C, k, id_base=0)
        idx = sdag_amd.FlatIPIndex(d, device=_dev())
        idx.add(C)
        s, i = idx.search(Q, k)
        assert np.array_equal(i, ri)
        assert np.array_equal(s.view(np.uint32), rs.view(np.uint32))
        fell_back.append(idx.last_search_fell_back())
    assert fell_back[0] and not fell_back[1]         # 800 near-ties: the exact scan had to answer


def test_attention_large_values_in_v():
    """V magnitudes beyond the fp16 range (the P V product runs on fp16 operands) must still be exact to 1e-3."""
    import sdag_amd
    from oracle import sdag_ref

    spans, qa, L = synthetic_layout(11, [70, 90, 50], 1, 20)
    g = torch.Generator().manual_seed(3)
    q = torch.randn(1, 4, L, 128, generator=g).to(torch.bfloat16)
    k = torch.randn(1, 1, L, 128, generator=g).to(torch.bfloat16)
    v = torch.randn(1, 1, L, 128, generator=g)
    v[:, :, 64:128] *= 3.0e4      # one tile reaches ~1.2e5 > 65504
    v[:, :, 130:140] *= 1.0e8     # and a few rows far beyond
    v = v.to(torch.bfloat16)
    mask = torch.from_numpy(sdag_ref.mask_from_spans(L, 11, spans, qa, None))
    ref = sdag_ref.masked_attention(q, k, v, mask, 128 ** -0.5)
    d = sdag_amd.from_spans(L, 11, spans, qa, None)
    o = sdag_amd.sdag_attention(q.to(_dev()), k.to(_dev()), v.to(_dev()), d, out_dtype=torch.float32).cpu()
    assert torch.isfinite(o).all()
    # per row: rows that never see the huge values must not be polluted by them
    num = (o - ref).abs().amax(dim=3)
    den = ref.abs().amax(dim=3)
    rel = float((num / den).max())
    print("large-V per-row rel err", rel)
    assert rel <= 1e-3
