"""Host-side descriptor builder (C++ in libsdag_hip.so, no GPU work) against the reference's masks."""
import numpy as np
import pytest

from helpers import (H, check_schedule, expand_descriptor, golden_bits, lane_mask_to_rows, load_masks, mask_sha,
                     synthetic_layout)


def test_descriptor_expands_to_reference_masks_bit_exact():
    import sdag_amd

    g = load_masks()
    for c in g["small"]:
        d = sdag_amd.from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        assert np.array_equal(expand_descriptor(d.host), golden_bits(c)), c
        ref = golden_bits(c)
        assert d.allowed_pairs == int(ref.sum())
        assert d.n_uniform_rows == int((ref.sum(1) == 0).sum())
    for c in g["big"]:
        d = sdag_amd.from_spans(c["L"], c["sys"], c["spans"], c["qa"], c["nbr"])
        assert mask_sha(expand_descriptor(d.host)) == c["sha256"], c["name"]
        assert d.allowed_pairs == c["true_count"] and d.n_uniform_rows == c["empty_rows"]


def test_unit_structure():
    import sdag_amd

    spans, qa, L = synthetic_layout(60, [300] * 10, 1, 30)
    for T in (0, 4, 8, 1000):
        d = sdag_amd.from_spans(L, 60, spans, qa, None, max_items_per_unit=T)
        w = d.host
        units = w[w[H["OFF_UNITS"]]: w[H["OFF_UNITS"]] + 8 * w[H["NUNITS"]]].reshape(-1, 8)
        lens = units[:, 2] - units[:, 1]
        assert (lens > 0).all() and (np.diff(lens) <= 0).all()          # longest units first
        if T > 0:
            assert lens.max() <= T
        split = w[w[H["OFF_SPLIT"]]: w[H["OFF_SPLIT"]] + 4 * w[H["NSPLIT"]]].reshape(-1, 4)
        slots = sorted(int(s) for s in units[:, 3] if s >= 0)
        assert slots == list(range(int(w[H["NSLOTS"]])))                  # every partial slot used once
        for row0, first, n, rows in split:
            mine = units[(units[:, 7] == row0)]
            assert len(mine) == n and sorted(mine[:, 3]) == list(range(first, first + n)) and 0 < rows <= 32
        uni = w[w[H["OFF_UNIFORM"]]: w[H["OFF_UNIFORM"]] + w[H["NUNIFORM"]]]
        assert list(uni) == [s[1] for s in spans]                          # the separator after each document
        assert np.array_equal(expand_descriptor(w), expand_descriptor(sdag_amd.from_spans(L, 60, spans, qa, None).host))


def test_tiles_follow_the_documents():
    """Row blocks start with the segments of the prompt and key tiles with the first key a block may attend: fewer
    tiles than a grid aligned to multiples of 32 rows / 64 keys (516 for this layout), same relation."""
    import sdag_amd
    from oracle import sdag_ref

    spans, qa, L = synthetic_layout(60, [300] * 10, 1, 30)
    d = sdag_amd.from_spans(L, 60, spans, qa, None)
    w = d.host
    rbinfo = w[w[H["OFF_RBINFO"]]: w[H["OFF_RBINFO"]] + 2 * w[H["NRB"]]].reshape(-1, 2)
    starts = set(int(x) for x in rbinfo[:, 1])
    assert {0, 60, qa} | {s for s, _ in spans} | {e for _, e in spans} <= starts
    assert d.n_items == 451
    assert np.array_equal(expand_descriptor(w), sdag_ref.mask_from_spans(L, 60, spans, qa, None))
    # a mask given as a dense array has no segments: blocks of 32 rows from row 0
    dd = sdag_amd.from_dense(sdag_ref.mask_from_spans(L, 60, spans, qa, None))
    rb2 = dd.host[dd.host[H["OFF_RBINFO"]]: dd.host[H["OFF_RBINFO"]] + 2 * dd.host[H["NRB"]]].reshape(-1, 2)
    assert list(rb2[:, 1]) == list(range(0, L, 32))
    assert np.array_equal(expand_descriptor(dd.host), expand_descriptor(w))


def test_from_dense_round_trip_arbitrary_masks():
    import sdag_amd

    rng = np.random.default_rng(5)
    for L in (1, 31, 32, 33, 64, 65, 130, 257):
        m = rng.random((L, L)) < 0.3
        m[rng.integers(0, L)] = False
        d = sdag_amd.from_dense(m)
        assert np.array_equal(expand_descriptor(d.host), m)
    full = np.ones((128, 128), bool)
    d = sdag_amd.from_dense(full)
    assert d.n_partial_tiles == 0 and d.n_items == 4 * 2


def test_invalid_arguments_raise():
    import sdag_amd

    with pytest.raises(ValueError):
        sdag_amd.from_spans(10, 11, [], 0)            # sys_user_len > L (the reference would index past the tensor)
    with pytest.raises(ValueError):
        sdag_amd.from_spans(10, 2, [(3, 12)], 5)      # span past L
    with pytest.raises(ValueError):
        sdag_amd.from_spans(10, 2, [(-1, 4)], 5)      # negative position (the reference would wrap around)
    with pytest.raises(ValueError):
        sdag_amd.from_spans(10, 2, [], -1)
    d = sdag_amd.from_spans(0, 0, [], 0)
    assert d.n_units == 0 and d.L == 0


def test_streams_cover_every_unit_once_and_lane_masks_match():
    import sdag_amd

    K = 20
    spans, qa, L = synthetic_layout(60, [380] * K, 1, 30)
    nbr = [[(i + 1) % K, (i + 2) % K] for i in range(K)]
    for n_streams, layout in ((0, (L, 60, spans, qa, nbr)), (7, (L, 60, spans, qa, None)), (3, (40, 5, [(6, 20)], 30, None))):
        d = sdag_amd.from_spans(*layout, n_streams=n_streams)
        w = d.host
        ns = int(w[H["NSTREAMS"]])
        assert ns == (n_streams or 64) and w[H["NSTEPS"]] == w[H["NITEMS"]]
        assert w[H["OFF_LMASKS"]] % 16 == 0 and w[H["OFF_STEPS"]] % 8 == 0
        seen, counts = check_schedule(w, int(w[H["OFF_STEPS"]]), int(w[H["OFF_STREAMS"]]), ns, lambda b: (w, 0))
        assert len(seen) == w[H["NUNITS"]] and counts.sum() == w[H["NITEMS"]]
        if w[H["NUNITS"]] >= ns:
            units = w[w[H["OFF_UNITS"]]: w[H["OFF_UNITS"]] + 8 * w[H["NUNITS"]]].reshape(-1, 8)
            longest = int((units[:, 2] - units[:, 1]).max())
            assert counts.max() - counts.min() <= longest + 2       # greedy dealing keeps the streams level
        masks = w[w[H["OFF_MASKS"]]: w[H["OFF_MASKS"]] + 64 * w[H["NMASKS"]]].view(np.uint64).reshape(-1, 32)
        lmasks = w[w[H["OFF_LMASKS"]]: w[H["OFF_LMASKS"]] + 64 * w[H["NMASKS"]]].view(np.uint64).reshape(-1, 32)
        for m in range(0, len(masks), max(1, len(masks) // 25)):
            assert np.array_equal(lane_mask_to_rows(lmasks[m]), masks[m])


def test_batch_table_streams():
    import sdag_amd

    a = sdag_amd.from_spans(*((lambda s, q, L: (L, 10, s, q))(*synthetic_layout(10, [70, 33], 2, 9))))
    b = sdag_amd.from_spans(*((lambda s, q, L: (L, 4, s, q))(*synthetic_layout(4, [100] * 3, 1, 40))), max_items_per_unit=2)
    bt = sdag_amd.BatchDescriptor([a, b, a], n_streams=5)
    w = bt.host
    B, ns = int(w[2]), int(w[14])
    assert B == 3 and ns == 5
    doff = w[w[7]: w[7] + B]
    assert doff[0] == doff[2] != doff[1] and (doff % 16 == 0).all()
    descs = [a, b, a]
    seen, counts = check_schedule(w, int(w[17]), int(w[16]), ns, lambda i: (descs[i].host, int(doff[i])))
    assert len(seen) == 2 * a.n_units + b.n_units and counts.sum() == w[15] == 2 * a.n_items + b.n_items
    steps = w[w[17]: w[17] + 8 * w[15]].reshape(-1, 8)
    slot_base = w[w[8]: w[8] + B]
    gl = sorted(int(s[6]) for s in steps if (s[2] & 2) and s[6] >= 0)
    assert gl == list(range(int(w[6])))                                # every global partial slot exactly once
    assert list(slot_base) == [0, a.n_slots, a.n_slots + b.n_slots]


def test_batch_table_stream_halves():
    """A batch with many units per stream: every sequence's units are cut in two by the keys they end at, one part
    per half of the streams; every unit still appears exactly once and the streams carry about equal work."""
    import sdag_amd

    spans, qa, L = synthetic_layout(60, [300] * 10, 1, 30)
    d = sdag_amd.from_spans(L, 60, spans, qa, max_items_per_unit=64)
    B, ns = 6, 8
    bt = sdag_amd.BatchDescriptor([d] * B, n_streams=ns)
    w = bt.host
    assert int(w[2]) == B and int(w[14]) == ns and d.n_units * B >= 8 * ns
    doff = w[w[7]: w[7] + B]
    seen, counts = check_schedule(w, int(w[17]), int(w[16]), ns, lambda i: (d.host, int(doff[i])))
    assert len(seen) == B * d.n_units and counts.sum() == w[15] == B * d.n_items
    so = w[int(w[16]): int(w[16]) + ns + 1]
    steps = w[w[17]: w[17] + 8 * w[15]].reshape(-1, 8)
    cost = np.zeros(ns, dtype=np.int64)
    ends = {}  # (sequence, half of the streams) -> first keys of the last tiles of its units
    for s in range(ns):
        r = steps[so[s]: so[s + 1]]
        cost[s] = 4 * len(r) + 5 * int((r[:, 2] & 1).sum())     # the builder's measure: a unit = 1.25 steps
        for st in r[(r[:, 2] & 2) != 0]:
            ends.setdefault((int(st[3]), s >= ns // 2), []).append(int(st[0]))
    for b in range(B):
        lo, hi = ends[(b, False)], ends[(b, True)]
        assert max(lo) <= min(hi) or max(hi) <= min(lo)          # the two halves split the sequence by key range
    assert cost.max() <= 1.05 * cost.mean()                       # levelled inside each half
    # ... and inside a stream the units stay in the order they were dealt in: sequence by sequence
    for s in range(ns):
        seqs = steps[so[s]: so[s + 1], 3]
        assert (np.diff(seqs) >= 0).all()


def test_random_layouts_expand_to_the_oracle_mask():
    """Property test: for random prompts (documents of 0-120 tokens, gaps, an overlapping pair of spans, neighbour lists
    with out-of-range and repeated ids or of the wrong length, which the reference ignores: SDAG.py:111), random unit
    length limits and stream counts, the descriptor decoded from its word layout equals the oracle's mask bit for bit,
    its streams cover every item once and a batch table of it is consistent."""
    from hypothesis import given, settings, strategies as st

    import sdag_amd
    from oracle import sdag_ref

    @st.composite
    def prompts(draw):
        n_docs = draw(st.integers(0, 7))
        sys_len = draw(st.integers(0, 70))
        lens = [draw(st.one_of(st.just(0), st.integers(1, 120))) for _ in range(n_docs)]
        gaps = [draw(st.integers(0, 3)) for _ in range(n_docs)]
        spans, pos = [], sys_len
        for n, g in zip(lens, gaps):
            spans.append((pos, pos + n))
            pos += n + g
        qa = pos
        L = max(1, qa + draw(st.integers(0, 40)))
        qa = min(qa, L)
        if n_docs >= 2 and draw(st.booleans()):
            a, b = spans[0]
            spans[1] = (a + (b - a) // 2, max(spans[1][1], a + (b - a) // 2))
        kind = draw(st.integers(0, 3))
        nbr = None
        if kind:
            nbr = [[draw(st.integers(-2, n_docs + 1)) for _ in range(draw(st.integers(0, 3)))] for _ in range(n_docs)]
            if kind == 3 and n_docs:
                nbr = nbr[:-1]                      # wrong length: ignored wholesale
        return L, sys_len, spans, qa, nbr, draw(st.integers(0, 5)), draw(st.sampled_from([0, 0, 1, 3, 64, 200]))

    @settings(max_examples=120, deadline=None, derandomize=True, database=None)
    @given(prompts())
    def check(p):
        L, sys_len, spans, qa, nbr, T, ns = p
        d = sdag_amd.from_spans(L, sys_len, spans, qa, nbr, max_items_per_unit=T, n_streams=ns)
        ref = sdag_ref.mask_from_spans(L, sys_len, spans, qa, nbr)
        assert np.array_equal(expand_descriptor(d.host), ref)
        assert d.allowed_pairs == int(ref.sum()) and d.n_uniform_rows == int((ref.sum(1) == 0).sum())
        w = d.host
        seen, counts = check_schedule(w, int(w[H["OFF_STEPS"]]), int(w[H["OFF_STREAMS"]]), int(w[H["NSTREAMS"]]), lambda i: (w, 0))
        assert len(seen) == d.n_units and counts.sum() == d.n_items
        many = sdag_amd.from_spans_many([(L, sys_len, spans, qa, nbr)] * 3, max_items_per_unit=T, n_streams=ns)
        assert all(np.array_equal(m.host, w) for m in many)   # the threaded builder gives the same words

    check()


def test_descriptor_builder_under_address_and_undefined_behaviour_sanitizers():
    """``make asan`` in csrc/: the host-side builder and a C++ driver (desc_fuzz.cpp: random prompts into buffers of
    exactly the requested size, expansion compared with the reference's rule, batch tables, refused arguments) built
    with g++ -fsanitize=address,undefined and run; any finding aborts the run."""
    import os
    import shutil
    import subprocess

    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no host compiler")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sparse-document-attention-rag-_amd", "csrc")
    out = subprocess.run(["make", "-C", csrc, "asan", "ASAN_CASES=150"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "desc_fuzz ok" in out.stdout
