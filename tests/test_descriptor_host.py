"""Host-side descriptor builder (C++ in libsdag_hip.so, no GPU work) against the reference's masks."""
import numpy as np
import pytest

from helpers import H, expand_descriptor, golden_bits, load_masks, mask_sha, synthetic_layout


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
        for rb, first, n, _ in split:
            mine = units[(units[:, 0] == rb)]
            assert len(mine) == n and sorted(mine[:, 3]) == list(range(first, first + n))
        uni = w[w[H["OFF_UNIFORM"]]: w[H["OFF_UNIFORM"]] + w[H["NUNIFORM"]]]
        assert list(uni) == [s[1] for s in spans]                          # the separator after each document
        assert np.array_equal(expand_descriptor(w), expand_descriptor(sdag_amd.from_spans(L, 60, spans, qa, None).host))


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
