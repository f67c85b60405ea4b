"""Shared test helpers (pure Python / numpy; no product code paths)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

H = dict(L=2, NRB=3, NKT=4, NUNITS=5, NITEMS=6, NMASKS=7, NSPLIT=8, NUNIFORM=9, OFF_UNITS=10, OFF_ITEMS=11,
         OFF_MASKS=12, OFF_RBINFO=13, OFF_SPLIT=14, OFF_UNIFORM=15, TOTAL=16, NSLOTS=17)


def load_masks():
    with open(os.path.join(GOLDEN, "masks.json")) as f:
        return json.load(f)


def golden_bits(case):
    L = case["L"]
    raw = np.frombuffer(bytes.fromhex(case["bits"]), dtype=np.uint8)
    return np.unpackbits(raw)[: L * L].reshape(L, L).astype(bool)


def mask_sha(m):
    return hashlib.sha256(np.packbits(np.asarray(m, dtype=bool)).tobytes()).hexdigest()


def expand_descriptor(words):
    """Dense bool [L, L] image of a descriptor, decoded in Python straight from the documented word layout."""
    w = np.asarray(words, dtype=np.int32)
    L = int(w[H["L"]])
    out = np.zeros((L, L), dtype=bool)
    units = w[w[H["OFF_UNITS"]]: w[H["OFF_UNITS"]] + 8 * w[H["NUNITS"]]].reshape(-1, 8)
    items = w[w[H["OFF_ITEMS"]]: w[H["OFF_ITEMS"]] + 2 * w[H["NITEMS"]]].reshape(-1, 2)
    masks = w[w[H["OFF_MASKS"]]: w[H["OFF_MASKS"]] + 64 * w[H["NMASKS"]]].view(np.uint64).reshape(-1, 32)
    cols = np.arange(64, dtype=np.uint64)
    seen = set()
    for rb, b, e, slot, uni, kt0, mi0, _ in units:
        assert (kt0, mi0) == tuple(items[b])
        for it in range(b, e):
            assert it not in seen
            seen.add(it)
            kt, mi = items[it]
            r0, c0 = rb * 32, kt * 64
            if mi < 0:
                assert r0 + 32 <= L and c0 + 64 <= L
                out[r0:r0 + 32, c0:c0 + 64] = True
            else:
                bits = ((masks[mi][:, None] >> cols[None, :]) & np.uint64(1)).astype(bool)
                assert bits.any() and not bits.all()
                rr, cc = min(32, L - r0), min(64, L - c0)
                assert not bits[rr:].any() and not bits[:, cc:].any()
                out[r0:r0 + rr, c0:c0 + cc] = bits[:rr, :cc]
    assert len(seen) == int(w[H["NITEMS"]])
    return out


def synthetic_layout(sys_len, doc_lens, gap, qa_len):
    spans, pos = [], sys_len
    for n in doc_lens:
        spans.append((pos, pos + n))
        pos += n + gap
    return spans, pos, pos + qa_len
