"""Shared test helpers (pure Python / numpy; no product code paths)."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

H = dict(L=2, NRB=3, NKT=4, NUNITS=5, NITEMS=6, NMASKS=7, NSPLIT=8, NUNIFORM=9, OFF_UNITS=10, OFF_ITEMS=11,
         OFF_MASKS=12, OFF_RBINFO=13, OFF_SPLIT=14, OFF_UNIFORM=15, TOTAL=16, NSLOTS=17, NSTREAMS=20, NSTEPS=21,
         OFF_LMASKS=22, OFF_STREAMS=23, OFF_STEPS=24)


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
    nrb = int(w[H["NRB"]])
    rbinfo = w[w[H["OFF_RBINFO"]]: w[H["OFF_RBINFO"]] + 2 * nrb].reshape(-1, 2)
    starts = [int(x) for x in rbinfo[:, 1]] + [L]
    assert starts[0] == 0 or nrb == 0
    assert all(0 < b - a <= 32 for a, b in zip(starts[:-1], starts[1:]))       # the row blocks partition [0, L)
    for rb, b, e, slot, skip, key00, mi0, row0 in units.tolist():
        assert (key00, mi0) == tuple(items[b]) and row0 == starts[rb] and skip == rbinfo[rb][0]
        rr = starts[rb + 1] - row0
        assert (int(skip) & 0xffffffff) >> rr == (0xffffffff >> rr)          # rows beyond the block are skipped
        for it in range(b, e):
            assert it not in seen
            seen.add(it)
            c0, mi = items[it]
            r0 = row0
            assert it == b or c0 >= items[it - 1][0] + 64                     # tiles of a block do not overlap
            if mi < 0:
                assert c0 + 64 <= L
                out[r0:r0 + rr, c0:c0 + 64] = True
            else:
                bits = ((masks[mi][:, None] >> cols[None, :]) & np.uint64(1)).astype(bool)
                cc = min(64, L - c0)
                assert bits.any() and not bits[:rr, :cc].all() if cc == 64 else bits.any()
                assert bits[:, 0].any()                                       # a tile starts at a key some row attends
                assert not bits[rr:].any() and not bits[:, cc:].any()
                out[r0:r0 + rr, c0:c0 + cc] = bits[:rr, :cc]
    assert len(seen) == int(w[H["NITEMS"]])
    return out


def lane_mask_to_rows(lm):
    """Undo the per-accumulator-register regrouping of a 32 x 64 bit tile mask (32 uint64 in, 32 uint64 out)."""
    rows = np.zeros(32, dtype=np.uint64)
    for kb in range(2):
        for i in range(16):
            w = int(lm[16 * kb + i])
            for h in range(2):
                key = 32 * kb + (i & 3) + 8 * (i >> 2) + 4 * h
                for r in range(32):
                    if (w >> (r + 32 * h)) & 1:
                        rows[r] |= np.uint64(1 << key)
    return rows


def check_schedule(words, steps_off, streams_off, n_streams, units_of, base_of=None):
    """Walk the streams of a descriptor or batch table: every unit must appear exactly once, as a contiguous run
    of steps that repeats its items in order.  units_of(b) -> (descriptor words, word offset of that descriptor in
    the table); returns the per-stream step counts."""
    w = np.asarray(words, dtype=np.int32)
    so = w[streams_off: streams_off + n_streams + 1]
    assert so[0] == 0 and (np.diff(so) >= 0).all()
    steps = w[steps_off: steps_off + 8 * so[-1]].reshape(-1, 8)
    seen = set()
    for s in range(n_streams):
        i = so[s]
        while i < so[s + 1]:
            kt, moff, flags, b, L, rb, slot, ubits = steps[i]
            assert flags & 1
            d, doff = units_of(int(b))
            units = d[d[H["OFF_UNITS"]]: d[H["OFF_UNITS"]] + 8 * d[H["NUNITS"]]].reshape(-1, 8)
            items = d[d[H["OFF_ITEMS"]]: d[H["OFF_ITEMS"]] + 2 * d[H["NITEMS"]]].reshape(-1, 2)
            cand = [u for u in units if u[7] == rb and items[u[1]][0] == kt]
            assert len(cand) == 1
            u = cand[0]
            key = (int(b), int(u[1]))
            assert key not in seen
            seen.add(key)
            n = int(u[2] - u[1])
            for j in range(n):
                st = steps[i + j]
                assert st[0] == items[u[1] + j][0] and st[3] == b and st[4] == d[H["L"]] and st[5] == rb == u[7]
                assert st[7] == u[4]
                assert (st[2] & 1) == (j == 0) and bool(st[2] & 2) == (j == n - 1)
                mi = items[u[1] + j][1]
                if mi < 0:
                    assert st[1] == 0
                else:
                    # the step's lane mask: 64 words at a 64-byte boundary of the table that equal lane mask mi of the
                    # sequence (its own copy in a single descriptor; one pooled copy per distinct mask in a batch)
                    assert st[1] % 16 == 0 and 0 < st[1] and st[1] + 64 <= len(w)
                    own = d[H["OFF_LMASKS"]] + 64 * mi
                    assert np.array_equal(w[st[1]: st[1] + 64], d[own: own + 64])
            i += n
    return seen, np.diff(so)


def synthetic_layout(sys_len, doc_lens, gap, qa_len):
    spans, pos = [], sys_len
    for n in doc_lens:
        spans.append((pos, pos + n))
        pos += n + gap
    return spans, pos, pos + qa_len


LLAMA3_SPLIT = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|"
                r"\s+(?!\S)|\s+")


def make_bpe_tokenizer(seed=0, vocab=700):
    """A byte-level BPE fast tokenizer of the Llama-3 kind (same split pattern, BOS from the post-processor, header
    / end-of-turn special tokens, a chat template), trained in memory on synthetic text: no file, no download."""
    import random

    from tokenizers import Regex, Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast

    rng = random.Random(seed)
    words = ["".join(rng.choice("abcdefghijklmnopqrstuvwxyzéüñ") for _ in range(rng.randint(1, 8))) for _ in range(300)]
    ends = ["", "", "", ",", ".", "!", "?", ";", ":", "'s", "'re", "'", ")", "\n", "\n\n", "..", " -", ".\n", "!\n\n", "  "]

    def sentence(n_lo=3, n_hi=20):
        toks = []
        for _ in range(rng.randint(n_lo, n_hi)):
            w = rng.choice(words)
            r = rng.random()
            if r < 0.1:
                w = w.capitalize()
            if r > 0.88:
                w = str(rng.randint(0, 999999))
            toks.append(w + rng.choice(ends))
        return " ".join(toks)

    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.Sequence([
        pre_tokenizers.Split(Regex(LLAMA3_SPLIT), behavior="isolated", invert=False),
        pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=False)])
    tk.decoder = decoders.ByteLevel()
    specials = ["<|begin_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>"]
    tr = trainers.BpeTrainer(vocab_size=vocab, special_tokens=specials, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                             show_progress=False)
    tk.train_from_iterator([sentence() for _ in range(3000)], tr)
    tk.post_processor = processors.TemplateProcessing(
        single="<|begin_of_text|> $A", special_tokens=[("<|begin_of_text|>", tk.token_to_id("<|begin_of_text|>"))])
    fast = PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<|begin_of_text|>", eos_token="<|eot_id|>")
    fast.chat_template = (
        "{% for m in messages %}<|start_header_id|>{{ m['role'] }}<|end_header_id|>\n\n{{ m['content'] }}<|eot_id|>{% endfor %}"
        "{% if add_generation_prompt %}<|start_header_id|>assistant<|end_header_id|>\n\n{% endif %}")
    return fast, sentence


class FakeRanker:
    """Deterministic stand-in for the sentence encoder: unit vectors seeded by a hash of the text (the interface
    the pipeline uses: ``encode(list, convert_to_tensor=True, normalize_embeddings=True)``)."""

    def __init__(self, d=64, device="cpu"):
        self.d, self.device = d, device

    def get_sentence_embedding_dimension(self):
        return self.d

    def encode(self, texts, convert_to_tensor=True, normalize_embeddings=True, batch_size=32, **kw):
        import torch

        rows = []
        for t in texts:
            seed = int.from_bytes(hashlib.sha256(t.encode()).digest()[:4], "little")
            v = np.random.default_rng(seed).standard_normal(self.d).astype(np.float32)
            rows.append(v / np.linalg.norm(v))
        out = torch.from_numpy(np.stack(rows)) if rows else torch.zeros((0, self.d))
        return out.to(self.device)


def load_e2e():
    """The end-to-end fixture made by tests/golden/make_golden_e2e.py from the reference's own entry points:
    (tokenizer rebuilt from its stored serialisation, fixture dict, path of the model weights)."""
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast

    with open(os.path.join(GOLDEN, "e2e.json")) as f:
        g = json.load(f)
    tok = PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_str(g["tokenizer_json"]), bos_token=g["bos_token"],
                                  eos_token=g["eos_token"])
    tok.chat_template = g["chat_template"]
    return tok, g, os.path.join(GOLDEN, "e2e_model.npz")


def load_e2e_model(g, path, device="cpu", dtype=None):
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM

    z = np.load(path)
    model = LlamaForCausalLM(LlamaConfig(**g["model_cfg"])).eval()
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")})
    if dtype is not None:
        model = model.to(dtype)
    return model.to(device), {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("logits::")}
