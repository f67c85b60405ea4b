"""SDAG document-isolated generation on MI355X behind the reference's module surface
(src/pipeline/sparse_attention_RAG/SDAG.py).

Same public names, arguments and return values: ``compute_doc_knn_for_docs``,
``build_blocked_causal_mask_full``, ``generate_with_custom_mask``, ``build_rag_chat_and_spans``,
``run_rag_with_doc_isolation``.  What changes is underneath: the L x L mask is never built on the prefill
path -- token spans become a block-sparse descriptor on the host in microseconds, and every attention layer of the
prefill runs the HIP kernel driven by it.  ``build_blocked_causal_mask_full`` still returns the dense boolean
tensor for callers that want it, expanded on the GPU from that same descriptor.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

import sdag_amd
from sdag_amd import hf_attention

try:  # frozen at import like the reference does (SDAG.py:9)
    from src.pipeline.config import MAX_GEN_TOKENS_RAG, TEMPERATURE
except Exception:  # pragma: no cover - standalone use
    MAX_GEN_TOKENS_RAG, TEMPERATURE = 500, 0.1
try:
    from src.pipeline.utils.prompts import SYSTEM_PROMPT_RAG, USER_RAG_PROMPT
except Exception:  # pragma: no cover - standalone use: the two templates the spans depend on (prompts.py:15,32-41)
    SYSTEM_PROMPT_RAG = "You are a helpful assistant, below is a query from a user and some relevant contexts."
    USER_RAG_PROMPT = ("Answer the question concisely, based on the following passages.\nKeep the answer concise.\n\n"
                       "passages:\n{docs_text}\n\n- Question: {query}\n\n- Answer:\n")
try:
    from src.pipeline.utils.ranked_list import inject_malicious_docs_into_ranked_list
except Exception:  # pragma: no cover - standalone use
    from sdag_amd.ranked_list import inject_malicious_docs_into_ranked_list


def _gpu_device(device) -> torch.device:
    dev = torch.device(device)
    if dev.type == "cuda":
        return dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cuda:0")  # kernels always run on the GPU; results are copied to `device` if it differs


def compute_doc_knn_for_docs(ranker, docs: List[str], k_neighbors: int) -> List[List[int]]:
    """k most similar other documents of each document, most similar first (SDAG.py:14-65).

    Empty / whitespace documents are left out of the comparison and get [] (SDAG.py:42-44,51); fewer than two
    non-empty documents, or k <= 0, give all-empty lists.  Texts are embedded with the "passage: " prefix
    (SDAG.py:47); the Gram matrix is computed on the GPU.
    """
    n = len(docs)
    if k_neighbors <= 0 or n == 0:
        return [[] for _ in range(n)]
    kept = [i for i, d in enumerate(docs) if d and d.strip()]
    if len(kept) <= 1:
        return [[] for _ in range(n)]
    emb = ranker.encode(["passage: " + docs[i] for i in kept], convert_to_tensor=True, normalize_embeddings=True)
    dev = emb.device if emb.is_cuda else torch.device("cuda:0")
    return sdag_amd.doc_knn(emb.float(), kept, n, k_neighbors, device=dev)


def build_blocked_causal_mask_full(base_seq_len: int, max_new_tokens: int, sys_user_len: int, doc_token_spans: list,
                                   qa_start: int, device: str = "cpu", doc_neighbors: Optional[List[List[int]]] = None):
    """Boolean [L, L] "row may attend column" tensor on ``device`` (SDAG.py:68-127).

    ``max_new_tokens`` is accepted and unused, as in the reference.  The returned tensor remembers the
    descriptor it was expanded from, so ``generate_with_custom_mask`` does not have to re-derive it from the
    dense tensor (it checks that the tensor still holds what the descriptor says).
    """
    desc = sdag_amd.from_spans(base_seq_len, sys_user_len, [tuple(s) for s in doc_token_spans], qa_start, doc_neighbors)
    mask = desc.materialize(_gpu_device(device))
    if mask.device != torch.device(device):
        mask = mask.to(device)
    mask._sdag_desc = desc
    mask._sdag_stamp = (mask._version, mask.data_ptr())  # in-place edits bump the version counter
    return mask


def _descriptor_of(prompt_mask: torch.Tensor, model_dtype: torch.dtype) -> Optional[sdag_amd.Descriptor]:
    desc = getattr(prompt_mask, "_sdag_desc", None)
    if desc is not None and desc.L == prompt_mask.shape[0] and prompt_mask.dtype == torch.bool:
        # The attribute survives in-place edits of the tensor (an ablation that clears a block, say), and the
        # reference always honours the tensor's contents (SDAG.py:167-171).  An untouched tensor (same storage, same
        # version counter as when it was built) is taken at its word; an edited one only if the remembered descriptor
        # still expands to exactly this mask (one L x L comparison on the GPU, with its host synchronisation).
        if getattr(prompt_mask, "_sdag_stamp", None) == (prompt_mask._version, prompt_mask.data_ptr()):
            return desc
        dev = _gpu_device(prompt_mask.device)
        if torch.equal(desc.materialize(dev), prompt_mask.to(dev)):
            return desc
    if prompt_mask.dtype == torch.bool:
        return sdag_amd.from_dense(prompt_mask)
    # "pre-computed attention matrix" (SDAG.py:170-171): usable only if it is a 0 / very-negative additive mask
    allowed = prompt_mask > torch.finfo(model_dtype).min / 2
    if bool(((prompt_mask == 0) | ~allowed).all()):
        return sdag_amd.from_dense(allowed)
    return None


def _sample(logits: torch.Tensor) -> torch.Tensor:
    if TEMPERATURE > 0:  # SDAG.py:182-187
        probs = torch.softmax(logits / TEMPERATURE, dim=-1)
        return torch.multinomial(probs, num_samples=1)
    return torch.argmax(logits, dim=-1, keepdim=True)


def _decode_loop(model, tokenizer, input_ids, first_out, max_new_tokens: int) -> str:
    """Sampling of the first token and the cached single-token steps, exactly as SDAG.py:178-213 (up to
    max_new_tokens + 1 tokens; EOS is only tested inside the loop)."""
    prompt_len = input_ids.size(1)
    past = first_out.past_key_values
    next_token = _sample(first_out.logits[:, -1, :])
    generated = torch.cat([input_ids, next_token], dim=1)
    for _ in range(max_new_tokens):
        with torch.no_grad():
            out = model(input_ids=generated[:, -1:], past_key_values=past, use_cache=True)
        past = out.past_key_values
        next_token = _sample(out.logits[:, -1, :])
        generated = torch.cat([generated, next_token], dim=1)
        if next_token.item() == tokenizer.eos_token_id:
            break
    return tokenizer.decode(generated[0][prompt_len:], skip_special_tokens=True).strip()


def decode_with_graph() -> bool:
    """SDAG_DECODE_GRAPH=1: the cached single-token steps of ``generate_with_custom_mask`` /
    ``run_rag_with_doc_isolation`` are replayed from one captured HIP graph (``_decode_loop_graphed``)."""
    return os.environ.get("SDAG_DECODE_GRAPH", "0") == "1"


def _replay_steps(dev, one_step, n_steps: int, after_step) -> None:
    """Run ``one_step`` n_steps times: the first call eagerly (it also warms up every lazily built object), the
    second under HIP graph capture, the rest as replays of that graph.  ``after_step()`` runs on the host after every
    step (it may read results back) and returns True to stop."""
    graph = None
    for step in range(n_steps):
        with torch.no_grad():
            if step == 0:
                one_step()
            else:
                if graph is None:
                    torch.cuda.synchronize(dev)
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        one_step()
                graph.replay()
        if after_step():
            break


def _static_cache_from(model, past, n_positions: int, capacity: int):
    """A preallocated cache of ``capacity`` positions holding the ``n_positions`` positions of ``past`` (the prefill's
    cache); transformers' StaticCache appends at a position it keeps on the device, so a captured step can be replayed."""
    from transformers import StaticCache

    cache = StaticCache(config=model.config, max_cache_len=capacity)
    for idx, layer in enumerate(past.layers):
        if layer.keys.shape[2] != n_positions:
            raise ValueError(f"layer {idx} of the prefill's cache holds {layer.keys.shape[2]} positions, expected {n_positions}")
        cache.update(layer.keys, layer.values, idx)
    return cache


def _decode_loop_graphed(model, tokenizer, input_ids, first_out, max_new_tokens: int) -> str:
    """``_decode_loop`` with the single-token step captured ONCE as a HIP graph and replayed: a step is some hundreds
    of small launches whose host cost exceeds their GPU time, and a replay is one launch.

    Same tokens and the same stopping rule as SDAG.py:178-213 (the end-of-sequence test reads the token back after
    every step, as the reference does).  What makes the step replayable: the cache is preallocated for
    prompt + max_new_tokens + 1 positions, the token fed in and the number of cached positions live in device tensors
    that the captured step itself updates, and the decode kernel reads that number on the device.
    """
    dev = input_ids.device
    if dev.type != "cuda" or input_ids.size(0) != 1:
        raise RuntimeError("graph-replayed decoding needs one prompt on a GPU")
    prompt_len = input_ids.size(1)
    cap = prompt_len + max_new_tokens + 1
    cache = _static_cache_from(model, first_out.past_key_values, prompt_len, cap)
    n_rows = torch.full((1,), prompt_len, dtype=torch.int32, device=dev)
    tok = _sample(first_out.logits[:, -1, :]).clone()
    tokens = [int(tok.item())]

    def one_step():
        n_rows.add_(1)  # the fed token's key and value are in the cache when a layer's attention runs
        out = model(input_ids=tok, past_key_values=cache, use_cache=True)
        tok.copy_(_sample(out.logits[:, -1, :]))

    def after_step():
        tokens.append(int(tok.item()))
        return tokens[-1] == tokenizer.eos_token_id

    with hf_attention.active_padding(None, 0, n_rows=n_rows):
        _replay_steps(dev, one_step, max_new_tokens, after_step)
    return tokenizer.decode(torch.tensor(tokens), skip_special_tokens=True).strip()


def _generate_with_descriptor(model, tokenizer, input_ids, desc, max_new_tokens: int) -> str:
    with hf_attention.use_on_model(model):
        with torch.no_grad(), hf_attention.active_descriptor(desc):
            out = model(input_ids=input_ids, use_cache=True)  # masked prefill: the kernel path
        if decode_with_graph():
            return _decode_loop_graphed(model, tokenizer, input_ids, out, max_new_tokens)
        return _decode_loop(model, tokenizer, input_ids, out, max_new_tokens)


def generate_with_custom_mask(model, tokenizer, input_ids: torch.Tensor, prompt_mask: torch.Tensor,
                              max_new_tokens: int = 128):
    """Prefill under ``prompt_mask`` (bool [L, L], True = attend), then ordinary cached decoding (SDAG.py:130-213)."""
    model_dtype = next(model.parameters()).dtype
    desc = _descriptor_of(prompt_mask, model_dtype)
    if desc is None:  # a general additive float mask: not expressible as allow / forbid, run it the stock way
        attn = prompt_mask.to(input_ids.device, dtype=model_dtype).unsqueeze(0).unsqueeze(1)
        with torch.no_grad():
            out = model(input_ids=input_ids, attention_mask=attn, use_cache=True)
        return _decode_loop(model, tokenizer, input_ids, out, max_new_tokens)
    return _generate_with_descriptor(model, tokenizer, input_ids, desc, max_new_tokens)


def _n_tokens(tokenizer, text: str) -> int:
    return len(tokenizer(text, return_tensors="pt").input_ids[0])


_TAIL_MARGIN = 8  # characters re-tokenised in front of a cut (the longest look-ahead of the split patterns is 3)


def _tokenizer_is_local(tokenizer) -> bool:
    """True for fast tokenizers whose text -> token map is a left-to-right scan without look-behind: no
    normaliser, a split-pattern / byte-level pre-tokeniser that adds no prefix space, BPE inside the pre-tokens
    (the Llama-3 family).  Only those may count prefix lengths from one tokenisation of the whole prompt."""
    cached = getattr(tokenizer, "_sdag_is_local", None)  # kept on the object itself: an id() can be reused after GC
    if cached is not None:
        return cached
    ok = False
    try:
        if getattr(tokenizer, "is_fast", False):
            import json

            cfg = json.loads(tokenizer.backend_tokenizer.to_str())
            pre = cfg.get("pre_tokenizer") or {}
            parts = pre.get("pretokenizers", [pre]) if pre.get("type") == "Sequence" else [pre]
            kinds = [p_.get("type") for p_ in parts]
            ok = (cfg.get("normalizer") is None and (cfg.get("model") or {}).get("type") == "BPE"
                  and bool(kinds) and set(kinds) <= {"Split", "ByteLevel"} and "ByteLevel" in kinds
                  and all(not p_.get("add_prefix_space", False) for p_ in parts if p_.get("type") == "ByteLevel")
                  and all(p_.get("behavior", "Isolated") == "Isolated" and not p_.get("invert", False)
                          for p_ in parts if p_.get("type") == "Split"))
    except Exception:
        ok = False
    try:
        tokenizer._sdag_is_local = ok
    except Exception:  # an object that refuses attributes: decide again next time
        pass
    return ok


def _prefix_token_counts(tokenizer, text: str, cuts: List[int]) -> List[int]:
    """``len(tokenizer(text[:c]).input_ids)`` for every cut c -- what the reference computes with one full
    tokenisation per cut (SDAG.py:289-302) -- from ONE tokenisation of ``text`` plus a few characters around each
    cut.  Pre-tokens that end well before a cut are the same in the prefix and in the whole text (the split
    pattern scans left to right and looks at most three characters ahead), and byte-level BPE never merges across
    pre-tokens; the stretch from the last such pre-token to the cut is tokenised on its own.  Tokenizers that are
    not of that kind take the reference's route."""
    if not cuts:
        return []
    if not _tokenizer_is_local(tokenizer):
        return [_n_tokens(tokenizer, text[:c]) for c in cuts]
    import bisect

    enc = tokenizer(text, return_offsets_mapping=True).encodings[0]
    offs, words, special = enc.offsets, enc.word_ids, enc.special_tokens_mask
    if any(w is None and not m for w, m in zip(words, special)):  # no pre-token ids: no safe place to cut
        return [_n_tokens(tokenizer, text[:c]) for c in cuts]
    n_added = sum(1 for m in special if m)  # what the post-processor adds to any text (BOS ...)
    ends, upto = [], []                      # char end of every complete pre-token, tokens up to and including it
    n_content = 0
    for i, (m, (_, e)) in enumerate(zip(special, offs)):
        if m:
            continue
        n_content += 1
        nxt = i + 1
        while nxt < len(offs) and special[nxt]:
            nxt += 1
        if nxt == len(offs) or words[nxt] is None or words[nxt] != words[i]:
            ends.append(e)
            upto.append(n_content)
    out = []
    for c in cuts:
        j = bisect.bisect_right(ends, c - _TAIL_MARGIN) - 1
        a, n0 = (ends[j], upto[j]) if j >= 0 else (0, 0)
        tail = text[a:c]
        n1 = len(tokenizer(tail, add_special_tokens=False).input_ids) if tail else 0
        out.append(n_added + n0 + n1)
    return out


def build_rag_chat_and_spans(tokenizer, system_prompt: str, user_template: str, query: str, malicious_docs: List[str],
                             retrieved_docs: list, add_attack_in_rank: int):
    """Prompt string and the token spans of its parts (SDAG.py:216-304).

    Returns (chat_str, sys_user_len, doc_token_spans, qa_start, ranked_docs).  Token positions are those of
    prefix tokenisations of ``chat_str`` -- the length in tokens of the text before each boundary -- which is how
    the reference defines them (SDAG.py:289-302); they are obtained from one tokenisation of the prompt where the
    tokenizer allows it (``_prefix_token_counts``) instead of the reference's 2K + 2.
    """
    ranked_docs = inject_malicious_docs_into_ranked_list(base_docs=retrieved_docs, malicious_docs=malicious_docs,
                                                         attack_pos=add_attack_in_rank)
    bullets = [f"- {d.strip()}" for d in ranked_docs if d and d.strip()]
    user_content = user_template.format(query=query, docs_text="\n\n".join(bullets))
    chat_str = tokenizer.apply_chat_template(
        [{"role": "system", "content": system_prompt}, {"role": "user", "content": user_content}],
        tokenize=False, add_generation_prompt=True)

    starts, cursor = [], 0
    for text in bullets:  # first occurrence at or after the previous bullet; a miss stays at the cursor (SDAG.py:277-282)
        at = chat_str.find(text, cursor)
        at = cursor if at == -1 else at
        starts.append(at)
        cursor = at + len(text)
    q_at = chat_str.find("- Question:")
    q_at = len(chat_str) if q_at == -1 else q_at

    cuts = [starts[0] if starts else q_at]
    for text, at in zip(bullets, starts):
        cuts += [at, at + len(text)]
    cuts.append(q_at)
    counts = _prefix_token_counts(tokenizer, chat_str, cuts)
    sys_user_len, qa_start = counts[0], counts[-1]
    spans = [(counts[1 + 2 * i], counts[2 + 2 * i]) for i in range(len(bullets))]
    return chat_str, sys_user_len, spans, qa_start, ranked_docs


def run_rag_with_doc_isolation(model, tokenizer, query: str, malicious_docs: List[str], retrieved_docs: list,
                               max_new_tokens: int, device: str, add_attack_in_rank: int, ranker, doc_neighbors_k: int):
    """Answer ``query`` over the ranked documents with document-isolated attention (SDAG.py:307-375)."""
    chat_str, sys_user_len, spans, qa_start, ranked_docs = build_rag_chat_and_spans(
        tokenizer=tokenizer, system_prompt=SYSTEM_PROMPT_RAG, user_template=USER_RAG_PROMPT, query=query,
        malicious_docs=malicious_docs, retrieved_docs=retrieved_docs, add_attack_in_rank=add_attack_in_rank)
    neighbors = None
    if doc_neighbors_k and doc_neighbors_k > 0:
        neighbors = compute_doc_knn_for_docs(ranker=ranker, docs=ranked_docs, k_neighbors=doc_neighbors_k)
    input_ids = tokenizer(chat_str, return_tensors="pt").to(device)["input_ids"]
    desc = sdag_amd.from_spans(input_ids.size(1), sys_user_len, spans, qa_start, neighbors)
    return _generate_with_descriptor(model, tokenizer, input_ids, desc, max_new_tokens=max_new_tokens)


def _cache_of_sequence(past, b: int, length: int):
    """The key/value cache of sequence ``b`` of a batched prefill, cut to its own ``length`` positions, as a cache
    of its own (what a prefill of that prompt alone would have returned)."""
    from transformers import DynamicCache

    own = DynamicCache()
    for layer_idx, layer in enumerate(past.layers):
        own.update(layer.keys[b:b + 1, :, :length].contiguous(), layer.values[b:b + 1, :, :length].contiguous(), layer_idx)
    return own


def _decode_loop_batched(model, tokenizer, ids: List[torch.Tensor], out, lengths: List[int], max_new_tokens: int) -> List[str]:
    """The reference's decode loop (SDAG.py:178-213) for all prompts of a right-padded prefill at once: one model
    call per step for the whole batch.  Per prompt the rules are the reference's: up to max_new_tokens + 1 tokens, the
    end-of-sequence token ends that prompt's answer (later steps of a finished prompt are computed and discarded);
    the loop stops when every prompt has finished.  New tokens sit behind the padding in the cache (rows l_max + t of
    every sequence) and carry their own positions lengths[b] + t; the padding rows are left out by the decode kernel.
    """
    n, l_max = len(ids), max(lengths)
    dev = out.logits.device
    valid = torch.tensor(lengths, dtype=torch.int32, device=dev)
    last = torch.tensor([l - 1 for l in lengths], device=dev)
    tok = _sample(out.logits[torch.arange(n, device=dev), last])                      # [n, 1]
    new_tokens = [tok.clone()]
    finished = torch.zeros(n, dtype=torch.bool, device=dev)
    pos = valid.to(torch.long).unsqueeze(1)                                             # position of the token fed next

    def after_step():
        # the reference tests for EOS after appending the token (SDAG.py:208-211); a finished prompt stays finished
        nonlocal finished
        new_tokens.append(tok.clone())
        finished = finished | (tok[:, 0] == tokenizer.eos_token_id)
        return bool(finished.all())

    if decode_with_graph() and dev.type == "cuda":
        # the same steps over a preallocated cache, captured once and replayed (see _decode_loop_graphed)
        cache = _static_cache_from(model, out.past_key_values, l_max, l_max + max_new_tokens + 1)
        n_rows = torch.full((1,), l_max, dtype=torch.int32, device=dev)
        tok, pos = tok.clone(), pos.clone()

        def one_step():
            n_rows.add_(1)
            step = model(input_ids=tok, past_key_values=cache, position_ids=pos, use_cache=True)
            tok.copy_(_sample(step.logits[:, -1, :]))
            pos.add_(1)

        with hf_attention.active_padding(valid, l_max, n_rows=n_rows):
            _replay_steps(dev, one_step, max_new_tokens, after_step)
    else:
        past = out.past_key_values
        with hf_attention.active_padding(valid, l_max):
            for _ in range(max_new_tokens):
                with torch.no_grad():
                    step = model(input_ids=tok, past_key_values=past, position_ids=pos, use_cache=True)
                past = step.past_key_values
                tok = _sample(step.logits[:, -1, :])
                pos = pos + 1
                if after_step():
                    break
    gen = torch.cat(new_tokens, dim=1).cpu()
    answers = []
    for b in range(n):
        row = gen[b]
        # keep the tokens up to and including the first EOS among those produced inside the loop (the first token,
        # sampled from the prefill, is never tested, as in the reference)
        hits = (row[1:] == tokenizer.eos_token_id).nonzero()
        end = int(hits[0]) + 2 if len(hits) else len(row)
        answers.append(tokenizer.decode(row[:end], skip_special_tokens=True).strip())
    return answers


def run_rag_with_doc_isolation_batch(model, tokenizer, queries: List[str], malicious_docs_batch: List[List[str]],
                                     retrieved_docs_batch: List[list], max_new_tokens: int, device: str,
                                     add_attack_in_rank, ranker, doc_neighbors_k: int,
                                     decode: str = "per_prompt") -> List[str]:
    """``run_rag_with_doc_isolation`` for several queries at once (no counterpart in the reference, which answers one
    query per call: src/pipeline/main.py:469-496).  The masked prefill -- where the time goes -- runs ONCE for all
    prompts: they are right-padded to a common length, every prompt keeps its own descriptor, and each attention
    layer is one batched kernel launch.  The answers are then decoded prompt by prompt from that prompt's slice of
    the cache with the reference's loop, in order, so sampling draws from the global RNG in the same sequence as one
    call per query would (the prefill draws nothing).  ``decode="batched"`` decodes all prompts together instead (one
    model call per step for the whole batch: the weights stream once per step rather than once per prompt and step);
    greedy answers are the same, sampled ones come from the same distributions but draw from the RNG in another order.
    Returns one answer per query.
    """
    if decode not in ("per_prompt", "batched"):
        raise ValueError("decode must be 'per_prompt' or 'batched'")
    n = len(queries)
    if not (n == len(malicious_docs_batch) == len(retrieved_docs_batch)):
        raise ValueError("one list of malicious documents and one of retrieved documents per query expected")
    if n == 0:
        return []
    ids, layouts = [], []
    for query, mal, docs in zip(queries, malicious_docs_batch, retrieved_docs_batch):
        chat_str, sys_user_len, spans, qa_start, ranked_docs = build_rag_chat_and_spans(
            tokenizer=tokenizer, system_prompt=SYSTEM_PROMPT_RAG, user_template=USER_RAG_PROMPT, query=query,
            malicious_docs=mal, retrieved_docs=docs, add_attack_in_rank=add_attack_in_rank)
        neighbors = None
        if doc_neighbors_k and doc_neighbors_k > 0:
            neighbors = compute_doc_knn_for_docs(ranker=ranker, docs=ranked_docs, k_neighbors=doc_neighbors_k)
        input_ids = tokenizer(chat_str, return_tensors="pt").to(device)["input_ids"]
        ids.append(input_ids)
        layouts.append((input_ids.size(1), sys_user_len, spans, qa_start, neighbors))
    descs = sdag_amd.from_spans_many(layouts, max_items_per_unit=64)  # (host threads: the builder is a C call)
    lengths = [int(t.size(1)) for t in ids]
    l_max = max(lengths)
    pad_id = getattr(tokenizer, "pad_token_id", None)  # any valid id: rows past a prompt's end are never read
    if pad_id is None or pad_id < 0:
        pad_id = max(0, tokenizer.eos_token_id or 0)
    batch_ids = torch.full((n, l_max), int(pad_id), dtype=ids[0].dtype, device=ids[0].device)
    for b, t in enumerate(ids):
        batch_ids[b, :lengths[b]] = t[0]
    batch = sdag_amd.BatchDescriptor(descs)
    answers: List[str] = []
    with hf_attention.use_on_model(model):
        with torch.no_grad(), hf_attention.active_descriptor(batch):
            out = model(input_ids=batch_ids, use_cache=True)  # rows past a prompt's end attend nothing and are never read
        if decode == "batched":
            return _decode_loop_batched(model, tokenizer, ids, out, lengths, max_new_tokens)
        for b in range(n):
            first = type(out)(logits=out.logits[b:b + 1, :lengths[b]], past_key_values=_cache_of_sequence(out.past_key_values, b, lengths[b]))
            answers.append(_decode_loop(model, tokenizer, ids[b], first, max_new_tokens))
    return answers
