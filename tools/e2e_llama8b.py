#!/usr/bin/env python3
"""End-to-end timing on a randomly initialised Llama-3-8B-shaped model (bf16, 32 layers, 32 / 8 heads, head_dim 128;
no weights are fetched): the masked prefill and the decode steps of one prompt of the benchmark's shape
(10 documents x 300 tokens, L = 3100), (a) the way the reference runs them on a GPU -- a dense additive [1,1,L,L] mask
handed to transformers' SDPA attention, stock decode -- and (b) through the descriptor-driven kernels.  Also the
batched prefill of 8 prompts.  Prints one JSON object (diagnostic, GPU only)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparse-document-attention-rag-_amd"))
sys.path.insert(0, ROOT)
import torch
from transformers import LlamaConfig, LlamaForCausalLM

import sdag_amd
from sdag_amd import hf_attention

dev = torch.device("cuda:0")
n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=n_layers, num_attention_heads=32,
                  num_key_value_heads=8, head_dim=128, max_position_embeddings=8192, rope_theta=500000.0)
cfg._attn_implementation = "sdpa"
torch.manual_seed(0)
old = torch.get_default_dtype()
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = LlamaForCausalLM(cfg).eval()
torch.set_default_dtype(old)

sys_len, n_docs, doc_len, gap, qa_len = 60, 10, 300, 1, 30
spans, pos = [], sys_len
for _ in range(n_docs):
    spans.append((pos, pos + doc_len)); pos += doc_len + gap
qa, L = pos, pos + qa_len
ids = torch.randint(5, 100000, (1, L), device=dev)
desc = sdag_amd.from_spans(L, sys_len, spans, qa, None)
n_dec = 16


def sync_time(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return best, r


def decode_steps(out, input_ids):
    past = out.past_key_values
    tok = out.logits[:, -1:].argmax(-1)
    for _ in range(n_dec):
        o = model(input_ids=tok, past_key_values=past, use_cache=True)
        past = o.past_key_values
        tok = o.logits[:, -1:].argmax(-1)
    return tok


res = {"model": f"Llama-3-8B shape, {n_layers} layers, random bf16 weights", "L": L}
with torch.no_grad():
    # (a) dense additive mask + SDPA: the reference's path on this GPU (mask expanded on the GPU, outside the clock)
    add = torch.zeros((L, L), dtype=torch.bfloat16, device=dev).masked_fill(~desc.materialize(dev), torch.finfo(torch.bfloat16).min)[None, None]
    t, out_ref = sync_time(lambda: model(input_ids=ids, attention_mask=add, use_cache=True))
    res["prefill_ms_dense_mask_sdpa"] = t * 1e3
    ref_logits = out_ref.logits[0, -1].float()
    t0 = time.perf_counter(); decode_steps(out_ref, ids); torch.cuda.synchronize()
    out_ref = model(input_ids=ids, attention_mask=add, use_cache=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); decode_steps(out_ref, ids); torch.cuda.synchronize()
    res["decode_ms_per_token_sdpa"] = (time.perf_counter() - t0) / n_dec * 1e3
    del out_ref
    # (b) descriptor-driven kernels through the transformers hook
    with hf_attention.use_on_model(model):
        def ours():
            with hf_attention.active_descriptor(desc):
                return model(input_ids=ids, use_cache=True)
        t, out = sync_time(ours)
        res["prefill_ms_sdag_kernel"] = t * 1e3
        lg = out.logits[0, -1].float()
        ours_logits = lg
        res["last_logits_max_abs_diff_over_max_abs"] = float((lg - ref_logits).abs().max() / ref_logits.abs().max())
        decode_steps(out, ids); torch.cuda.synchronize()
        out = ours(); torch.cuda.synchronize()
        t0 = time.perf_counter(); decode_steps(out, ids); torch.cuda.synchronize()
        res["decode_ms_per_token_sdag_kernel"] = (time.perf_counter() - t0) / n_dec * 1e3
        # the same steps replayed from one captured HIP graph over a preallocated cache (the drop-in's
        # SDAG_DECODE_GRAPH=1 loop): per-token time from the difference of a long and a short run
        SDAG = __import__("importlib").import_module("src.pipeline.sparse_attention_RAG.SDAG")
        SDAG.TEMPERATURE = 0

        class _Tok:
            eos_token_id = -1

            def decode(self, ids_, skip_special_tokens=True):
                return ""

        def graphed(n):
            torch.cuda.synchronize(); t_ = time.perf_counter()
            SDAG._decode_loop_graphed(model, _Tok(), ids, out, n); torch.cuda.synchronize()
            return time.perf_counter() - t_
        out = ours(); torch.cuda.synchronize()  # a fresh cache: the eager steps above appended to the other one
        graphed(4)
        t_short, t_long = graphed(8), graphed(8 + 64)
        res["decode_ms_per_token_sdag_kernel_graph_replay"] = (t_long - t_short) / 64 * 1e3
        res["decode_graph_setup_ms"] = (t_short - 8 * (t_long - t_short) / 64) * 1e3
        del out
        # batched prefill: 8 prompts, one launch per layer
        B = 8
        bids = torch.randint(5, 100000, (B, L), device=dev)
        batch = sdag_amd.BatchDescriptor([sdag_amd.from_spans(L, sys_len, spans, qa, None, max_items_per_unit=64) for _ in range(B)])
        def ours_b():
            with hf_attention.active_descriptor(batch):
                return model(input_ids=bids, use_cache=True).logits[:, -1]
        t, _ = sync_time(ours_b, reps=2)
        res["prefill_ms_per_prompt_sdag_kernel_batch8"] = t * 1e3 / B
        # the 8 prompts decoded together: one model call per step for the batch (padded decode kernel)
        with hf_attention.active_descriptor(batch):
            outb = model(input_ids=bids, use_cache=True)
        valid = torch.full((B,), L, dtype=torch.int32, device=dev)
        def decode_batch(o):
            past, tok, pos = o.past_key_values, o.logits[:, -1:].argmax(-1), valid.long().unsqueeze(1)
            with hf_attention.active_padding(valid, L):
                for _ in range(n_dec):
                    st = model(input_ids=tok, past_key_values=past, position_ids=pos, use_cache=True)
                    past, tok, pos = st.past_key_values, st.logits[:, -1:].argmax(-1), pos + 1
            return tok
        decode_batch(outb); torch.cuda.synchronize()
        with hf_attention.active_descriptor(batch):
            outb = model(input_ids=bids, use_cache=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); decode_batch(outb); torch.cuda.synchronize()
        res["decode_ms_per_step_batch8_sdag_kernel"] = (time.perf_counter() - t0) / n_dec * 1e3
        res["decode_tokens_per_s_batch8_sdag_kernel"] = B * n_dec / (time.perf_counter() - t0)
        # ... and the same batched steps replayed from one captured HIP graph (SDAG_DECODE_GRAPH=1)
        with hf_attention.active_descriptor(batch):
            outb = model(input_ids=bids, use_cache=True)
        os.environ["SDAG_DECODE_GRAPH"] = "1"

        def graphed_batch(n):
            torch.cuda.synchronize(); t_ = time.perf_counter()
            SDAG._decode_loop_batched(model, _Tok(), [None] * B, outb, [L] * B, n); torch.cuda.synchronize()
            return time.perf_counter() - t_
        graphed_batch(4)
        t_short, t_long = graphed_batch(8), graphed_batch(8 + 64)
        os.environ["SDAG_DECODE_GRAPH"] = "0"
        res["decode_ms_per_step_batch8_sdag_kernel_graph_replay"] = (t_long - t_short) / 64 * 1e3
        res["decode_tokens_per_s_batch8_sdag_kernel_graph_replay"] = B * 64 / (t_long - t_short)
        del outb
    addb = add.expand(B, 1, L, L)
    t, _ = sync_time(lambda: model(input_ids=bids, attention_mask=addb, use_cache=True).logits[:, -1], reps=2)
    res["prefill_ms_per_prompt_dense_mask_sdpa_batch8"] = t * 1e3 / B
    # both bf16 paths against the same model in fp32 (dense additive mask, SDPA): whose last-position logits are closer?
    model.float()
    add32 = torch.zeros((L, L), dtype=torch.float32, device=dev).masked_fill(~desc.materialize(dev), torch.finfo(torch.float32).min)[None, None]
    l32 = model(input_ids=ids, attention_mask=add32).logits[0, -1]
    den = float(l32.abs().max())
    res["err_vs_fp32_model_dense_mask_sdpa_bf16"] = float((ref_logits - l32).abs().max()) / den
    res["err_vs_fp32_model_sdag_kernel_bf16"] = float((ours_logits - l32).abs().max()) / den
print(json.dumps(res), flush=True)
