/*
 * C ABI of libsdag_hip.so -- the MI355X (gfx950) implementation of the SDAG document-attention
 * prefill and the dense-retrieval inner-product search.
 *
 * The reference is pure Python with no native layer: its arithmetic runs inside torch / transformers /
 * faiss.  Each entry point below therefore cites the reference CALL SITE whose work it replaces (paths
 * relative to the reference tree).  All functions return 0 on success or a negative SDAG_ERR_* code;
 * sdag_last_error() returns a thread-local message for the last failure.  No entry point allocates or
 * frees caller memory.  Every function that launches kernels is asynchronous on the stream it is given
 * (a hipStream_t passed as void*; NULL = the default stream) and re-entrant per (device, stream).
 * Device pointers must belong to the current HIP device.
 */
#ifndef SDAG_HIP_H
#define SDAG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDAG_OK 0
#define SDAG_ERR_ARG (-1)      /* invalid argument (shape, alignment, range) */
#define SDAG_ERR_CAPACITY (-2) /* caller buffer too small; the needed size is reported */
#define SDAG_ERR_HIP (-3)      /* a HIP runtime call failed */
#define SDAG_ERR_DESC (-4)     /* malformed descriptor */

#define SDAG_OUT_BF16 0
#define SDAG_OUT_F32 1

#define SDAG_ROW_BLOCK 32 /* query rows per descriptor row block */
#define SDAG_KV_TILE 64   /* keys per descriptor tile */

int sdag_version(void);
const char* sdag_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Block-sparse descriptor (host side, no GPU work).
 *
 * Replaces the L x L boolean mask that build_blocked_causal_mask_full() fills with Python row loops
 * (src/pipeline/sparse_attention_RAG/SDAG.py:104-127) by a list of (row block, key tile) work items
 * plus a 32 x 64 bit mask for every partially allowed tile.  Semantics are exactly those of the
 * reference loops (union of: causal rows below sys_user_len and from qa_start on; per document the
 * prefix [0, sys_user_len), the causal part of the document and the whole span of each valid
 * neighbour; rows in no region allow nothing).  Rows that allow nothing are listed as "uniform rows":
 * after the reference's finfo.min additive mask (SDAG.py:164-173) their softmax is uniform over all L
 * keys, and the attention kernels reproduce that.
 *
 *   spans      2 * n_docs ints (start, end), 0 <= start, end <= L (start >= end: empty document)
 *   nbr_off    n_docs + 1 offsets into nbr_idx, or NULL for "no neighbours" (the reference's
 *              doc_neighbors=None, and also its len(doc_neighbors) != num_docs case, SDAG.py:111)
 *   nbr_idx    neighbour document ids; ids outside [0, n_docs) are skipped (SDAG.py:119-120)
 *   max_items_per_unit  longest run of key tiles walked for one row block before it is split into
 *              several units that are merged afterwards; 0 picks a default from the layout
 *   n_streams  the units are dealt to this many work streams; the attention launch starts one persistent
 *              workgroup per stream and group of (up to 4) query heads sharing a key/value head.  0 = 64,
 *              which fills 256 compute units at 2 workgroups each when Hkv * ceil(Hq / Hkv / 4) = 8
 *   out        caller buffer of cap_words int32 words (may be NULL to query the size)
 *   n_words    receives the number of words the descriptor needs
 * Returns SDAG_ERR_CAPACITY when cap_words is too small (n_words is still set).
 */
int sdag_desc_build(int L, int sys_user_len, int n_docs, const int32_t* spans, int qa_start,
                    const int32_t* nbr_off, const int32_t* nbr_idx, int max_items_per_unit,
                    int n_streams, int32_t* out, int64_t cap_words, int64_t* n_words);

/* Same descriptor from an arbitrary dense mask (uint8 [L, L] on the host, nonzero = may attend), for
 * callers of generate_with_custom_mask() that bring their own prompt_mask (SDAG.py:130-136,167-171). */
int sdag_desc_from_dense(const uint8_t* mask, int L, int max_items_per_unit, int n_streams, int32_t* out,
                         int64_t cap_words, int64_t* n_words);

/* Number of (row, column) pairs the descriptor allows, not counting uniform rows; n_uniform_rows may
 * be NULL.  Used for FLOP accounting. */
int sdag_desc_stats(const int32_t* desc_host, int64_t* allowed_pairs, int32_t* n_uniform_rows,
                    int32_t* n_units, int32_t* n_items, int32_t* n_partial_tiles);

/* Expand a descriptor (device copy) into a dense uint8 [L, L] mask on the device: 1 = may attend.
 * This is what build_blocked_causal_mask_full() returns (SDAG.py:105,127) and is how the descriptor
 * is proven bit-identical to the reference mask. */
int sdag_mask_materialize(const int32_t* desc_host, const int32_t* desc_dev, uint8_t* out_dev,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * Masked attention prefill: O = softmax(Q K^T * scale + M) V with grouped-query heads, M given by the
 * descriptor.  Replaces the attention the reference reaches through
 *   model(input_ids=input_ids, attention_mask=attn[1,1,L,L], use_cache=True)   (SDAG.py:176)
 * i.e. transformers' LlamaAttention -> attention_interface(query[B,Hq,L,D], key[B,Hkv,L,D],
 * value[B,Hkv,L,D], attention_mask, scaling) -> output [B,L,Hq,D].
 *
 * q, k, v: bf16, head_dim D = 128 contiguous; element strides for batch / head / row are passed so
 * the transposed views transformers produces need no copy (all strides multiples of 8 elements, base
 * pointers 16-byte aligned).  o: [B, L, Hq, D] with element strides o_sb / o_sl / o_sh, bf16 or fp32
 * (out_dtype).  The same descriptor applies to every batch element.  Hq must be a multiple of Hkv.
 * Limits (SDAG_ERR_ARG beyond them): every stride below 2^31 elements; the L rows of one (batch, head)
 * slice of q, of k and of v span less than 4 GiB (L * k_sl * 2 bytes), those of o less than 2 GiB.
 * workspace: device scratch of at least sdag_attn_workspace_bytes() bytes (may be NULL when that is 0).
 */
int64_t sdag_attn_workspace_bytes(const int32_t* desc_host, int B, int Hq, int Hkv);

int sdag_attn_fwd(const void* q, const void* k, const void* v, void* o,
                  int64_t q_sb, int64_t q_sh, int64_t q_sl,
                  int64_t k_sb, int64_t k_sh, int64_t k_sl,
                  int64_t v_sb, int64_t v_sh, int64_t v_sl,
                  int64_t o_sb, int64_t o_sl, int64_t o_sh,
                  int B, int Hq, int Hkv, int L, int D, float scale,
                  const int32_t* desc_host, const int32_t* desc_dev,
                  void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* sdag_attn_fwd for callers that cannot allocate device memory or carry a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference (which never manages
 * device memory itself).  Same arguments and results as sdag_attn_fwd without the workspace: the scratch (partial
 * results of split row blocks, column sums of V) lives in a device array that is part of the library's image,
 * SDAG_STATIC_SCRATCH_BYTES per GPU, held until the process ends.  Returns SDAG_ERR_CAPACITY when
 * sdag_attn_workspace_bytes(desc_host, B, Hq, Hkv) exceeds that size.  The call allocates and frees nothing.  All
 * calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different streams or threads)
 * race; calls on one stream, or otherwise ordered, are safe.  Nothing in the array has to be cleared between calls:
 * every word a call reads was written earlier by that same call.  The FIRST call on a device looks up the array's
 * address there (hipGetSymbolAddress, which may load the library's code object: not a capturable operation), so make
 * one eager call per device before calling under stream capture; every later call only launches kernels. */
#define SDAG_STATIC_SCRATCH_BYTES (8 << 20)

int sdag_attn_fwd_static(const void* q, const void* k, const void* v, void* o,
                         int64_t q_sb, int64_t q_sh, int64_t q_sl,
                         int64_t k_sb, int64_t k_sh, int64_t k_sl,
                         int64_t v_sb, int64_t v_sh, int64_t v_sl,
                         int64_t o_sb, int64_t o_sl, int64_t o_sh,
                         int B, int Hq, int Hkv, int L, int D, float scale,
                         const int32_t* desc_host, const int32_t* desc_dev, int out_dtype, void* stream);

/* Several sequences in one launch, each with its OWN descriptor (different prompts have different spans and
 * lengths).  The reference runs one query at a time (src/pipeline/main.py:469-496 calls
 * run_rag_with_doc_isolation per query); batching the masked prefill of many queries is how this build fills
 * the GPU, so these entry points have no single call site to cite beyond SDAG.py:176.
 *
 * sdag_batch_build concatenates B host descriptors (the same pointer may repeat) and deals their units, in
 * sequence-major order, to n_streams work streams (0 = 64; with eight or more units per stream, the units of a
 * sequence that end at its low keys go to the first half of the streams or to the second, the others to the other
 * half, so that the two halves share no K/V rows, and the streams of a half are then levelled to within one
 * short unit of each other).  Tensors are padded to Lmax rows per sequence:
 * q [B, Hq, Lmax, D], k / v [B, Hkv, Lmax, D], o [B, Lmax, Hq, D] through the usual strides; rows at and
 * beyond a sequence's own length are neither read as queries nor written. */
int sdag_batch_build(const int32_t* const* descs_host, int B, int n_streams, int32_t* out, int64_t cap_words,
                     int64_t* n_words);

int64_t sdag_attn_batch_workspace_bytes(const int32_t* batch_host, int Hq, int Hkv);

int sdag_attn_fwd_batch(const void* q, const void* k, const void* v, void* o,
                        int64_t q_sb, int64_t q_sh, int64_t q_sl,
                        int64_t k_sb, int64_t k_sh, int64_t k_sl,
                        int64_t v_sb, int64_t v_sh, int64_t v_sl,
                        int64_t o_sb, int64_t o_sl, int64_t o_sh,
                        int B, int Hq, int Hkv, int Lmax, int D, float scale,
                        const int32_t* batch_host, const int32_t* batch_dev,
                        void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Single-token attention over the key/value cache: O = softmax(q K^T * scale) V for ONE query position per
 * sequence, every cached key allowed, grouped-query heads.  Replaces the attention of the decode steps
 *   model(input_ids=generated[:, -1:], past_key_values=past_key_values, use_cache=True)   (SDAG.py:191-199)
 * to which the reference passes no mask.
 *
 * q [B, Hq, 1, D] (element strides q_sb, q_sh), k / v [B, Hkv, S, D] (strides _sb, _sh, _sl), o [B, 1, Hq, D]
 * (strides o_sb, o_sh), bf16, D = 128 contiguous, strides multiples of 8 elements, pointers 16-byte aligned;
 * o bf16 or fp32.  workspace: at least sdag_decode_workspace_bytes(B, Hq, S) bytes of device scratch.
 */
int64_t sdag_decode_workspace_bytes(int B, int Hq, int S);

int sdag_decode_attn(const void* q, const void* k, const void* v, void* o,
                     int64_t q_sb, int64_t q_sh,
                     int64_t k_sb, int64_t k_sh, int64_t k_sl,
                     int64_t v_sb, int64_t v_sh, int64_t v_sl,
                     int64_t o_sb, int64_t o_sh,
                     int B, int Hq, int Hkv, int S, int D, float scale,
                     void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* The same step for several right-padded prompts decoded together (no call site in the reference, which decodes one
 * query at a time, SDAG.py:191-211; used by the batched form of run_rag_with_doc_isolation): the cache of sequence b
 * holds its prompt in rows [0, valid_len[b]), padding in rows [valid_len[b], pad_end) and the tokens generated so far
 * in rows [pad_end, S); the padding rows take no part.  valid_len: B int32 on the device, 0 <= valid_len[b] <= pad_end
 * <= S; a sequence must be left with at least one key. */
int sdag_decode_attn_padded(const void* q, const void* k, const void* v, void* o,
                            int64_t q_sb, int64_t q_sh,
                            int64_t k_sb, int64_t k_sh, int64_t k_sl,
                            int64_t v_sb, int64_t v_sh, int64_t v_sl,
                            int64_t o_sb, int64_t o_sh,
                            int B, int Hq, int Hkv, int S, int D, float scale,
                            const int32_t* valid_len, int pad_end,
                            void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* The same step over a PREALLOCATED cache of S rows of which only the first *n_rows hold keys (no call site in the
 * reference, whose cache grows by concatenation, SDAG.py:191-199; for decode steps captured once as a HIP graph and
 * replayed: the launch arguments stay the same from step to step, the number of cached rows is read on the device).
 * n_rows: one int32 on the device, 1 <= *n_rows <= S at the time the kernel runs (the same for every sequence).
 * valid_len / pad_end as for sdag_decode_attn_padded, or valid_len NULL for no padding rows. */
int sdag_decode_attn_prealloc(const void* q, const void* k, const void* v, void* o,
                              int64_t q_sb, int64_t q_sh,
                              int64_t k_sb, int64_t k_sh, int64_t k_sl,
                              int64_t v_sb, int64_t v_sh, int64_t v_sl,
                              int64_t o_sb, int64_t o_sh,
                              int B, int Hq, int Hkv, int S, int D, float scale,
                              const int32_t* n_rows, const int32_t* valid_len, int pad_end,
                              void* workspace, int64_t workspace_bytes, int out_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Exact inner-product search.  Replaces  scores, indices = index.search(q, top_k)
 * (src/pipeline/retrieval/dense.py:82-84; the arithmetic of the reference is inside faiss).
 *
 *   score(q, c) = fp32 FMA chain over the dimensions in order, starting from +0
 *   order       = (score descending, id ascending); NaN scores are never returned
 *   fill        = (score -FLT_MAX, id -1) when fewer than k rows exist
 *
 * Q [nq, d] fp32 row-major (device), C [N, d] fp32 row-major (device), d a multiple of 32, 1 <= k <= 32.
 * out_s [nq, k] fp32 and out_i [nq, k] int64 on the device; ids are id_base + row.
 * workspace: at least sdag_dense_workspace_bytes(nq, N, d, k) bytes of device scratch.
 */
int64_t sdag_dense_workspace_bytes(int nq, int64_t N, int d, int k);

int sdag_dense_topk(const float* Q, int nq, const float* C, int64_t N, int d, int k,
                    float* out_s, int64_t* out_i, int64_t id_base,
                    void* workspace, int64_t workspace_bytes, void* stream);

/* The next page of the same search: the best k rows among those that come strictly AFTER the entry
 * (after_s[q], after_i[q]) of query q in the order of the results (after_s [nq] fp32, after_i [nq] int64 ids as
 * returned, i.e. including id_base; both on the device).  With the last entry of the previous page as the cursor,
 * successive calls return ranks k+1 .. 2k, 2k+1 .. 3k, ...: the reference's index.search(q, top_k)
 * (src/pipeline/retrieval/dense.py:82-84) takes any top_k, one call here returns at most 32 entries per query.
 * The cursor may name a row of another shard (the order is over global ids).  A cursor that is a fill entry
 * (-FLT_MAX, -1) leaves nothing to return.  Same arguments and workspace as sdag_dense_topk otherwise. */
int sdag_dense_topk_after(const float* Q, int nq, const float* C, int64_t N, int d, int k,
                          const float* after_s, const int64_t* after_i,
                          float* out_s, int64_t* out_i, int64_t id_base,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* The same search, same results bit for bit, in two stages that read half the bytes: a scan over a bf16 copy of
 * the corpus picks candidates (every row whose approximate score is within a proven error bound of the k-th best),
 * which are then scored with the exact fp32 chain and ranked.  When the candidate set cannot be vouched for (more
 * candidates than the kernels hold), the exact scan of sdag_dense_topk runs instead, decided on the device.
 * Replaces the same reference call as sdag_dense_topk (dense.py:82-84).
 *
 * sdag_corpus_to_bf16: C16 [N, d] (bf16, round to nearest even) and *max_row_norm (device float: an upper bound of
 * the largest row norm of C) from C [N, d]; d a multiple of 4.  Called once per corpus.
 * sdag_dense_topk2: d a multiple of 64, 1 <= k <= 32; values of Q and C finite. */
int sdag_corpus_to_bf16(const float* C, int64_t N, int d, void* C16, float* max_row_norm, void* stream);

int64_t sdag_dense_topk2_workspace_bytes(int nq, int64_t N, int d, int k);

/* Diagnostics: byte offset inside the workspace of an int that is non-zero after a call whose last batch of up to 32
 * queries was answered by the exact scan instead of the rescoring kernel. */
int64_t sdag_dense_topk2_flag_offset(int nq, int64_t N, int d, int k);

int sdag_dense_topk2(const float* Q, int nq, const float* C, const void* C16, const float* max_row_norm, int64_t N,
                     int d, int k, float* out_s, int64_t* out_i, int64_t id_base,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* Merge n_parts partial result lists (each [nq, k], as written by sdag_dense_topk on one corpus
 * shard, e.g. after an all-gather) into the global best k per query, same ordering rule.
 * in_s [n_parts, nq, k], in_i [n_parts, nq, k] on the device; entries with id < 0 are fill. */
int sdag_topk_merge(const float* in_s, const int64_t* in_i, int n_parts, int nq, int k,
                    float* out_s, int64_t* out_i, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small similarity ops on unit-norm fp32 embeddings.
 *
 * sdag_gram: sims = E E^T for E [n, d] (device) -> sims [n, n]; replaces np.matmul(emb, emb.T) in
 * compute_doc_knn_for_docs (SDAG.py:50).  sdag_centroid_sims: centroid = mean of R [nr, d] rows,
 * sims[j] = <Cand[j], centroid>; replaces malicious_selection.py:89-95,117-123.  Both use the same
 * dimension-ordered fp32 FMA chain as the search. */
int sdag_gram(const float* E, int n, int d, float* sims, void* stream);
int sdag_centroid_sims(const float* R, int nr, const float* Cand, int nc, int d, float* centroid,
                       float* sims, void* stream);

/* sdag_centroid_sims for all queries of a retrieval batch in one launch: what select_malicious_docs_for_batch
 * (malicious_selection.py:140-180) computes query by query.  Query b owns rows r_off[b] .. r_off[b+1] of R (at
 * least one) and candidates c_off[b] .. c_off[b+1] of Cand; r_off / c_off: nb + 1 int32 offsets on the device.
 * centroid [nb, d] and sims [c_off[nb]] on the device; the same bits as nb calls of sdag_centroid_sims. */
int sdag_centroid_sims_batch(const float* R, const int32_t* r_off, const float* Cand, const int32_t* c_off, int nb, int d,
                             float* centroid, float* sims, void* stream);

#ifdef __cplusplus
}
#endif
#endif
