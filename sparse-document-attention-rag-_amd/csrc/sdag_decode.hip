// Single-token attention over the key/value cache for gfx950 (MI355X), bf16 in, fp32 softmax/accumulate.
//
// Replaces the attention of the reference's decode steps
//   out = model(input_ids=generated[:, -1:], past_key_values=past_key_values, use_cache=True)
// (src/pipeline/sparse_attention_RAG/SDAG.py:191-199): one query token per sequence, every cached key allowed
// (the reference passes no mask there), grouped-query heads, output [B, 1, Hq, D].
//
// The step is a pure stream over K and V (2 * S * D * 2 bytes per key/value head against 4 * G * S * D flops), so
// it runs on the vector ALUs and is laid out for memory parallelism: one wave per (sequence, key/value head,
// group of up to 4 query heads, part of the keys).  For every 64 keys of its part a wave
//   - reads one K row per lane (16 x 16 bytes) and takes the dot products with the (up to) 4 query rows, whose
//     bf16 pairs sit in scalar registers (v_dot2c_f32_bf16: exact products, fp32 sums);
//   - does the online-softmax bookkeeping per query head with wave reductions;
//   - accumulates P V with each lane owning two output columns: V rows are read coalesced (one 256-byte row per
//     wave instruction), the probability of a key comes out of its lane with v_readlane.
// Parts are merged by a second kernel (log-sum-exp weights).
#include <hip/hip_runtime.h>

#include "common.h"

namespace sdag {
namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define AS4 __attribute__((address_space(4)))

constexpr int D = 128;
constexpr int GW = 4;            // query heads per wave
constexpr int PART_STRIDE = D + 2;  // floats per partial: o[128], m, l
constexpr int TARGET_WAVES = 2048;  // waves a launch aims for: what the chip holds at two per SIMD

struct DecodeParams {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    void* o;
    int64_t q_sb, q_sh, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh;
    int B, Hq, Hkv, G, HG, S;
    int n_parts, groups_per_part;  // a part is groups_per_part groups of 64 keys
    int out_f32;
    float c;       // scale * log2(e)
    float* part;   // [B][Hq][n_parts][PART_STRIDE]
    // Right-padded prompts decoded together: keys [valid_len[b], pad_end) of sequence b are padding and take no part
    // (valid_len == nullptr: every key below S counts).
    const int32_t* valid_len;
    int pad_end;
    // A preallocated cache: only rows below *n_rows (on the device, 1 <= *n_rows <= S) hold keys; nullptr: all S rows.
    const int32_t* n_rows;
};

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

__global__ __launch_bounds__(64) void sdag_decode_main(const DecodeParams p) {
    const int lane = threadIdx.x;
    int bid = blockIdx.x;
    const int part = bid % p.n_parts; bid /= p.n_parts;
    const int hg = bid % p.HG; bid /= p.HG;
    const int kvh = bid % p.Hkv;
    const int b = bid / p.Hkv;
    const int h0 = kvh * p.G + hg * GW;              // first query head of this wave
    const int nh = min(GW, p.G - hg * GW);           // heads it really has

    // The query rows are wave-uniform: bf16 pairs through scalar loads (rows beyond nh repeat the last one).
    const AS4 unsigned* qrow[GW];
#pragma unroll
    for (int j = 0; j < GW; ++j)
        qrow[j] = (const AS4 unsigned*)(p.q + (int64_t)b * p.q_sb + (int64_t)(h0 + min(j, nh - 1)) * p.q_sh);

    const unsigned short* kh = p.k + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
    const unsigned short* vh = p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;

    float m[GW], l[GW], o0[GW], o1[GW];  // running max / sum (uniform), this lane's two output columns
#pragma unroll
    for (int j = 0; j < GW; ++j) { m[j] = -INFINITY; l[j] = 0.f; o0[j] = 0.f; o1[j] = 0.f; }

    const int S = p.n_rows ? max(1, min(p.S, *p.n_rows)) : p.S;  // wave-uniform (a scalar load)
    const int g_begin = part * p.groups_per_part;
    const int g_end = min(g_begin + p.groups_per_part, (S + 63) / 64);
    const int hole_lo = p.valid_len ? p.valid_len[b] : S, hole_hi = p.valid_len ? p.pad_end : S;  // wave-uniform
    for (int g = g_begin; g < g_end; ++g) {
        if (g * 64 >= hole_lo && g * 64 + 64 <= hole_hi) continue;  // 64 keys of padding
        const int key = g * 64 + lane;
        const bool live = key < S && !(key >= hole_lo && key < hole_hi);
        // ---- all loads of the group first: this lane's K row (16 x 16 bytes) and, coalesced, one 4-byte column pair
        // of each of the 64 V rows, so that the V rows travel while the scores are computed (rows past the end of
        // the sequence are clamped to the last key; their probability is 0)
        const u32x4* krow = reinterpret_cast<const u32x4*>(kh + (int64_t)min(key, S - 1) * p.k_sl);
        u32x4 kreg[16];
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) kreg[ch] = krow[ch];
        const unsigned* vbase = reinterpret_cast<const unsigned*>(vh) + lane;
        unsigned vw[64];
#pragma unroll
        for (int t = 0; t < 64; ++t) vw[t] = vbase[(int64_t)min(g * 64 + t, S - 1) * (p.v_sl / 2)];
        // ---- scores of this lane's key against the query rows
        float s[GW] = {0.f, 0.f, 0.f, 0.f};
        // the query words are re-read through the scalar cache for every group of keys: 256 of them do not fit
        // the scalar registers, and keeping the pointers opaque stops the compiler from hoisting the loads
        // out of the loop into spilled registers
#pragma unroll
        for (int j = 0; j < GW; ++j) asm volatile("" : "+s"(qrow[j]));
#pragma unroll
        for (int ch = 0; ch < 16; ++ch) {
            const u32x4 kk = kreg[ch];
            const unsigned kw[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int j = 0; j < GW; ++j) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned qw = qrow[j][ch * 4 + e];
                    s[j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, kw[e]),
                                                           __builtin_bit_cast(bf16x2, qw), s[j], false);
                }
            }
        }
        // ---- online softmax per head (base-2 exponent of scaled scores)
        float pr[GW];
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            const float x = live ? s[j] * p.c : -INFINITY;
            const float m_new = fmaxf(m[j], wave_max(x));
            const float alpha = __builtin_amdgcn_exp2f(m[j] - m_new);  // m = -inf on the first group: 0
            pr[j] = __builtin_amdgcn_exp2f(x - m_new);
            l[j] = l[j] * alpha + wave_sum(pr[j]);
            o0[j] *= alpha;
            o1[j] *= alpha;
            m[j] = m_new;
        }
        // ---- O += P V: each key's probability read out of the key's lane
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const unsigned w = vw[t];
            const float v0 = __uint_as_float(w << 16), v1 = __uint_as_float(w & 0xffff0000u);
#pragma unroll
            for (int j = 0; j < GW; ++j) {
                const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr[j]), t));
                o0[j] = __builtin_fmaf(pj, v0, o0[j]);
                o1[j] = __builtin_fmaf(pj, v1, o1[j]);
            }
        }
    }
    // ---- this part's result
#pragma unroll
    for (int j = 0; j < GW; ++j) {
        if (j >= nh) break;
        float* dst = p.part + (((int64_t)b * p.Hq + h0 + j) * p.n_parts + part) * PART_STRIDE;
        reinterpret_cast<float2*>(dst)[lane] = make_float2(o0[j], o1[j]);
        if (lane == 0) {
            dst[D] = m[j];
            dst[D + 1] = l[j];
        }
    }
}

// One block of 256 threads per (sequence, query head): log-sum-exp merge of the parts.  Thread -> column pair
// (tid & 63), parts tid >> 6, + 4, + 8, ...: every load is independent of the others, and the four partial sums per
// column meet in LDS.
__global__ __launch_bounds__(256) void sdag_decode_merge(const DecodeParams p) {
    __shared__ float red_m[4];
    __shared__ float red_den[4];
    __shared__ float2 red_o[4][64];
    const int hq = blockIdx.x % p.Hq, b = blockIdx.x / p.Hq;
    const int tid = threadIdx.x, cp = tid & 63, pg = tid >> 6;
    const float* src = p.part + ((int64_t)b * p.Hq + hq) * p.n_parts * PART_STRIDE;
    // the largest running maximum of all parts (at most 128 parts: one per thread of the first two waves)
    float mx = tid < p.n_parts ? src[tid * PART_STRIDE + D] : -INFINITY;
    mx = wave_max(mx);
    if (cp == 0) red_m[pg] = mx;
    __syncthreads();
    const float mmax = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    float2 acc = make_float2(0.f, 0.f);
    float den = 0.f;
#pragma unroll 4
    for (int t = pg; t < p.n_parts; t += 4) {
        const float* s = src + t * PART_STRIDE;
        const float w = __builtin_amdgcn_exp2f(s[D] - mmax);  // empty part: m = -inf -> 0
        const float2 x = reinterpret_cast<const float2*>(s)[cp];
        den += w * s[D + 1];
        acc.x += w * x.x;
        acc.y += w * x.y;
    }
    red_o[pg][cp] = acc;
    if (cp == 0) red_den[pg] = den;
    __syncthreads();
    if (pg != 0) return;
    const float dsum = (red_den[0] + red_den[1]) + (red_den[2] + red_den[3]);
    const float inv = 1.0f / dsum;
    const float r0 = ((red_o[0][cp].x + red_o[1][cp].x) + (red_o[2][cp].x + red_o[3][cp].x)) * inv;
    const float r1 = ((red_o[0][cp].y + red_o[1][cp].y) + (red_o[2][cp].y + red_o[3][cp].y)) * inv;
    const int64_t oi = (int64_t)b * p.o_sb + (int64_t)hq * p.o_sh + 2 * cp;
    if (p.out_f32) {
        *reinterpret_cast<float2*>(reinterpret_cast<float*>(p.o) + oi) = make_float2(r0, r1);
    } else {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        bf16x2_t w = {(__bf16)r0, (__bf16)r1};
        *reinterpret_cast<bf16x2_t*>(reinterpret_cast<__bf16*>(p.o) + oi) = w;
    }
}

// Upper bound of the parts per (sequence, head): what the workspace is sized for.
void plan(int S, int* n_parts, int* groups_per_part) {
    const int groups = (S + 63) / 64;
    const int gpp = (groups + 127) / 128;  // at most 128 parts per head
    *groups_per_part = gpp;
    *n_parts = (groups + gpp - 1) / gpp;
}

// Parts of a launch with `pairs` (sequence, key/value head, head group) triples: enough waves to fill the chip once,
// no more (every part costs a partial result and a share of the merge, and a part of one 64-key group has no steady
// state).  Measured at S = 3100: 32 sequences 89.9 -> 85.5 us, 128 sequences 344 -> 315 us; 1 to 8 sequences unchanged.
void plan_launch(int S, int64_t pairs, int* n_parts, int* groups_per_part) {
    int np_max, gpp_min;
    plan(S, &np_max, &gpp_min);
    const int groups = (S + 63) / 64;
    int64_t want = (TARGET_WAVES + pairs - 1) / pairs;
    if (want < 1) want = 1;
    if (want > np_max) want = np_max;
    int gpp = (int)((groups + want - 1) / want);
    if (gpp < gpp_min) gpp = gpp_min;
    *groups_per_part = gpp;
    *n_parts = (groups + gpp - 1) / gpp;
}

}  // namespace
}  // namespace sdag

using namespace sdag;

extern "C" int64_t sdag_decode_workspace_bytes(int B, int Hq, int S) {
    if (B <= 0 || Hq <= 0 || S <= 0) return -1;
    int n_parts, gpp;
    plan(S, &n_parts, &gpp);
    return (((int64_t)B * Hq * n_parts * PART_STRIDE * (int64_t)sizeof(float)) + 255) & ~255ll;
}

namespace {
int decode_launch(const void* q, const void* k, const void* v, void* o,
                                int64_t q_sb, int64_t q_sh,
                                int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                int64_t v_sb, int64_t v_sh, int64_t v_sl,
                                int64_t o_sb, int64_t o_sh,
                                int B, int Hq, int Hkv, int S, int Dh, float scale, const int32_t* valid_len, int pad_end,
                                const int32_t* n_rows, void* workspace, int64_t workspace_bytes_, int out_dtype, void* stream) {
    if (Dh != D) return fail(SDAG_ERR_ARG, "head_dim %d not supported (only 128)", Dh);
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || S <= 0 || Hq % Hkv) return fail(SDAG_ERR_ARG, "bad B/Hq/Hkv/S");
    if (!(scale > 0.f)) return fail(SDAG_ERR_ARG, "scale must be positive");
    if (out_dtype != SDAG_OUT_BF16 && out_dtype != SDAG_OUT_F32) return fail(SDAG_ERR_ARG, "bad out_dtype");
    if (!q || !k || !v || !o) return fail(SDAG_ERR_ARG, "NULL tensor pointer");
    const int64_t strides[] = {q_sb, q_sh, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh};
    for (int64_t st : strides)
        if (st < 0 || (st & 7)) return fail(SDAG_ERR_ARG, "strides must be non-negative multiples of 8 elements");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return fail(SDAG_ERR_ARG, "tensor pointers must be 16-byte aligned");
    const int64_t need = sdag_decode_workspace_bytes(B, Hq, S);
    if (!workspace || workspace_bytes_ < need) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)need);
    DecodeParams p;
    p.q = (const unsigned short*)q; p.k = (const unsigned short*)k; p.v = (const unsigned short*)v; p.o = o;
    p.q_sb = q_sb; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
    p.v_sb = v_sb; p.v_sh = v_sh; p.v_sl = v_sl; p.o_sb = o_sb; p.o_sh = o_sh;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.G = Hq / Hkv; p.HG = (p.G + GW - 1) / GW; p.S = S;
    plan_launch(S, (int64_t)B * Hkv * p.HG, &p.n_parts, &p.groups_per_part);
    p.out_f32 = out_dtype == SDAG_OUT_F32;
    p.c = scale * 1.4426950408889634f;
    p.part = (float*)workspace;
    p.valid_len = valid_len;
    p.pad_end = pad_end;
    p.n_rows = n_rows;
    const int64_t blocks = (int64_t)B * Hkv * p.HG * p.n_parts;
    if (blocks > 0x7fffffff) return fail(SDAG_ERR_ARG, "grid too large");
    hipLaunchKernelGGL(sdag_decode_main, dim3((unsigned)blocks), dim3(64), 0, (hipStream_t)stream, p);
    SDAG_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(sdag_decode_merge, dim3((unsigned)(B * Hq)), dim3(256), 0, (hipStream_t)stream, p);
    SDAG_HIP_CHECK(hipGetLastError());
    return SDAG_OK;
}
}  // namespace

extern "C" int sdag_decode_attn(const void* q, const void* k, const void* v, void* o,
                                int64_t q_sb, int64_t q_sh,
                                int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                int64_t v_sb, int64_t v_sh, int64_t v_sl,
                                int64_t o_sb, int64_t o_sh,
                                int B, int Hq, int Hkv, int S, int Dh, float scale,
                                void* workspace, int64_t workspace_bytes_, int out_dtype, void* stream) {
    return decode_launch(q, k, v, o, q_sb, q_sh, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, B, Hq, Hkv, S, Dh, scale, nullptr, 0,
                         nullptr, workspace, workspace_bytes_, out_dtype, stream);
}

extern "C" int sdag_decode_attn_padded(const void* q, const void* k, const void* v, void* o,
                                       int64_t q_sb, int64_t q_sh,
                                       int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                       int64_t v_sb, int64_t v_sh, int64_t v_sl,
                                       int64_t o_sb, int64_t o_sh,
                                       int B, int Hq, int Hkv, int S, int Dh, float scale,
                                       const int32_t* valid_len, int pad_end,
                                       void* workspace, int64_t workspace_bytes_, int out_dtype, void* stream) {
    if (!valid_len) return fail(SDAG_ERR_ARG, "valid_len is NULL");
    if (pad_end < 0 || pad_end > S) return fail(SDAG_ERR_ARG, "pad_end outside [0, S]");
    return decode_launch(q, k, v, o, q_sb, q_sh, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, B, Hq, Hkv, S, Dh, scale, valid_len,
                         pad_end, nullptr, workspace, workspace_bytes_, out_dtype, stream);
}

extern "C" int sdag_decode_attn_prealloc(const void* q, const void* k, const void* v, void* o,
                                         int64_t q_sb, int64_t q_sh,
                                         int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                         int64_t v_sb, int64_t v_sh, int64_t v_sl,
                                         int64_t o_sb, int64_t o_sh,
                                         int B, int Hq, int Hkv, int S, int Dh, float scale,
                                         const int32_t* n_rows, const int32_t* valid_len, int pad_end,
                                         void* workspace, int64_t workspace_bytes_, int out_dtype, void* stream) {
    if (!n_rows) return fail(SDAG_ERR_ARG, "n_rows is NULL");
    if (valid_len && (pad_end < 0 || pad_end > S)) return fail(SDAG_ERR_ARG, "pad_end outside [0, S]");
    return decode_launch(q, k, v, o, q_sb, q_sh, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh, B, Hq, Hkv, S, Dh, scale, valid_len,
                         valid_len ? pad_end : 0, n_rows, workspace, workspace_bytes_, out_dtype, stream);
}
