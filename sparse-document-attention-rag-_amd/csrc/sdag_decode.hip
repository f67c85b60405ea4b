// Single-token attention over the key/value cache for gfx950 (MI355X), bf16 in, fp32 softmax/accumulate.
//
// Replaces the attention of the reference's decode steps
//   out = model(input_ids=generated[:, -1:], past_key_values=past_key_values, use_cache=True)
// (src/pipeline/sparse_attention_RAG/SDAG.py:191-199): one query token per sequence, every cached key allowed
// (the reference passes no mask there), grouped-query heads, output [B, 1, Hq, D].
//
// The step is a pure stream over K and V (2 * S * D * 2 bytes per key/value head against 4 * G * S * D flops).  The
// keys that count (all of them, minus the padding rows of a right-padded batch) are numbered 0 .. n_live - 1 and cut
// into one contiguous range per wave; a workgroup is up to 8 waves of one (sequence, key/value head, group of up to 4
// query heads) and joins its waves' results in LDS, so a launch with enough sequences needs no second kernel.  Per
// window of 64 keys a wave
//   - holds the K rows as the A operand of v_mfma_f32_16x16x32_bf16 (16 rows x 64 bytes per load instruction), the
//     query rows as the B operand (loaded once): 16 MFMAs give the scores of 64 keys x 4 heads (exact products,
//     fp32 sums), head h in the lanes l with l % 16 == h, 16 keys per lane;
//   - does the online-softmax bookkeeping per lane for that lane's head (two lane swaps per window for the maximum;
//     the sum of the probabilities stays per lane until the end);
//   - hands the probabilities to the lanes that hold V through 1 KiB of wave-private LDS;
//   - accumulates P V with V rows read whole (4 rows x 256 bytes per load instruction): lane l owns the 8 columns
//     8 (l % 16) .. + 7 for the 16 keys 16 (l / 16) .. + 15 of the window (v_pk_fma_f32).
// The loads of the next window are issued while the current one is computed (K behind the MFMAs, every V row behind
// its last use).  Rows that do not count are never read as data: their row index is clamped to a row that counts.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"

namespace sdag {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int D = 128;
constexpr int GW = 4;               // query heads per wave
constexpr int MAX_W = 8;            // waves per workgroup (two per SIMD: the kernel needs most of 256 registers)
constexpr int PART_STRIDE = D + 2;  // floats per partial: o[128], m, l
constexpr int TARGET_WAVES = 2048;  // waves a launch aims for: what the chip holds at two per SIMD
// LDS words per wave: probabilities [4 heads][64 keys] + 4 rescale factors (padded), joined output [4][128], m and l [4][2]
constexpr int LDS_P = GW * 64 + 16;
constexpr int LDS_O = GW * D;
constexpr int LDS_ML = GW * 2;
constexpr int LDS_WAVE = LDS_P + LDS_O + LDS_ML;

struct DecodeParams {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    void* o;
    int64_t q_sb, q_sh, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sh;
    int B, Hq, Hkv, G, HG, S;
    int n_parts, W;  // workgroups per (sequence, key/value head, head group) and waves per workgroup
    int out_f32;
    float c;       // scale * log2(e)
    float* part;   // [B][Hq][n_parts][PART_STRIDE] (n_parts > 1 only)
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

// x of this lane and of the lanes 16, 32 and 48 away, combined (no LDS).  v_permlane16_swap_b32 a, b trades the odd
// 16-lane rows of a with the even rows of b, v_permlane32_swap_b32 lanes 32-63 of a with lanes 0-31 of b; with a = b = x
// the two results hold x of this lane and of its partner.  Inline asm with the wait states the instructions need
// around them: given the same value twice, the compiler's builtins fold the two results into one.
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float quad_rows_max(float x) {
    float a = x, b = x;
    swap16(a, b);
    a = b = fmaxf(a, b);
    swap32(a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float quad_rows_sum(float x) {
    float a = x, b = x;
    swap16(a, b);
    a = b = a + b;
    swap32(a, b);
    return a + b;
}

// 16 bytes of the cache (read once per step: non-temporal unless SDAG_DECODE_NT=0)
#ifndef SDAG_DECODE_NT
#define SDAG_DECODE_NT 1
#endif
__device__ __forceinline__ u32x4 ld16(const char* ptr) {
#if SDAG_DECODE_NT
    return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
#else
    return *reinterpret_cast<const u32x4*>(ptr);
#endif
}
__device__ __forceinline__ u32x4 load16(const unsigned short* base, int64_t byte_off) {
    return ld16(reinterpret_cast<const char*>(base) + byte_off);
}

__global__ __launch_bounds__(64 * MAX_W) void sdag_decode_main(const DecodeParams p) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (uniform: tell the compiler)
    const int lr = lane & 15, lq = lane >> 4;  // MFMA: row / column index, 8-element block of the reduced dimension
    int bid = blockIdx.x;
    const int part = bid % p.n_parts; bid /= p.n_parts;
    const int hg = bid % p.HG; bid /= p.HG;
    const int kvh = bid % p.Hkv;
    const int b = bid / p.Hkv;
    const int h0 = kvh * p.G + hg * GW;              // first query head of this workgroup
    const int nh = min(GW, p.G - hg * GW);           // heads it really has
    float* pbuf = lds + w * LDS_WAVE;                // this wave's probabilities
    float* obuf = pbuf + LDS_P;
    float* mlbuf = obuf + LDS_O;

    const unsigned short* kh = p.k + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh;
    const unsigned short* vh = p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh;

    // B operand: Q[head lr][32 kk + 8 lq .. + 7], zero for the columns past the last head
    bf16x8 qb[4];
    {
        const unsigned short* qrow = p.q + (int64_t)b * p.q_sb + (int64_t)(h0 + min(lr, nh - 1)) * p.q_sh + 8 * lq;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4 x = *reinterpret_cast<const u32x4*>(qrow + 32 * kk);
            if (lr >= nh) x = u32x4{0u, 0u, 0u, 0u};
            qb[kk] = __builtin_bit_cast(bf16x8, x);
        }
    }

    // the keys that count, numbered 0 .. n_live - 1: key j is row j below the padding and row j + hole_len above it
    // (wave-uniform values; readfirstlane tells the compiler, which cannot prove it for loads that stores may alias)
    const int S = p.n_rows ? max(1, min(p.S, __builtin_amdgcn_readfirstlane(*p.n_rows))) : p.S;
    int hole_lo = p.valid_len ? max(0, min(__builtin_amdgcn_readfirstlane(p.valid_len[b]), S)) : S;
    const int hole_len = p.valid_len ? max(0, min(p.pad_end, S) - hole_lo) : 0;
    if (hole_len == 0) hole_lo = S;  // no padding: one segment, the windows (and the bits) of the plain call
    const int n_live = S - hole_len;
    const int T = p.n_parts * p.W, wi = part * p.W + w;
    const int per_wave = (n_live + T - 1) / T;
    const int r_begin = min(wi * per_wave, n_live), r_end = min(r_begin + per_wave, n_live);

    float m = -INFINITY, l = 0.f;  // running maximum (the same in the four lanes of a head) and this lane's sum
    f32x2 acc[GW][4];              // columns 8 lr + 2 d, + 1 of head j, over this lane's keys
#pragma unroll
    for (int j = 0; j < GW; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[j][d] = f32x2{0.f, 0.f};

    const int64_t k_row_bytes = p.k_sl * 2, v_row_bytes = p.v_sl * 2;
    const unsigned k_lane_off = (unsigned)(lr * k_row_bytes + 16 * lq);         // A operand: row lr, dims 8 lq .. of a 32-block
    const unsigned v_lane_off = (unsigned)(16 * lq * v_row_bytes + 16 * lr);    // V: row 16 lq (+ r), columns 8 lr ..

    // the range in at most two segments, one on each side of the padding
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const int a = seg == 0 ? r_begin : max(r_begin, hole_lo);
        const int e = seg == 0 ? min(r_end, hole_lo) : r_end;
        if (a >= e) continue;
        const int shift = seg == 0 ? 0 : hole_len;  // row = key number + shift
        // windows: the first takes the (e - a) % 64 keys that do not fill a window (or a whole one), the others are full
        int nv = (e - a) & 63;
        if (nv == 0) nv = 64;
        u32x4 kreg[4][4], vreg[16];
        {   // first window: row indices clamped per lane to its last key
            const int last = a + nv - 1 + shift;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t off = (int64_t)min(a + shift + 16 * t + lr, last) * k_row_bytes + 16 * lq;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) kreg[t][kk] = load16(kh, off + 64 * kk);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                vreg[r] = load16(vh, (int64_t)min(a + shift + 16 * lq + r, last) * v_row_bytes + 16 * lr);
        }
        // one window; MORE: the loads of the next window (first key jn, always a whole window) are issued inside
        auto window = [&](auto more_tag, const int jn) {
            constexpr bool more = decltype(more_tag)::value;
            // ---- scores: 64 keys x 4 heads; this lane: head lr, keys 16 t + 4 lq + i
            f32x4 s[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kreg[t][kk]), qb[kk], z, 0, 0, 0);
                s[t] = z;
            }
            if constexpr (more) {  // the next window's K rows (whole windows only: one row base for the wave)
                const unsigned short* kb = kh + (int64_t)(jn + shift) * p.k_sl;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const char* kt = reinterpret_cast<const char*>(kb) + (int64_t)(16 * t) * k_row_bytes;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) kreg[t][kk] = ld16(kt + k_lane_off + 64 * kk);
                }
            }
            // ---- online softmax for this lane's head (base-2 exponent of scaled scores)
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = s[t][i] * p.c;
                    if (16 * t + 4 * lq + i >= nv) x = -INFINITY;
                    s[t][i] = x;
                    mx = fmaxf(mx, x);
                }
            const float m_new = fmaxf(m, quad_rows_max(mx));        // nv >= 1: finite unless a score is not
            const float alpha = m == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(m - m_new);  // (acc and l are 0 while m is -inf)
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pr = s[t][i] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[t][i] - m_new);
                    s[t][i] = pr;
                    psum += pr;
                }
            l = l * alpha + psum;
            m = m_new;
            // ---- probabilities to the lanes that hold V: [head][key of the window] in LDS, the rescale factor behind
            if (lr < GW) {
#pragma unroll
                for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(pbuf + lr * 64 + 16 * t + 4 * lq) = s[t];
                if (lq == 0) pbuf[GW * 64 + lr] = alpha;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const f32x4 al = *reinterpret_cast<const f32x4*>(pbuf + GW * 64);
#pragma unroll
            for (int j = 0; j < GW; ++j)
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[j][d] *= f32x2{al[j], al[j]};
            // ---- O += P V over this lane's 16 keys; every V row is replaced by the next window's behind its last use
            const char* vb = reinterpret_cast<const char*>(vh + (int64_t)(jn + shift) * p.v_sl);
#pragma unroll
            for (int rc = 0; rc < 4; ++rc) {
                f32x4 pc[GW];
#pragma unroll
                for (int j = 0; j < GW; ++j) pc[j] = *reinterpret_cast<const f32x4*>(pbuf + j * 64 + 16 * lq + 4 * rc);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * rc + rr;
                    const u32x4 vw = vreg[r];
                    const unsigned wds[4] = {vw.x, vw.y, vw.z, vw.w};
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const f32x2 vv = {__uint_as_float(wds[d] << 16), __uint_as_float(wds[d] & 0xffff0000u)};
#pragma unroll
                        for (int j = 0; j < GW; ++j) acc[j][d] = __builtin_elementwise_fma(f32x2{pc[j][rr], pc[j][rr]}, vv, acc[j][d]);
                    }
                    if constexpr (more) vreg[r] = ld16(vb + (int64_t)r * v_row_bytes + v_lane_off);
                }
            }
            __builtin_amdgcn_wave_barrier();  // the next window's probabilities overwrite pbuf: reads above come first
            nv = 64;
        };
        int jn = a + nv;
#pragma unroll 1
        for (; jn < e; jn += 64) window(std::true_type{}, jn);
        window(std::false_type{}, jn);
    }

    // ---- the wave's result: sums over the four key groups of the lanes (l / 16), then the workgroup's waves in LDS
    l = quad_rows_sum(l);
#pragma unroll
    for (int j = 0; j < GW; ++j)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc[j][d].x = quad_rows_sum(acc[j][d].x);
            acc[j][d].y = quad_rows_sum(acc[j][d].y);
        }
    if (lq == 0) {
#pragma unroll
        for (int j = 0; j < GW; ++j) {
            *reinterpret_cast<f32x4*>(obuf + j * D + 8 * lr) = f32x4{acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y};
            *reinterpret_cast<f32x4*>(obuf + j * D + 8 * lr + 4) = f32x4{acc[j][2].x, acc[j][2].y, acc[j][3].x, acc[j][3].y};
        }
        if (lr < GW) {
            mlbuf[2 * lr] = m;
            mlbuf[2 * lr + 1] = l;
        }
    }
    __syncthreads();
    const int nthreads = 64 * p.W;
    for (int idx = threadIdx.x; idx < nh * D; idx += nthreads) {
        const int j = idx >> 7, col = idx & (D - 1);
        float mmax = -INFINITY;
        for (int ww = 0; ww < p.W; ++ww) mmax = fmaxf(mmax, lds[ww * LDS_WAVE + LDS_P + LDS_O + 2 * j]);
        float num = 0.f, den = 0.f;
        for (int ww = 0; ww < p.W; ++ww) {
            const float* base = lds + ww * LDS_WAVE + LDS_P;
            const float mw = base[LDS_O + 2 * j];
            const float wgt = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - mmax);  // a wave without keys
            num += wgt * base[j * D + col];
            den += wgt * base[LDS_O + 2 * j + 1];
        }
        const int hq = h0 + j;
        if (p.n_parts > 1) {
            float* dst = p.part + (((int64_t)b * p.Hq + hq) * p.n_parts + part) * PART_STRIDE;
            dst[col] = num;
            if (col == 0) {
                dst[D] = mmax;
                dst[D + 1] = den;
            }
        } else {
            const float r = den > 0.f ? num / den : 0.f;
            const int64_t oi = (int64_t)b * p.o_sb + (int64_t)hq * p.o_sh + col;
            if (p.out_f32) reinterpret_cast<float*>(p.o)[oi] = r;
            else reinterpret_cast<__bf16*>(p.o)[oi] = (__bf16)r;
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
        const float w = s[D] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(s[D] - mmax);  // a part without keys
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
    const float inv = dsum > 0.f ? 1.0f / dsum : 0.f;
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

// Upper bound of the parts (workgroups) per (sequence, head): what the workspace is sized for.
constexpr int MAX_PARTS = 128;
void plan(int S, int* n_parts) {
    const int windows = (S + 63) / 64;
    const int np = (windows + MAX_W - 1) / MAX_W;
    *n_parts = np < MAX_PARTS ? np : MAX_PARTS;
}

// Shape of a launch with `pairs` (sequence, key/value head, head group) triples: enough waves to fill the chip once
// (two per SIMD), no more, and no wave with less than one window of 64 keys; up to 8 waves share a workgroup and join
// their results there, so with 256 triples or more a launch is one workgroup per triple and needs no merge kernel.
void plan_launch(int S, int64_t pairs, int* n_parts, int* W) {
    const int windows = (S + 63) / 64;
    int64_t want = (TARGET_WAVES + pairs - 1) / pairs;  // waves per triple
    if (want > windows) want = windows;
    if (want < 1) want = 1;
    if (want <= MAX_W) {
        *W = (int)want;
        *n_parts = 1;
        return;
    }
    *W = MAX_W;
    int np_max;
    plan(S, &np_max);
    const int64_t np = (want + MAX_W - 1) / MAX_W;
    *n_parts = np < np_max ? (int)np : np_max;
}

}  // namespace
}  // namespace sdag

using namespace sdag;

extern "C" int64_t sdag_decode_workspace_bytes(int B, int Hq, int S) {
    if (B <= 0 || Hq <= 0 || S <= 0) return -1;
    int n_parts;
    plan(S, &n_parts);
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
    if (k_sl > (1 << 22) || v_sl > (1 << 22)) return fail(SDAG_ERR_ARG, "row strides of the cache must be at most 2^22 elements");
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return fail(SDAG_ERR_ARG, "tensor pointers must be 16-byte aligned");
    const int64_t need = sdag_decode_workspace_bytes(B, Hq, S);
    if (!workspace || workspace_bytes_ < need) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)need);
    DecodeParams p;
    p.q = (const unsigned short*)q; p.k = (const unsigned short*)k; p.v = (const unsigned short*)v; p.o = o;
    p.q_sb = q_sb; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sh = k_sh; p.k_sl = k_sl;
    p.v_sb = v_sb; p.v_sh = v_sh; p.v_sl = v_sl; p.o_sb = o_sb; p.o_sh = o_sh;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv; p.G = Hq / Hkv; p.HG = (p.G + GW - 1) / GW; p.S = S;
    plan_launch(S, (int64_t)B * Hkv * p.HG, &p.n_parts, &p.W);
    p.out_f32 = out_dtype == SDAG_OUT_F32;
    p.c = scale * 1.4426950408889634f;
    p.part = (float*)workspace;
    p.valid_len = valid_len;
    p.pad_end = pad_end;
    p.n_rows = n_rows;
    const int64_t blocks = (int64_t)B * Hkv * p.HG * p.n_parts;
    if (blocks > 0x7fffffff) return fail(SDAG_ERR_ARG, "grid too large");
    hipLaunchKernelGGL(sdag_decode_main, dim3((unsigned)blocks), dim3(64 * p.W), (size_t)p.W * LDS_WAVE * sizeof(float),
                       (hipStream_t)stream, p);
    SDAG_HIP_CHECK(hipGetLastError());
    if (p.n_parts > 1) {
        hipLaunchKernelGGL(sdag_decode_merge, dim3((unsigned)(B * Hq)), dim3(256), 0, (hipStream_t)stream, p);
        SDAG_HIP_CHECK(hipGetLastError());
    }
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
