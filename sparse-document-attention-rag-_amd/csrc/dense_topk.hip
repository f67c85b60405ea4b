// Exact inner-product search on gfx950 (MI355X): fused fp32 GEMM-scan + running top-k, and the merge of
// partial top-k lists.
//
// Replaces   scores, indices = index.search(q.astype(np.float32), top_k)
// (src/pipeline/retrieval/dense.py:82-84), whose arithmetic the reference delegates to faiss.
//
// Arithmetic contract (shared with the CPU oracle): score(q, c) is a single fp32 FMA chain over the
// dimensions in increasing order starting from +0; results are ordered by (score descending, id ascending).
// v_mfma_f32_32x32x2_f32 computes D = fma(a1, b1, fma(a0, b0, C)) per output element, so chaining the
// accumulator through d/2 MFMAs in dimension order reproduces that chain bit for bit.
//
// Scan kernel: every workgroup (4 waves) owns a contiguous stripe of corpus rows and walks it in tiles of
// 128 rows; wave w multiplies rows 32 w .. 32 w + 31 of the tile against up to 32 queries.  Corpus and query
// slices of 32 dimensions are brought into a 3-stage LDS ring by LDS-DMA (global_load_lds_dwordx4, one
// 1 KiB piece per wave instruction, 128-byte source lines, XOR-swizzled through the SOURCE address so that
// the later ds_read_b128 fragment reads are bank-conflict free).  Loads run two stages ahead of the MFMAs
// behind a counted s_waitcnt vmcnt + raw s_barrier, so HBM latency stays hidden while the matrix pipe runs
// the 64-cycle fp32 MFMAs back to back.  After the last slice of a tile the 32 x 32 scores of each wave are
// filtered against a per-query threshold (the current k-th best of the stripe) and survivors are appended to
// per-query candidate lists in LDS, which are compacted to their best k whenever they could overflow.
// Each workgroup finally writes its best k per query; a merge kernel reduces those lists.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>

#include "common.h"

namespace sdag {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define SDAG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define SDAG_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int QT = 32;                      // queries per pass
constexpr int TR = 128;                     // corpus rows per tile
constexpr int KC = 32;                      // dimensions per stage
constexpr int NST = 3;                      // LDS ring depth
constexpr int C_STAGE = TR * KC * 4;        // 16 KiB
constexpr int Q_STAGE = QT * KC * 4;        // 4 KiB
constexpr int STAGE_BYTES = C_STAGE + Q_STAGE;
constexpr int CAP = 64;                     // candidate list capacity per query
constexpr int PUSH_MAX = 32;                // most entries one filter round can append to one list
constexpr int MAX_K = 32;
constexpr int MAX_WG = 512;
constexpr int OFF_CS = NST * STAGE_BYTES;           // float cand_s[QT][CAP]
constexpr int OFF_CI = OFF_CS + QT * CAP * 4;       // int   cand_i[QT][CAP]
constexpr int OFF_CNT = OFF_CI + QT * CAP * 4;      // int   cnt[QT]
constexpr int OFF_TS = OFF_CNT + QT * 4;            // float thr_s[QT]
constexpr int OFF_TI = OFF_TS + QT * 4;             // int   thr_i[QT]
constexpr int SCAN_LDS = OFF_TI + QT * 4;

__device__ __forceinline__ bool better(float s, int64_t i, float s2, int64_t i2) {
    return (s > s2) || (s == s2 && i < i2);
}

// s_waitcnt for LDS traffic + workgroup barrier that leaves vector-memory operations (the LDS-DMA ring) in
// flight; __syncthreads() would drain them.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// v_permlane32_swap_b32 x, y: lanes 32-63 of x trade places with lanes 0-31 of y.  Written as inline asm with
// its own wait states (a VALU write of an operand needs two before the swap reads it, and the MFMA that
// consumes the result is padded the same way); the compiler builtin of the same name folds both results
// into one register on ROCm 7.2.
__device__ __forceinline__ void swap_halves(float& x, float& y) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}

// One wave reduces list j to its best min(n, k) entries, sorted best first, and refreshes the threshold.
__device__ __forceinline__ void compact_list(unsigned char* smem, int j, int k, int lane) {
    float* cs = reinterpret_cast<float*>(smem + OFF_CS) + j * CAP;
    int* ci = reinterpret_cast<int*>(smem + OFF_CI) + j * CAP;
    int* cnt = reinterpret_cast<int*>(smem + OFF_CNT);
    const int n = __builtin_amdgcn_readfirstlane(cnt[j]);
    float s = lane < n ? cs[lane] : -INFINITY;
    int id = lane < n ? ci[lane] : INT_MAX;
    int rank = 0;
    for (int t = 0; t < n; ++t) {
        const float st = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), t));
        const int it = __builtin_amdgcn_readlane(id, t);
        rank += better(st, it, s, id) ? 1 : 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < n && rank < k) {
        cs[rank] = s;
        ci[rank] = id;
    }
    if (n >= k && lane < n && rank == k - 1) {
        reinterpret_cast<float*>(smem + OFF_TS)[j] = s;
        reinterpret_cast<int*>(smem + OFF_TI)[j] = id;
    }
    if (lane == 0) cnt[j] = n < k ? n : k;
}

__global__ __launch_bounds__(256, 2) void dense_scan(const float* __restrict__ Q, int nq, const float* __restrict__ C,
                                                     int64_t N, int d, int k, int rows_per_wg,
                                                     float* __restrict__ part_s, int* __restrict__ part_i) {
    // Dynamic LDS on purpose: the ring is filled only by LDS-DMA, which the optimiser does not see as a store, so
    // a static array would look never-written and its loads would be folded away.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t wg_row0 = (int64_t)blockIdx.x * rows_per_wg;
    const int64_t wg_rows = min((int64_t)rows_per_wg, N - wg_row0);
    const int ntile = (int)((wg_rows + TR - 1) / TR);
    const int nkc = d / KC;
    const int S = ntile * nkc;

    if (tid < QT) {
        reinterpret_cast<int*>(smem + OFF_CNT)[tid] = 0;
        reinterpret_cast<float*>(smem + OFF_TS)[tid] = -INFINITY;
        reinterpret_cast<int*>(smem + OFF_TI)[tid] = INT_MAX;
    }
    lds_barrier();

    // ---- LDS-DMA geometry.  A wave instruction writes 1 KiB = 8 rows x 128 B; lane -> (row lane >> 3,
    // physical chunk lane & 7); the source chunk is physical ^ ((row >> 1) & 7).
    const int dr = lane >> 3, dc = lane & 7;
    auto issue = [&](int s) {
        s = s < S ? s : S - 1;  // past the end: reload the last stage into the free slot (keeps vmcnt counts fixed)
        const int tile = s / nkc, kc = s - tile * nkc;
        unsigned char* sb = smem + (s % NST) * STAGE_BYTES;
        const int64_t trow0 = wg_row0 + (int64_t)tile * TR + w * 32;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rl = n * 8 + dr;  // row inside the wave's 32
            int64_t row = trow0 + rl;
            row = row < N ? row : N - 1;
            const int csrc = dc ^ ((rl >> 1) & 7);
            const float* src = C + row * d + kc * KC + csrc * 4;
            __builtin_amdgcn_global_load_lds(SDAG_GLB_PTR(src), SDAG_LDS_PTR(sb + (w * 4 + n) * 1024), 16, 0, 0);
        }
        {
            int j = w * 8 + dr;
            const int csrc = dc ^ ((j >> 1) & 7);
            j = j < nq ? j : nq - 1;
            const float* src = Q + (int64_t)j * d + kc * KC + csrc * 4;
            __builtin_amdgcn_global_load_lds(SDAG_GLB_PTR(src), SDAG_LDS_PTR(sb + C_STAGE + w * 1024), 16, 0, 0);
        }
    };

    const int r = lane & 31, h = lane >> 5;
    const int swz = (r >> 1) & 7;
    const int a_off = (w * 32 + r) * 128;
    const int b_off = C_STAGE + r * 128;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    issue(0);
    issue(1);
    int tile = 0, kc = 0;
    for (int s = 0; s < S; ++s) {
        // stage s landed for this wave (5 newer DMA pieces may still fly), then for every wave
        asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
        issue(s + 2);
        const unsigned char* sb = smem + (s % NST) * STAGE_BYTES;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ch = ((2 * m + h) ^ swz) << 4;
            const f32x4 a = *reinterpret_cast<const f32x4*>(sb + a_off + ch);
            const f32x4 b = *reinterpret_cast<const f32x4*>(sb + b_off + ch);
            // Lane half 0 holds dimensions 8m..8m+3, half 1 holds 8m+4..8m+7.  Swapping the upper half of
            // register 0 with the lower half of register 1 (and 2 with 3) leaves register 0 = (8m, 8m+1),
            // register 2 = (8m+2, 8m+3), register 1 = (8m+4, 8m+5), register 3 = (8m+6, 8m+7) across the
            // two halves: the MFMA operand layout, in dimension order.
            float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
            swap_halves(a0, a1);
            swap_halves(a2, a3);
            swap_halves(b0, b1);
            swap_halves(b2, b3);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc, 0, 0, 0);
        }
        if (++kc < nkc) continue;
        kc = 0;

        // ---- tile finished: accumulator register i is corpus row 32 w + (i & 3) + 8 (i >> 2) + 4 h of the
        // tile, column r is the query.
        const int64_t row_base = wg_row0 + (int64_t)tile * TR + w * 32 + 4 * h;
        ++tile;
        float* cs = reinterpret_cast<float*>(smem + OFF_CS);
        int* ci = reinterpret_cast<int*>(smem + OFF_CI);
        int* cnt = reinterpret_cast<int*>(smem + OFF_CNT);
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const float ts = reinterpret_cast<const float*>(smem + OFF_TS)[r];
            const int ti = reinterpret_cast<const int*>(smem + OFF_TI)[r];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = round * 4 + e;
                const float sc = acc[i];
                const int64_t row = row_base + (i & 3) + 8 * (i >> 2);
                if (r < nq && row < N && sc == sc && better(sc, (int)row, ts, ti)) {
                    const int pos = atomicAdd(&cnt[r], 1);
                    cs[r * CAP + pos] = sc;
                    ci[r * CAP + pos] = (int)row;
                }
            }
            lds_barrier();
#pragma unroll 1
            for (int jj = 0; jj < 8; ++jj) {
                const int j = w * 8 + jj;
                if (__builtin_amdgcn_readfirstlane(cnt[j]) > CAP - PUSH_MAX) compact_list(smem, j, k, lane);
            }
            lds_barrier();
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    for (int jj = 0; jj < 8; ++jj) compact_list(smem, w * 8 + jj, k, lane);
    lds_barrier();
    // ---- write this stripe's best k per query (fill: -FLT_MAX / -1)
    for (int e = tid; e < QT * k; e += 256) {
        const int j = e / k, t = e - j * k;
        const int n = reinterpret_cast<const int*>(smem + OFF_CNT)[j];
        const bool ok = j < nq && t < n;
        const int64_t o = ((int64_t)blockIdx.x * QT + j) * k + t;
        part_s[o] = ok ? reinterpret_cast<const float*>(smem + OFF_CS)[j * CAP + t] : -FLT_MAX;
        part_i[o] = ok ? reinterpret_cast<const int*>(smem + OFF_CI)[j * CAP + t] : -1;
    }
}

// ---- merge ---------------------------------------------------------------------------------------
constexpr int MRG_A = 8192, MRG_B = 4224;

struct Ent { float s; int64_t i; };

// One wave: best k of the 64 entries it holds (one per lane; invalid = id < 0), written sorted to dst[0..k).
__device__ __forceinline__ void wave_select(float s, int64_t id, int k, int lane, float* ds, int64_t* di) {
    const bool valid = id >= 0;
    const float ss = valid ? s : -INFINITY;
    const int64_t ii = valid ? id : INT64_MAX;
    int rank = 0;
    const unsigned long long vmask = __builtin_amdgcn_ballot_w64(valid);
    for (int t = 0; t < 64; ++t) {
        if (!((vmask >> t) & 1ull)) continue;  // wave-uniform
        const float st = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ss), t));
        const int lo = __builtin_amdgcn_readlane((int)(ii & 0xffffffffll), t);
        const int hi = __builtin_amdgcn_readlane((int)(ii >> 32), t);
        const int64_t it = ((int64_t)hi << 32) | (unsigned)lo;
        rank += better(st, it, ss, ii) ? 1 : 0;
    }
    const int nvalid = __builtin_popcountll(vmask);
    if (valid && rank < k) {
        ds[rank] = ss;
        di[rank] = ii;
    }
    if (lane < k && lane >= nvalid) {
        ds[lane] = -INFINITY;
        di[lane] = -1;
    }
}

constexpr int MRG_THREADS = 1024;
constexpr int MRG_WAVES = MRG_THREADS / 64;

template <typename IdT>
__global__ __launch_bounds__(MRG_THREADS) void topk_merge_kernel(const float* __restrict__ in_s, const IdT* __restrict__ in_i,
                                                         int n_parts, int nq, int q_stride, int k, int64_t id_base,
                                                         float* __restrict__ out_s, int64_t* __restrict__ out_i) {
    __shared__ float a_s[MRG_A];
    __shared__ int64_t a_i[MRG_A];
    __shared__ float b_s[MRG_B];
    __shared__ int64_t b_i[MRG_B];
    const int q = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int M = n_parts * k;
    // level 0: global -> A.  Entry e = part * k + t lives at in[(part * q_stride + q) * k + t].
    int nch = (M + 63) / 64;
    for (int c = w; c < nch; c += MRG_WAVES) {
        const int e = c * 64 + lane;
        float s = -INFINITY;
        int64_t id = -1;
        if (e < M) {
            const int part = e / k, t = e - part * k;
            const int64_t src = ((int64_t)part * q_stride + q) * k + t;
            const IdT raw = in_i[src];
            s = in_s[src];
            id = raw < 0 ? -1 : (int64_t)raw + id_base;
        }
        wave_select(s, id, k, lane, a_s + c * k, a_i + c * k);
    }
    __syncthreads();
    int n = nch * k;
    bool in_a = true;
    while (nch > 1) {
        const float* ss = in_a ? a_s : b_s;
        const int64_t* si = in_a ? a_i : b_i;
        float* dsd = in_a ? b_s : a_s;
        int64_t* did = in_a ? b_i : a_i;
        const int nch2 = (n + 63) / 64;
        for (int c = w; c < nch2; c += MRG_WAVES) {
            const int e = c * 64 + lane;
            const float s = e < n ? ss[e] : -INFINITY;
            const int64_t id = e < n ? si[e] : -1;
            wave_select(s, id, k, lane, dsd + c * k, did + c * k);
        }
        __syncthreads();
        n = nch2 * k;
        nch = nch2;
        in_a = !in_a;
    }
    const float* fs = in_a ? a_s : b_s;
    const int64_t* fi = in_a ? a_i : b_i;
    if (tid < k) {
        const int64_t id = M > 0 ? fi[tid] : -1;
        out_s[(int64_t)q * k + tid] = id < 0 ? -FLT_MAX : fs[tid];
        out_i[(int64_t)q * k + tid] = id < 0 ? -1 : id;
    }
}

// ---- small similarity ops ---------------------------------------------------------------------------
__global__ void gram_kernel(const float* __restrict__ E, int n, int d, float* __restrict__ sims) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int a = idx / n, b = idx - a * n;
    float s = 0.f;
    for (int j = 0; j < d; ++j) s = __builtin_fmaf(E[(int64_t)a * d + j], E[(int64_t)b * d + j], s);
    sims[idx] = s;
}

__global__ void centroid_kernel(const float* __restrict__ R, int nr, int d, float* __restrict__ centroid) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    float s = 0.f;
    for (int i = 0; i < nr; ++i) s += R[(int64_t)i * d + j];  // row order, as a reduction over axis 0
    centroid[j] = s / (float)nr;
}

__global__ void centroid_sims_kernel(const float* __restrict__ Cand, int nc, int d, const float* __restrict__ centroid,
                                     float* __restrict__ sims) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nc) return;
    float s = 0.f;
    for (int j = 0; j < d; ++j) s = __builtin_fmaf(Cand[(int64_t)c * d + j], centroid[j], s);
    sims[c] = s;
}

struct ScanPlan { int n_wg; int rows_per_wg; };
inline ScanPlan plan_scan(int64_t N) {
    int64_t tiles = (N + TR - 1) / TR;
    int64_t n_wg = tiles < MAX_WG ? tiles : MAX_WG;
    if (n_wg < 1) n_wg = 1;
    int64_t tiles_per = (tiles + n_wg - 1) / n_wg;
    if (tiles_per < 1) tiles_per = 1;
    n_wg = (tiles + tiles_per - 1) / tiles_per;
    if (n_wg < 1) n_wg = 1;
    return {(int)n_wg, (int)(tiles_per * TR)};
}

}  // namespace
}  // namespace sdag

using namespace sdag;

extern "C" int64_t sdag_dense_workspace_bytes(int nq, int64_t N, int d, int k) {
    (void)nq; (void)d;
    if (k <= 0 || k > MAX_K || N < 0) return -1;
    const ScanPlan pl = plan_scan(N);
    return ((int64_t)pl.n_wg * QT * k * 8 + 255) & ~255ll;
}

extern "C" int sdag_dense_topk(const float* Q, int nq, const float* C, int64_t N, int d, int k,
                               float* out_s, int64_t* out_i, int64_t id_base,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    if (nq < 0 || N < 0) return fail(SDAG_ERR_ARG, "negative nq or N");
    if (d <= 0 || d % KC) return fail(SDAG_ERR_ARG, "d=%d must be a positive multiple of %d", d, KC);
    if (k <= 0 || k > MAX_K) return fail(SDAG_ERR_ARG, "k=%d outside [1, %d]", k, MAX_K);
    if (N > INT_MAX) return fail(SDAG_ERR_ARG, "N too large for one shard");
    if (nq == 0) return SDAG_OK;
    if (!Q || !out_s || !out_i || (N > 0 && !C)) return fail(SDAG_ERR_ARG, "NULL pointer");
    if (((uintptr_t)Q | (uintptr_t)C) & 15) return fail(SDAG_ERR_ARG, "Q and C must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const ScanPlan pl = plan_scan(N);
    const int64_t need = sdag_dense_workspace_bytes(nq, N, d, k);
    if (!workspace || workspace_bytes < need) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)need);
    float* part_s = (float*)workspace;
    int* part_i = (int*)(part_s + (int64_t)pl.n_wg * QT * k);
    if ((( (int64_t)pl.n_wg * k + 63) / 64) * k > MRG_A) return fail(SDAG_ERR_ARG, "too many partial lists for the merge");
    for (int q0 = 0; q0 < nq; q0 += QT) {
        const int nqt = nq - q0 < QT ? nq - q0 : QT;
        if (N > 0) {
            SDAG_HIP_CHECK(hipFuncSetAttribute((const void*)dense_scan, hipFuncAttributeMaxDynamicSharedMemorySize, SCAN_LDS));
            hipLaunchKernelGGL(dense_scan, dim3(pl.n_wg), dim3(256), SCAN_LDS, st, Q + (int64_t)q0 * d, nqt, C, N, d, k,
                               pl.rows_per_wg, part_s, part_i);
            SDAG_HIP_CHECK(hipGetLastError());
        }
        hipLaunchKernelGGL(topk_merge_kernel<int>, dim3(nqt), dim3(MRG_THREADS), 0, st, part_s, part_i, N > 0 ? pl.n_wg : 0, nqt, QT,
                           k, id_base, out_s + (int64_t)q0 * k, out_i + (int64_t)q0 * k);
        SDAG_HIP_CHECK(hipGetLastError());
    }
    return SDAG_OK;
}

extern "C" int sdag_topk_merge(const float* in_s, const int64_t* in_i, int n_parts, int nq, int k,
                               float* out_s, int64_t* out_i, void* stream) {
    if (n_parts < 0 || nq < 0) return fail(SDAG_ERR_ARG, "negative n_parts or nq");
    if (k <= 0 || k > MAX_K) return fail(SDAG_ERR_ARG, "k=%d outside [1, %d]", k, MAX_K);
    if (nq == 0) return SDAG_OK;
    if (!out_s || !out_i || (n_parts > 0 && (!in_s || !in_i))) return fail(SDAG_ERR_ARG, "NULL pointer");
    if ((((int64_t)n_parts * k + 63) / 64) * k > MRG_A) return fail(SDAG_ERR_ARG, "too many partial lists");
    hipLaunchKernelGGL(topk_merge_kernel<int64_t>, dim3(nq), dim3(MRG_THREADS), 0, (hipStream_t)stream, in_s, in_i, n_parts, nq, nq, k,
                       (int64_t)0, out_s, out_i);
    SDAG_HIP_CHECK(hipGetLastError());
    return SDAG_OK;
}

extern "C" int sdag_gram(const float* E, int n, int d, float* sims, void* stream) {
    if (n < 0 || d <= 0) return fail(SDAG_ERR_ARG, "bad n or d");
    if (n == 0) return SDAG_OK;
    if (!E || !sims) return fail(SDAG_ERR_ARG, "NULL pointer");
    hipLaunchKernelGGL(gram_kernel, dim3((n * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, E, n, d, sims);
    SDAG_HIP_CHECK(hipGetLastError());
    return SDAG_OK;
}

extern "C" int sdag_centroid_sims(const float* R, int nr, const float* Cand, int nc, int d, float* centroid,
                                  float* sims, void* stream) {
    if (nr <= 0 || nc < 0 || d <= 0) return fail(SDAG_ERR_ARG, "bad nr, nc or d");
    if (!R || !centroid || (nc > 0 && (!Cand || !sims))) return fail(SDAG_ERR_ARG, "NULL pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(centroid_kernel, dim3((d + 255) / 256), dim3(256), 0, st, R, nr, d, centroid);
    SDAG_HIP_CHECK(hipGetLastError());
    if (nc > 0) {
        hipLaunchKernelGGL(centroid_sims_kernel, dim3((nc + 63) / 64), dim3(64), 0, st, Cand, nc, d, centroid, sims);
        SDAG_HIP_CHECK(hipGetLastError());
    }
    return SDAG_OK;
}
