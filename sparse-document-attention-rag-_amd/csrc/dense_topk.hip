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
// Scan kernel: every workgroup owns a contiguous stripe of corpus rows and walks it in tiles of 128 rows against up
// to 32 NQB queries per pass over the corpus (NQB = 1, 2 or 4, chosen from the number of queries: a staged corpus
// tile is multiplied against NQB blocks of 32 queries, so 128 queries cost one read of the corpus, not four).  With
// 4 waves (NQB = 1) wave w multiplies rows 32 w .. 32 w + 31 of the tile against the one query block; with 8 waves
// waves w and w + 4 share those rows and split the query blocks.  Corpus and query
// slices of 32 dimensions are brought into a 3-stage LDS ring by LDS-DMA (global_load_lds_dwordx4, one
// 1 KiB piece per wave instruction, 128-byte source lines, XOR-swizzled through the SOURCE address so that
// the later ds_read_b128 fragment reads are bank-conflict free).  Loads run two stages ahead of the MFMAs
// behind a counted s_waitcnt vmcnt + raw s_barrier, so HBM latency stays hidden while the matrix pipe runs
// the 64-cycle fp32 MFMAs back to back.  After the last slice of a tile the 32 x 32 scores of each wave are
// filtered against a per-query threshold (the current k-th best of the stripe) and survivors are appended to
// per-query candidate lists in LDS, which are compacted to their best k whenever they could overflow.
// Each workgroup finally writes its best k per query; a merge kernel reduces those lists.
//
// The same kernel template has a bf16 form (a bf16 copy of the corpus, one v_mfma_f32_32x32x16_bf16 per 16
// dimensions) that only picks candidates for exact rescoring (sdag_dense_topk2: same results bit for bit, half the
// bytes), and for more than 32 queries a threshold form (up to 256 queries per pass, two corpus tiles per staged query
// slab, no lists: every row at or above a per-query threshold taken from a pilot over a sample goes to that query's
// candidate list in global memory); see dense_scan.  sdag_dense_topk_after is the exact scan behind a cursor (paging
// for k above 32).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <type_traits>
#include <climits>

#include "common.h"

namespace sdag {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define SDAG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define SDAG_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

constexpr int QT = 32;                      // queries per query block (one MFMA tile wide)
constexpr int TR = 128;                     // corpus rows per tile
constexpr int KC = 32;                      // fp32 dimensions per stage (128 bytes of a row)
constexpr int KC16 = 64;                    // bf16 dimensions per stage (the same 128 bytes)
#ifndef SDAG_SCAN_NST
#define SDAG_SCAN_NST 3
#define SDAG_SCAN_WG_PER_CU 2
#endif
constexpr int NST = SDAG_SCAN_NST;          // LDS ring depth: loads run NST - 1 stages ahead of the MFMAs
constexpr int C_STAGE = TR * KC * 4;        // 16 KiB
constexpr int Q_STAGE = QT * KC * 4;        // 4 KiB per query block
constexpr int PUSH_MAX = 32;                // most entries one filter round can append to one list
constexpr int MAX_K = 32;
constexpr int MAX_WG = 512;                 // stripe lists the merge and rescoring kernels can take
constexpr int SCAN_WG = 256 * SDAG_SCAN_WG_PER_CU;  // workgroups of the narrow scan: what the chip holds at once
#ifndef SDAG_SCAN_EXP
#define SDAG_SCAN_EXP 0   // timing-only builds: 1 no filtering (the MFMAs go with it: nothing reads the sums), 2 no MFMAs,
                          // 4 no query pieces, 8 no corpus pieces, 16 no row passes the thresholds, 32 MFMA operands not read
#endif
constexpr int MAX_NQB = 4;                  // query blocks of the widest scan
constexpr int WIDE_KS = 16;                 // longest stripe list the widest scan keeps (its lists are shorter: LDS)
#ifndef SDAG_SCAN_NSTC
#define SDAG_SCAN_NSTC 4
#endif
#ifndef SDAG_SCAN_QDMA
#define SDAG_SCAN_QDMA 1   // 1: the wide scans bring their query slabs by LDS-DMA as well (waves 4-7), 0: through registers
#endif
#define SDAG_SCAN_NSTQ 2
#ifndef SDAG_SCAN_QMID
#define SDAG_SCAN_QMID 2   // the MFMA group (of four) of a pair of stages in front of which waves 4-7 issue the next slab (0: at the barrier)
#endif
static_assert(SCAN_WG <= MAX_WG, "the rescoring kernel holds one maximum per stripe in MAX_WG / 64 registers");
static_assert(MAX_WG % 64 == 0, "stripe maxima are held 64 per register");

// Geometry of one scan variant: NQB query blocks, NW waves.
//   NW = 4: every wave issues corpus and query pieces of a stage into ONE ring of NST stages (two workgroups per
//           compute unit keep 4 x 16 KiB of corpus in flight there).
//   NW = 8: one workgroup per compute unit, so the corpus ring is deeper (NSTC stages); the query slabs, which come
//           from the L2, sit in a shallow ring of their own (NSTQ).  Waves 0-3 issue only corpus pieces and waves 4-7
//           only query pieces: the s_waitcnt counter is per wave and in issue order, so a wave that issued both could
//           not wait for a recent query piece without also waiting for every older corpus piece.
//   THR: the threshold scan (see dense_scan): no per-query lists, one buffer of THR_CB survivors per workgroup instead,
//           so the LDS takes the rings of up to 8 query blocks (256 queries per pass over the corpus).
constexpr int THR_CB = 1024;                // survivors a workgroup buffers before it appends them to the global lists
template <int NQB, int NW, bool THR = false>
struct ScanCfg {
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(NQB % (NW / 4) == 0, "the query blocks are split over the wave groups");
    static constexpr bool SPLIT = NW == 8;               // separate rings and loader roles
    static constexpr int QTW = QT * NQB;                 // queries per pass
    static constexpr int QPW = NQB / (NW / 4);           // query blocks per wave
    static constexpr int NSTC = SPLIT ? SDAG_SCAN_NSTC : NST;
    static constexpr int NSTQ = SPLIT ? SDAG_SCAN_NSTQ : NST;
    static constexpr int Q_SLAB = NQB * Q_STAGE;
    static constexpr int C_STRIDE = SPLIT ? C_STAGE : C_STAGE + Q_SLAB;   // distance of consecutive corpus stages
    static constexpr int Q_STRIDE = SPLIT ? Q_SLAB : C_STAGE + Q_SLAB;
    static constexpr int Q_BASE = SPLIT ? NSTC * C_STAGE : C_STAGE;
    static constexpr int RING = SPLIT ? NSTC * C_STAGE + NSTQ * Q_SLAB : NST * (C_STAGE + Q_SLAB);
    static constexpr int PPW_C = SPLIT ? 4 : 4;          // corpus pieces (1 KiB) a loading wave issues per stage
    static constexpr int PPW_Q = NQB;                    // query pieces a loading wave issues per stage
    // vector-memory operations a wave may leave in flight when it needs stage s: those of the later stages
    static constexpr int WAIT_C = SPLIT ? PPW_C * (NSTC - 2) : (PPW_C + PPW_Q) * (NST - 2);
    // candidate list capacity per query: 64, or what the LDS leaves (a list is compacted when it could overflow in the
    // next filter round, so a short list is compacted after every few survivors)
    static constexpr int CAP_ROOM = THR ? 64 : (160 * 1024 - RING - 3 * QTW * 4) / (QTW * 8);
    static constexpr int CAP = CAP_ROOM < 64 ? CAP_ROOM : 64;
    static_assert(THR || CAP >= (NQB == MAX_NQB ? WIDE_KS : MAX_K) + PUSH_MAX, "a list holds what is kept plus one round of survivors");
    static constexpr int LPW = QTW / NW;                 // lists a wave looks after
    static constexpr int OFF_CS = RING;                  // float cand_s[QTW][CAP]
    static constexpr int OFF_CI = OFF_CS + QTW * CAP * 4;  // int   cand_i[QTW][CAP]
    static constexpr int OFF_CNT = OFF_CI + QTW * CAP * 4; // int   cnt[QTW]
    static constexpr int OFF_TS = OFF_CNT + QTW * 4;     // float thr_s[QTW]
    static constexpr int OFF_TI = OFF_TS + QTW * 4;      // int   thr_i[QTW]
    // threshold scan: int buf_q[THR_CB], float buf_s[THR_CB], int buf_i[THR_CB], int buf_n
    static constexpr int OFF_BQ = RING;
    static constexpr int OFF_BS = OFF_BQ + THR_CB * 4;
    static constexpr int OFF_BI = OFF_BS + THR_CB * 4;
    static constexpr int OFF_BN = OFF_BI + THR_CB * 4;
    static constexpr int LDS = THR ? OFF_BN + 16 : OFF_TI + QTW * 4;
    static constexpr int WG_PER_CU = SPLIT ? 1 : SDAG_SCAN_WG_PER_CU;
    static_assert(LDS <= 160 * 1024, "LDS of a compute unit");
    static_assert(THR || LPW <= 64, "a wave checks its lists' lengths one per lane");
    static_assert(!THR || SPLIT, "the threshold scan runs 8 waves");
};

// What the threshold scan reads and writes beside the corpus and the queries (all for the pass's queries).
struct ThrArgs {
    const float* pilot_s;  // [nq][m]: the m best approximate scores of a sample of the corpus, best first
    const int64_t* pilot_i;  // [nq][m]: their rows (-1 = the sample held fewer rows)
    int m;
    const float* eps;      // [nq]: an approximate score is within eps of the exact one
    float* cand_s;         // [nq][cap]: approximate scores of the rows at or above the query's threshold
    int* cand_i;           // [nq][cap]: those rows
    int* cand_n;           // [nq]: how many there are (may exceed cap: the rest was dropped), zero when the pass starts
    int cap;
    int* flag;             // set when a workgroup had to drop survivors
    // block maxima (MODE 2): entry e = 4 * sampled tile + row group of a query sits at ((e / bm_m) * bm_qtw + query) * bm_m
    // + e % bm_m, the layout the merge kernel reads as lists of bm_m entries; bm_i takes e (prefilled with -1)
    float* bm_s;
    int* bm_i;
    int bm_m, bm_qtw;
    // exact list scan only (the other fields unused): a cursor per query.  Only rows that come strictly after
    // (after_s[q], after_i[q] - after_base) in the order of the results are kept: the next page of a search whose
    // previous page ended at that entry.  Null: no cursor.
    const float* after_s;
    const int64_t* after_i;
    int64_t after_base;
};

// The threshold of a query in the threshold scan: the sample's m-th best approximate score less 2 eps (every row whose
// approximate score reaches the line is a candidate); without m sample rows every row is one.
__device__ __forceinline__ float thr_of(const ThrArgs& a, int q) {
    return a.pilot_i[(int64_t)q * a.m + a.m - 1] >= 0 ? a.pilot_s[(int64_t)q * a.m + a.m - 1] - 2.0f * a.eps[q] : -FLT_MAX;
}
constexpr int MODE_LISTS = 0, MODE_THR = 1, MODE_BMAX = 2;

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

// Order-preserving map of a float's bits to an unsigned integer (larger float = larger key; -inf lowest of all values
// that occur here), and back.
__device__ __forceinline__ unsigned float_key(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// One wave reduces list j to its best min(n, k) entries, sorted best first, and refreshes the threshold.
template <class CFG>
__device__ __forceinline__ void compact_list(unsigned char* smem, int j, int k, int lane) {
    constexpr int CAP = CFG::CAP, OFF_CS = CFG::OFF_CS, OFF_CI = CFG::OFF_CI, OFF_CNT = CFG::OFF_CNT, OFF_TS = CFG::OFF_TS,
                  OFF_TI = CFG::OFF_TI;
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

// BF16 = false: the exact scan described above.  BF16 = true: the same walk over bf16 copies of corpus and queries
// (64 dimensions per 128-byte stage, one v_mfma_f32_32x32x16_bf16 per 16 dimensions): approximate scores, used only
// to pick candidates for exact rescoring.  run_flag (optional): the kernel does nothing when *run_flag == 0.
// Partial lists: part[(stripe * QTW + query) * k + t].
//
// tile_step > 1: the scan walks every tile_step-th 128-row tile of the corpus only (a sample: rows_per_wg counts sampled
// rows).
// THR = true (the threshold scan, for many queries at once): instead of keeping the best k per stripe and query, the
// scan appends every row whose approximate score reaches the query's threshold (ThrArgs) to that query's global
// candidate list, through a workgroup-wide buffer in LDS; k, part_s and part_i are not used.  No lists, no compaction
// and, tile by tile, no barrier: with thresholds taken from a sample so that a few hundred rows of the whole corpus
// pass, a workgroup sees a handful of survivors per tile.
// MODE_BMAX (the pilot of the threshold scan, over a sample of the tiles): per query and 32-row block of every tile the
// largest approximate score goes to ta.bm_s; the m-th largest of those (scores of m distinct rows) is the sample
// statistic the thresholds are taken from.  No lists, no atomics.
template <bool BF16, int NQB, int NW, int MODE>
__global__ __launch_bounds__(64 * NW, (ScanCfg<NQB, NW, MODE != 0>::WG_PER_CU)) void dense_scan(const void* __restrict__ Qv, int nq, const void* __restrict__ Cv,
                                                     int64_t N, int d, int k, int rows_per_wg,
                                                     float* __restrict__ part_s, int* __restrict__ part_i,
                                                     const int* __restrict__ run_flag, int q_row_bytes, int q_kc_bytes,
                                                     int tile_step, const ThrArgs ta) {
    // Query j's 128 bytes of stage kc sit at Q + j * q_row_bytes + kc * q_kc_bytes (row-major queries: d * element size
    // and 128; slab-major, see query_prepare_kernel: 128 and 128 * queries per pass).
    if (run_flag && *run_flag == 0) return;
    constexpr bool THR = MODE == MODE_THR;
    using CFG = ScanCfg<NQB, NW, MODE != 0>;
    constexpr int QTW = CFG::QTW, QPW = CFG::QPW, CAP = CFG::CAP, LPW = CFG::LPW;
    constexpr int OFF_CS = CFG::OFF_CS, OFF_CI = CFG::OFF_CI, OFF_CNT = CFG::OFF_CNT, OFF_TS = CFG::OFF_TS, OFF_TI = CFG::OFF_TI;
    constexpr int KCE = BF16 ? KC16 : KC;  // elements per stage
    constexpr int ES = BF16 ? 2 : 4;       // bytes per element
    const unsigned char* Q = reinterpret_cast<const unsigned char*>(Qv);
    const unsigned char* C = reinterpret_cast<const unsigned char*>(Cv);
    // Dynamic LDS on purpose: the ring is filled only by LDS-DMA, which the optimiser does not see as a store, so
    // a static array would look never-written and its loads would be folded away.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    __builtin_assume(w >= 0 && w < NW);
    const int rg = w & 3, qg = w >> 2;  // the wave's 32 rows of a tile, its group of query blocks
    // this workgroup's tiles: sampled tile t of the stripe is corpus tile (tile0 + t) * tile_step
    const int64_t tile0 = (int64_t)blockIdx.x * (rows_per_wg / TR);
    const int64_t n_sampled = ((N + TR - 1) / TR + tile_step - 1) / tile_step;
    const int ntile = (int)max((int64_t)0, min((int64_t)(rows_per_wg / TR), n_sampled - tile0));
    const int nkc = d / KCE;
    // The threshold scan multiplies TWO consecutive corpus tiles against every query slab it stages (accumulators for
    // both): the slab's pieces are then a third of what a pair of tiles brings into the LDS instead of two thirds of
    // what one does, and the scan's time follows the pieces.  Corpus stage s is tile TPS * (s / (TPS nkc)) + s % TPS,
    // dimensions block (s % (TPS nkc)) / TPS; query stage sq = s / TPS.  (An odd last tile has a phantom partner:
    // loaded and multiplied like any other, never filtered.)
    constexpr int TPS = MODE == MODE_THR ? 2 : 1;
    const int S = ((ntile + TPS - 1) / TPS) * TPS * nkc;
    const int SQ = S / TPS;
    auto tile_row0 = [&](int tile) { return (tile0 + tile) * tile_step * TR; };

    if constexpr (MODE != MODE_LISTS) {
        if (tid == 0) *reinterpret_cast<int*>(smem + CFG::OFF_BN) = 0;
    } else {
        for (int j = tid; j < QTW; j += 64 * NW) {
            reinterpret_cast<int*>(smem + OFF_CNT)[j] = 0;
            reinterpret_cast<float*>(smem + OFF_TS)[j] = -INFINITY;
            reinterpret_cast<int*>(smem + OFF_TI)[j] = INT_MAX;
        }
    }
    lds_barrier();
    if (S == 0) {  // (a stripe beyond the sample: nothing to scan, fill lists to write)
        if constexpr (MODE == MODE_LISTS)
            for (int e = tid; e < QTW * k; e += 64 * NW) {
                part_s[(int64_t)blockIdx.x * QTW * k + e] = -FLT_MAX;
                part_i[(int64_t)blockIdx.x * QTW * k + e] = -1;
            }
        return;
    }

    // ---- LDS-DMA geometry.  A wave instruction writes 1 KiB = 8 rows x 128 B; lane -> (row lane >> 3,
    // physical chunk lane & 7); the source chunk is physical ^ ((row >> 1) & 7).  A corpus stage is 16 pieces (rows
    // 8 c .. 8 c + 7 of the tile), a query slab 4 NQB pieces (queries 8 c ..).
    const int dr = lane >> 3, dc = lane & 7;
    auto issue_c = [&](int s, int c) {  // corpus piece c of stage s
        if (SDAG_SCAN_EXP & 8) return;
        s = s < S ? s : S - 1;  // past the end: reload the last stage into the free slot (keeps vmcnt counts fixed)
        const int grp = s / (TPS * nkc), within = s - grp * (TPS * nkc);
        const int tile = grp * TPS + within % TPS, kc = within / TPS;
        unsigned char* sb = smem + (s % CFG::NSTC) * CFG::C_STRIDE;
        const int rl = c * 8 + dr;  // row inside the tile (the swizzle only looks at its low bits)
        int64_t row = tile_row0(tile) + rl;
        row = row < N ? row : N - 1;
        const int csrc = dc ^ ((rl >> 1) & 7);
        const unsigned char* src = C + (row * d + kc * KCE) * ES + csrc * 16;
        __builtin_amdgcn_global_load_lds(SDAG_GLB_PTR(src), SDAG_LDS_PTR(sb + c * 1024), 16, 0, 0);
    };
    auto issue_q = [&](int s, int c) {  // query piece c of query stage s
        if (SDAG_SCAN_EXP & 4) return;
        s = s < SQ ? s : SQ - 1;
        const int kc = s % nkc;
        unsigned char* sb = smem + CFG::Q_BASE + (s % CFG::NSTQ) * CFG::Q_STRIDE;
        int j = c * 8 + dr;
        const int csrc = dc ^ ((j >> 1) & 7);
        j = j < nq ? j : nq - 1;
        const unsigned char* src = Q + (int64_t)j * q_row_bytes + (int64_t)kc * q_kc_bytes + csrc * 16;
        __builtin_amdgcn_global_load_lds(SDAG_GLB_PTR(src), SDAG_LDS_PTR(sb + c * 1024), 16, 0, 0);
    };
    // what this wave issues for the stage that is `ahead` stages in front of stage s
    auto issue_corpus = [&](int s) {
#pragma unroll
        for (int n = 0; n < CFG::PPW_C; ++n) issue_c(s, (w & 3) * 4 + n);
    };
    auto issue_queries = [&](int s) {
#pragma unroll
        for (int n = 0; n < CFG::PPW_Q; ++n) issue_q(s, (w & 3) * NQB + n);
    };
    // The same pieces through registers (waves 4-7 of the wide scans): requested when a stage starts, written to the
    // slab of the next stage when it ends.  The fill rate of LDS-DMA is about one piece per 100 cycles and compute
    // unit whatever the source, so query pieces sent that way would halve the corpus rate.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    u32x4 qreg[CFG::PPW_Q];
    auto load_queries = [&](int s) {
        if (SDAG_SCAN_EXP & 4) return;
        s = s < SQ ? s : SQ - 1;
        const int kc = s % nkc;
#pragma unroll
        for (int n = 0; n < CFG::PPW_Q; ++n) {
            int j = ((w & 3) * NQB + n) * 8 + dr;
            const int csrc = dc ^ ((j >> 1) & 7);
            j = j < nq ? j : nq - 1;
            qreg[n] = *reinterpret_cast<const u32x4*>(Q + (int64_t)j * q_row_bytes + (int64_t)kc * q_kc_bytes + csrc * 16);
        }
    };
    auto store_queries = [&](int s) {
        unsigned char* sb = smem + CFG::Q_BASE + (s % CFG::NSTQ) * CFG::Q_STRIDE;
#pragma unroll
        for (int n = 0; n < CFG::PPW_Q; ++n) *reinterpret_cast<u32x4*>(sb + ((w & 3) * NQB + n) * 1024 + lane * 16) = qreg[n];
    };

    const int r = lane & 31, h = lane >> 5;
    const int swz = (r >> 1) & 7;
    const int a_off = (rg * 32 + r) * 128;
    const int b_off = (qg * QPW * 32 + r) * 128;

    f32x16 accs[TPS][QPW];
    f32x16 (&acc)[QPW] = accs[0];  // (the scans of one tile per slab)
    float thr[QPW];  // threshold scan: this lane's queries' thresholds
#pragma unroll
    for (int qb = 0; qb < QPW; ++qb) {
#pragma unroll
        for (int hh = 0; hh < TPS; ++hh)
#pragma unroll
            for (int i = 0; i < 16; ++i) accs[hh][qb][i] = 0.f;
        thr[qb] = 0.f;
        if constexpr (THR) {
            const int qi = (qg * QPW + qb) * 32 + r;
            thr[qb] = qi < nq && !(SDAG_SCAN_EXP & 16) ? thr_of(ta, qi) : INFINITY;
        }
    }
    // Threshold scan: the workgroup's buffered survivors go to the queries' global lists (every thread takes some; the
    // waits for the atomics' results drain this wave's loads in flight, which is why it happens rarely).
    auto flush_buffer = [&]() {
        lds_barrier();
        const int n_raw = *reinterpret_cast<const int*>(smem + CFG::OFF_BN);
        if (n_raw > THR_CB && tid == 0) atomicOr(ta.flag, 1);  // survivors were dropped: the exact scan has to answer
        const int n = n_raw < THR_CB ? n_raw : THR_CB;
        for (int e = tid; e < n; e += 64 * NW) {
            const int q = reinterpret_cast<const int*>(smem + CFG::OFF_BQ)[e];
            const int pos = atomicAdd(ta.cand_n + q, 1);
            if (pos < ta.cap) {
                ta.cand_s[(int64_t)q * ta.cap + pos] = reinterpret_cast<const float*>(smem + CFG::OFF_BS)[e];
                ta.cand_i[(int64_t)q * ta.cap + pos] = reinterpret_cast<const int*>(smem + CFG::OFF_BI)[e];
            }
        }
        lds_barrier();
        if (tid == 0) *reinterpret_cast<int*>(smem + CFG::OFF_BN) = 0;
        lds_barrier();
    };

    // corpus stages a loading wave runs ahead of the MFMAs: all but one slot of the ring; with two tiles per slab the
    // ring holds the pair that is computed and the pair that is loaded
    constexpr int LOOK_C = TPS == 2 ? 2 : CFG::NSTC - 1;
    static_assert(TPS == 1 || CFG::NSTC == 4, "two tiles per slab: a ring of two pairs");
    if constexpr (CFG::SPLIT) {
        if (w < 4) {
#pragma unroll
            for (int s0 = 0; s0 < LOOK_C; ++s0) issue_corpus(s0);
        } else if (SDAG_SCAN_QDMA) {
#pragma unroll
            for (int s0 = 0; s0 < CFG::NSTQ - 1; ++s0) issue_queries(s0);
        } else {
            static_assert(!CFG::SPLIT || CFG::NSTQ == 2, "query slabs through registers: one stage ahead");
            load_queries(0);
            store_queries(0);
        }
    } else {
#pragma unroll
        for (int s0 = 0; s0 < NST - 1; ++s0) {
            issue_corpus(s0);
            issue_queries(s0);
        }
    }
    // ---- filtering a finished tile's scores into the per-query lists.  Register i of a tile's accumulator is corpus
    // row 32 rg + (i & 3) + 8 (i >> 2) + 4 h of the tile, column r the query (of the block).  A round appends the
    // survivors of four registers per query block (at most PUSH_MAX per list; one atomic per lane and block for all
    // of its survivors); between rounds, behind a barrier, every wave compacts those of its lists that the next round
    // could overflow.  (Riding the rounds on the barriers of the next tile's stages instead was measured slower.)
    float* cs = reinterpret_cast<float*>(smem + OFF_CS);
    int* ci = reinterpret_cast<int*>(smem + OFF_CI);
    int* cnt = reinterpret_cast<int*>(smem + OFF_CNT);
    // the cursors of this lane's queries (exact scan): every row is after (+inf, -1)
    constexpr bool CURSOR = !BF16 && MODE == MODE_LISTS;
    float cur_s[QPW];
    int64_t cur_i[QPW];
    if constexpr (CURSOR) {
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) {
            const int qi = (qg * QPW + qb) * 32 + r;
            const bool has = ta.after_s != nullptr && qi < nq;
            cur_s[qb] = has ? ta.after_s[qi] : INFINITY;
            cur_i[qb] = has ? ta.after_i[qi] - ta.after_base : -1;
        }
    }
    auto push_round = [&](int round_base, const int64_t row_base) {
#pragma unroll
        for (int qb = 0; qb < QPW; ++qb) {
            const int qi = (qg * QPW + qb) * 32 + r;  // the query's list
            const float ts = reinterpret_cast<const float*>(smem + OFF_TS)[qi];
            const int ti = reinterpret_cast<const int*>(smem + OFF_TI)[qi];
            bool keep[4];
            int n_keep = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = round_base + e;
                const float sc = acc[qb][i];
                const int64_t row = row_base + (i & 3) + 8 * (i >> 2);
                keep[e] = qi < nq && row < N && sc == sc && better(sc, (int)row, ts, ti);
                if constexpr (CURSOR) keep[e] = keep[e] && better(cur_s[qb], cur_i[qb], sc, row);
                n_keep += keep[e] ? 1 : 0;
            }
            if (n_keep) {
                int pos = atomicAdd(&cnt[qi], n_keep);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = round_base + e;
                    if (keep[e]) {
                        cs[qi * CAP + pos] = acc[qb][i];
                        ci[qi * CAP + pos] = (int)(row_base + (i & 3) + 8 * (i >> 2));
                        ++pos;
                    }
                }
            }
        }
    };
    auto compact_check = [&]() {  // the lengths of this wave's lists, one per lane
        const int len = lane < LPW ? cnt[w * LPW + lane] : 0;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(len > CAP - PUSH_MAX);
        while (todo) {
            const int jj = __builtin_ctzll(todo);
            todo &= todo - 1;
            compact_list<CFG>(smem, w * LPW + jj, k, lane);
        }
    };

    int tile = 0, kc = 0;
    for (int s = 0; s < S; s += TPS) {
        // stage s (TPS = 2: stages s and s + 1, and with them everything this wave has in flight) landed for this wave
        // (the pieces of the newer stages may still fly), then for every wave
        if constexpr (CFG::SPLIT) {
            if (w < 4) {
                // (the next pieces right behind the barrier: spread between the MFMAs below they land too late for
                // the next stage's wait -- measured 20 % slower)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(TPS == 2 ? 0 : CFG::WAIT_C) : "memory");
#pragma unroll
                for (int hh = 0; hh < TPS; ++hh) issue_corpus(s + hh + LOOK_C);
            } else if (SDAG_SCAN_QDMA) {
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(CFG::PPW_Q * (CFG::NSTQ - 2)) : "memory");
                if (TPS == 1 || !(SDAG_SCAN_QMID && NQB == 8)) issue_queries(s / TPS + CFG::NSTQ - 1);
            } else {
                static_assert(SDAG_SCAN_QDMA || TPS == 1, "query slabs through registers: one tile per slab");
                lds_barrier();  // (this wave's slab of stage s was written at the end of stage s - 1)
                load_queries(s + 1);
            }
        } else {
            static_assert(CFG::SPLIT || TPS == 1, "two tiles per slab: the scans with separate rings");
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(CFG::WAIT_C) : "memory");
            issue_corpus(s + NST - 1);
            issue_queries(s + NST - 1);
        }
        if constexpr (THR) {
            // between a stage's barrier and the tiles' end nobody appends: every wave reads the same count here
            if (kc == 0 && __builtin_amdgcn_readfirstlane(*reinterpret_cast<const int*>(smem + CFG::OFF_BN)) > THR_CB / 2)
                flush_buffer();
        }
        const unsigned char* sbc[TPS];
#pragma unroll
        for (int hh = 0; hh < TPS; ++hh) sbc[hh] = smem + ((s + hh) % CFG::NSTC) * CFG::C_STRIDE;
        const unsigned char* sbq = smem + CFG::Q_BASE + ((s / TPS) % CFG::NSTQ) * CFG::Q_STRIDE;
#pragma unroll
        for (int m = 0; m < ((SDAG_SCAN_EXP & 2) ? 0 : 4); ++m) {
            if constexpr (TPS == 2 && SDAG_SCAN_QMID && NQB == 8 && CFG::SPLIT) {
                // The waves that bring the query slabs issue their pieces in the middle of their MFMAs: right behind
                // the barrier the SIMD's other wave is issuing corpus pieces, and the matrix pipe would idle.
                if (m == SDAG_SCAN_QMID) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (w >= 4) issue_queries(s / TPS + CFG::NSTQ - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const int ch = ((2 * m + h) ^ swz) << 4;
            if constexpr (BF16) {
                // 16-byte chunk 2 m + h of a row = dimensions 16 m + 8 h .. + 7 of the stage: the MFMA operand as it is;
                // every query operand read from the LDS serves the TPS corpus tiles of the stage
                bf16x8 a[TPS];
#pragma unroll
                for (int hh = 0; hh < TPS; ++hh) a[hh] = *reinterpret_cast<const bf16x8*>((SDAG_SCAN_EXP & 32) ? smem + a_off : sbc[hh] + a_off + ch);
#pragma unroll
                for (int qb = 0; qb < QPW; ++qb) {
                    bf16x8 b;
                    if (SDAG_SCAN_EXP & 32) b = a[0]; else b = *reinterpret_cast<const bf16x8*>(sbq + b_off + qb * Q_STAGE + ch);
#pragma unroll
                    for (int hh = 0; hh < TPS; ++hh) accs[hh][qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[hh], b, accs[hh][qb], 0, 0, 0);
                }
            } else {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sbc[0] + a_off + ch);
                // Lane half 0 holds dimensions 8m..8m+3, half 1 holds 8m+4..8m+7.  Swapping the upper half of
                // register 0 with the lower half of register 1 (and 2 with 3) leaves register 0 = (8m, 8m+1),
                // register 2 = (8m+2, 8m+3), register 1 = (8m+4, 8m+5), register 3 = (8m+6, 8m+7) across the
                // two halves: the MFMA operand layout, in dimension order.
                float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
                swap_halves(a0, a1);
                swap_halves(a2, a3);
#pragma unroll
                for (int qb = 0; qb < QPW; ++qb) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sbq + b_off + qb * Q_STAGE + ch);
                    float b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
                    swap_halves(b0, b1);
                    swap_halves(b2, b3);
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[qb], 0, 0, 0);
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[qb], 0, 0, 0);
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[qb], 0, 0, 0);
                    acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[qb], 0, 0, 0);
                }
            }
        }
        if constexpr (CFG::SPLIT && !SDAG_SCAN_QDMA)
            if (w >= 4) store_queries(s + 1);  // slab of stage s - 1: every wave passed this stage's barrier after reading it
        if (++kc < nkc) continue;
        kc = 0;
        // ---- tile(s) finished
        const int64_t row_base = tile_row0(tile) + rg * 32 + 4 * h;
        if constexpr (MODE == MODE_BMAX) {
#pragma unroll
            for (int qb = 0; qb < QPW; ++qb) {
                float mx = acc[qb][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) mx = fmaxf(mx, acc[qb][i]);
                float other = mx;
                swap_halves(mx, other);  // mx: own half | other half's value -> both halves end with the block's maximum
                mx = fmaxf(mx, other);
                const int qi = (qg * QPW + qb) * 32 + r;
                if (h == 0 && qi < nq) {
                    const int64_t e = (tile0 + tile) * 4 + rg;
                    const int64_t at = ((e / ta.bm_m) * ta.bm_qtw + qi) * ta.bm_m + e % ta.bm_m;
                    ta.bm_s[at] = mx;
                    ta.bm_i[at] = (int)e;
                }
            }
        } else if constexpr (THR) {
            if (!(SDAG_SCAN_EXP & 1)) {
#pragma unroll
                for (int hh = 0; hh < TPS; ++hh) {
                    if (tile + hh >= ntile) break;  // the phantom partner of an odd last tile
                    const int64_t rb = tile_row0(tile + hh) + rg * 32 + 4 * h;
                    // Survivors are rare (about one per wave and pair of tiles): each block of 32 rows x 32 queries is
                    // tested on its own, and only a block that holds one walks its sixteen registers lane by lane.
                    // Rows past the corpus' end exist in its last tile only (their scores repeat the last row's).
                    const bool whole = tile_row0(tile + hh) + TR <= N;
                    int* bn = reinterpret_cast<int*>(smem + CFG::OFF_BN);
#pragma unroll
                    for (int qb = 0; qb < QPW; ++qb) {
                        bool mine = false;
#pragma unroll
                        for (int i = 0; i < 16; ++i) mine |= accs[hh][qb][i] >= thr[qb];
                        if (__builtin_amdgcn_ballot_w64(mine) == 0) continue;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int64_t row = rb + (i & 3) + 8 * (i >> 2);
                            if (accs[hh][qb][i] >= thr[qb] && (whole || row < N)) {  // (a score that is not a number fails the test)
                                const int pos = atomicAdd(bn, 1);
                                if (pos < THR_CB) {
                                    reinterpret_cast<int*>(smem + CFG::OFF_BQ)[pos] = (qg * QPW + qb) * 32 + r;
                                    reinterpret_cast<float*>(smem + CFG::OFF_BS)[pos] = accs[hh][qb][i];
                                    reinterpret_cast<int*>(smem + CFG::OFF_BI)[pos] = (int)row;
                                }
                            }
                        }
                    }
                }
            }
        } else if (!(SDAG_SCAN_EXP & 1)) {
#pragma unroll
            for (int round = 0; round < 4; ++round) {
                push_round(round * 4, row_base);
                lds_barrier();
                compact_check();
                lds_barrier();
            }
        }
        tile += TPS;
#pragma unroll
        for (int hh = 0; hh < TPS; ++hh)
#pragma unroll
            for (int qb = 0; qb < QPW; ++qb)
#pragma unroll
                for (int i = 0; i < 16; ++i) accs[hh][qb][i] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if constexpr (MODE == MODE_BMAX) return;
    if constexpr (THR) {
        flush_buffer();
        return;
    }
    for (int jj = 0; jj < LPW; ++jj) compact_list<CFG>(smem, w * LPW + jj, k, lane);
    lds_barrier();
    // ---- write this stripe's best k per query (fill: -FLT_MAX / -1)
    for (int e = tid; e < QTW * k; e += 64 * NW) {
        const int j = e / k, t = e - j * k;
        const int n = reinterpret_cast<const int*>(smem + OFF_CNT)[j];
        const bool ok = j < nq && t < n;
        const int64_t o = ((int64_t)blockIdx.x * QTW + j) * k + t;
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
                                                         float* __restrict__ out_s, int64_t* __restrict__ out_i,
                                                         const int* __restrict__ run_flag) {
    if (run_flag && *run_flag == 0) return;  // nothing to do (see sdag_dense_topk2)
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

// ---- two-stage search: bf16 candidate scan + exact rescoring ---------------------------------------------------
// Rounding a value to bf16 (round to nearest even) changes it by at most 2^-8 of its magnitude, so for the rounded
// copies q', c' of a query and a corpus row  |q.c - q'.c'| <= (2^-7 + 2^-16) |q| |c|  (Cauchy-Schwarz on the
// per-dimension terms); both this approximate score and the exact one are fp32 sums of d terms, each with a rounding
// error of at most d 2^-24 |q| |c|.  With eps = EPS_REL(d) |q| max|c| every approximate score is within eps of the
// exact score of the same row.  Then every row of the exact top k has an approximate score >= a_k - 2 eps, a_k being
// the k-th best approximate score: the k rows with the best approximate scores have exact scores >= a_k - eps, so the
// k-th best exact score is at least that, and a row that reaches it exactly is within eps of it approximately.
// The candidate scan keeps the best ks approximate scores per stripe (ks = scan_keep(k) >= k: a margin, so that a
// stripe holding several of the best rows does not look as if it had dropped one); the rescoring kernel takes every
// entry >= LB - 2 eps with LB <= a_k (see there), scores it with the exact fp32 chain and ranks (RS_G blocks per query,
// their lists merged by the merge kernel).  A stripe whose ks entries are all candidates, or more candidates than fit,
// raise a flag, and the exact scan then runs for the batch (the flag is read on the device: no host round trip).

__host__ __device__ inline float eps_rel(int d) { return 0.0079f + (float)d * 1.2e-7f; }

// Entries the candidate scan keeps per stripe and query for a search of the best k: k plus a margin of a quarter (at
// least 2), never fewer than 8.  Every kept entry costs scan time (the running lists are filtered and compacted per
// tile: about 12 us per entry at 2 M x 1024), so the margin is small; it decides how many near-ties of the best rows
// one stripe may hold before the search has to fall back to the exact scan.
inline int scan_keep(int k) {
    int m = k + (k / 4 > 2 ? k / 4 : 2);
    if (m < 8) m = 8;
    return m < MAX_K ? m : MAX_K;
}

// One block per query: bf16 copy (round to nearest even) and eps = eps_rel(d) * |q| * cmax, rounded up.  Block 0 also
// clears the fallback flags of the call's query batches (n_flags of them).
__global__ __launch_bounds__(256) void query_prepare_kernel(const float* __restrict__ Q, int d, const float* __restrict__ cmax,
                                                            unsigned short* __restrict__ Q16, float* __restrict__ eps,
                                                            int* __restrict__ flags, int n_flags, int qtw, int* __restrict__ cand_n) {
    __shared__ float red[256];
    const int q = blockIdx.x, tid = threadIdx.x;
    if (cand_n && tid == 0) cand_n[q] = 0;  // the query's candidate list of the threshold scan starts empty
    if (q == 0)
        for (int f = tid; f < n_flags; f += 256) flags[f] = 0;
    float ss = 0.f;
    for (int j = tid; j < d; j += 256) {
        const float x = Q[(int64_t)q * d + j];
        ss = __builtin_fmaf(x, x, ss);
        // slab-major per pass of qtw queries: [pass][64-dimension stage][query of the pass][64]: the 128-byte pieces of
        // one stage are contiguous, so every compute unit reading the same stage spreads over all L2 channels (rows of
        // a row-major matrix are d * 2 bytes apart and land on a few)
        const int64_t pass = q / qtw, jq = q % qtw;
        Q16[((pass * (d / 64) + j / 64) * qtw + jq) * 64 + (j & 63)] = __builtin_bit_cast(unsigned short, (__bf16)x);
    }
    red[tid] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) eps[q] = eps_rel(d) * sqrtf(red[0]) * cmax[0] * 1.001f + 1e-30f;
}

// bf16 copy of corpus rows and the largest row norm (as float bits: norms are non-negative, so integer order =
// float order).  One wave per row.
__global__ __launch_bounds__(256) void corpus_prepare_kernel(const float* __restrict__ C, int64_t N, int d,
                                                             unsigned short* __restrict__ C16, unsigned* __restrict__ max_norm_bits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float ss = 0.f;
    for (int j = lane * 4; j < d; j += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(C + row * d + j);
        ss = __builtin_fmaf(x[0], x[0], ss);
        ss = __builtin_fmaf(x[1], x[1], ss);
        ss = __builtin_fmaf(x[2], x[2], ss);
        ss = __builtin_fmaf(x[3], x[3], ss);
        ushort4 o;
        o.x = __builtin_bit_cast(unsigned short, (__bf16)x[0]);
        o.y = __builtin_bit_cast(unsigned short, (__bf16)x[1]);
        o.z = __builtin_bit_cast(unsigned short, (__bf16)x[2]);
        o.w = __builtin_bit_cast(unsigned short, (__bf16)x[3]);
        *reinterpret_cast<ushort4*>(C16 + row * d + j) = o;
    }
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    // Rows of one corpus have nearly equal norms: after the first few, a plain read shows that the row changes
    // nothing, and millions of atomics on one address (each a round trip to the L2) are avoided.
    if (lane == 0) {
        const unsigned bits = __float_as_uint(sqrtf(ss) * 1.0001f);
        if (bits > __atomic_load_n(max_norm_bits, __ATOMIC_RELAXED)) atomicMax(max_norm_bits, bits);
    }
}

// Rescoring: find the candidates in the stripe lists, score them with the exact chain, keep the best k.
//
// The stripe lists hold the best ks >= k approximate scores of every stripe, sorted best first.  Threshold: the k-th
// largest of the stripes' maxima is a lower bound LB of a_k (they are scores of k distinct rows), so every row of the
// exact top k has an approximate score >= LB - 2 eps; every list entry above that line is a candidate (a superset of
// the set a_k itself would give: no separate merge of the lists is needed to know a_k).  A stripe whose ks entries are
// ALL candidates may have dropped one, and more than RS_CB candidates among one block's stripes do not fit: either
// raises the flag that makes the exact scan run for the batch.
constexpr int RS_G = 16;      // blocks per query: block g takes the stripes p with p % RS_G == g
constexpr int RS_CB = 32;     // candidates one block holds (and scores together)
constexpr int RS_DC = 256;    // dimensions of every candidate row in LDS at a time
constexpr int RS_RW = RS_DC + 4;  // row stride of the LDS image in floats: 16 threads reading 16 bytes each hit 64 banks

// The k-th largest of the 512 keys a wave holds (8 per lane), ties counted with multiplicity: the key is built from the
// top bit down, keeping a bit whenever at least k keys reach the value it would give.  Wave-uniform, no LDS.
__device__ __forceinline__ unsigned wave_kth_largest(const unsigned (&key)[8], int k) {
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] >= trial));
        if (cnt >= k) prefix = trial;
    }
    return prefix;
}

// grid (RS_G, queries).  Every block finds the line (all stripes' maxima: 2 KB of loads), collects the candidates of
// ITS stripes, scores them with the exact chain and writes its best k, sorted, as one partial list per (block, query)
// in the layout the merge kernel reads; the merge of the RS_G lists gives the result.
// q_stride: queries per stripe in the scan's lists and per block in the lists written here (the scan's pass width).
__global__ __launch_bounds__(256) void rescore_kernel(const float* __restrict__ Q, const float* __restrict__ C, int d, int k, int ks,
                                                      const float* __restrict__ part_s, const int* __restrict__ part_i, int n_wg,
                                                      int q_stride, const float* __restrict__ eps, float* __restrict__ blk_s,
                                                      int* __restrict__ blk_i, int* __restrict__ flag) {
    __shared__ __attribute__((aligned(16))) float rows[RS_CB * RS_RW];
    __shared__ __attribute__((aligned(16))) float qs[RS_DC];
    __shared__ int cand[RS_CB];
    __shared__ float ex[RS_CB];
    __shared__ int qual[MAX_WG / RS_G];
    __shared__ int n_s, nqual_s, bad_s;
    const int g = blockIdx.x, q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { n_s = 0; nqual_s = 0; bad_s = 0; }
    // ---- the stripes' maxima (stripe 64 i + lane in key[i]; every wave holds all of them), and LB = the k-th largest
    // (-inf when fewer than k stripes hold a row: then everything is a candidate)
    unsigned key[MAX_WG / 64];
#pragma unroll
    for (int i = 0; i < MAX_WG / 64; ++i) {
        const int p = 64 * i + lane;
        const int64_t src = ((int64_t)(p < n_wg ? p : 0) * q_stride + q) * ks;
        const int id0 = part_i[src];   // both words requested together: one round trip
        const float s0 = part_s[src];
        key[i] = float_key(p < n_wg && id0 >= 0 ? s0 : -INFINITY);
    }
    const float lb = key_float(wave_kth_largest(key, k));
    const float thr = lb == -INFINITY ? -INFINITY : lb - 2.0f * eps[q];
    if (thr != thr && tid == 0) atomicOr(flag, 1);  // a line that is not a number selects nothing: let the exact scan answer
    const unsigned thr_key = float_key(thr);
    __syncthreads();  // the counters are zero
    // ---- candidates: this block's stripes whose maximum reaches the line, then every (stripe, place) pair of those at
    // once (the lists are sorted, so the entries above the line are a prefix of each)
    if (w == 0 && (lane % RS_G) == g % RS_G) {
#pragma unroll
        for (int i = 0; i < MAX_WG / 64; ++i) {
            const int p = 64 * i + lane;
            if (p % RS_G == g && p < n_wg && key[i] >= thr_key) qual[atomicAdd(&nqual_s, 1)] = p;
        }
    }
    __syncthreads();
    const int npairs = nqual_s * ks;
    for (int e = tid; e < npairs; e += 256) {
        const int p = qual[e / ks], t = e - (e / ks) * ks;
        const int64_t src = ((int64_t)p * q_stride + q) * ks + t;
        const int id = part_i[src];
        const float sc = part_s[src];
        if (id < 0 || !(sc >= thr)) continue;
        const int pos = atomicAdd(&n_s, 1);
        if (pos < RS_CB) cand[pos] = id;
        if (t == ks - 1) bad_s = 1;  // the stripe's list is full of candidates: it may have dropped one
    }
    __syncthreads();
    int n = n_s;
    if (n > RS_CB) {
        n = RS_CB;
        if (tid == 0) bad_s = 1;
    }
    // ---- exact scores: thread t < n walks candidate t's chain; the rows pass through LDS RS_DC dimensions at a time,
    // the next piece of every row already requested (in registers) while the chains consume the current one.  (A large
    // row image would be no faster: workgroups that take most of a compute unit's LDS start tens of microseconds apart.)
    constexpr int PER_ROW = RS_DC / 4, PIECES = RS_CB * PER_ROW / 256;
    f32x4 pre[PIECES], preq;
    auto fetch = [&](int dc) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int e = tid + 256 * i, rr = e / PER_ROW, c4 = e % PER_ROW;
            if (rr < n && dc + 4 * c4 < d) pre[i] = *reinterpret_cast<const f32x4*>(C + (int64_t)cand[rr] * d + dc + 4 * c4);
        }
        if (tid < PER_ROW && dc + 4 * tid < d) preq = *reinterpret_cast<const f32x4*>(Q + (int64_t)q * d + dc + 4 * tid);
    };
    float acc = 0.f;
    if (n > 0) fetch(0);
    for (int dc = 0; n > 0 && dc < d; dc += RS_DC) {
        const int dn = d - dc < RS_DC ? d - dc : RS_DC;  // a multiple of 64
        __syncthreads();  // the previous piece has been consumed
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int e = tid + 256 * i, rr = e / PER_ROW, c4 = e % PER_ROW;
            if (rr < n && dc + 4 * c4 < d) *reinterpret_cast<f32x4*>(rows + rr * RS_RW + 4 * c4) = pre[i];
        }
        if (tid < PER_ROW && dc + 4 * tid < d) *reinterpret_cast<f32x4*>(qs + 4 * tid) = preq;
        __syncthreads();
        if (dc + RS_DC < d) fetch(dc + RS_DC);
        if (tid < n) {
            // 32 dimensions of the row and of the query in registers while the 32 before them go through the chain
            const float* rp = rows + tid * RS_RW;
            f32x4 ca[8], qa[8], cb[8], qb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ca[u] = *reinterpret_cast<const f32x4*>(rp + 4 * u);
                qa[u] = *reinterpret_cast<const f32x4*>(qs + 4 * u);
            }
            for (int j = 0; j < dn; j += 64) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    cb[u] = *reinterpret_cast<const f32x4*>(rp + j + 32 + 4 * u);
                    qb[u] = *reinterpret_cast<const f32x4*>(qs + j + 32 + 4 * u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = __builtin_fmaf(ca[u][0], qa[u][0], acc);
                    acc = __builtin_fmaf(ca[u][1], qa[u][1], acc);
                    acc = __builtin_fmaf(ca[u][2], qa[u][2], acc);
                    acc = __builtin_fmaf(ca[u][3], qa[u][3], acc);
                }
                if (j + 64 < dn) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        ca[u] = *reinterpret_cast<const f32x4*>(rp + j + 64 + 4 * u);
                        qa[u] = *reinterpret_cast<const f32x4*>(qs + j + 64 + 4 * u);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc = __builtin_fmaf(cb[u][0], qb[u][0], acc);
                    acc = __builtin_fmaf(cb[u][1], qb[u][1], acc);
                    acc = __builtin_fmaf(cb[u][2], qb[u][2], acc);
                    acc = __builtin_fmaf(cb[u][3], qb[u][3], acc);
                }
            }
        }
    }
    if (tid < n) {
        const bool ok = acc == acc;  // a NaN score (infinities in the data) is never returned: it becomes a fill entry
        ex[tid] = ok ? acc : -FLT_MAX;
        if (!ok) cand[tid] = -1;
    }
    __syncthreads();
    // ---- this block's best k, sorted, as partial list g of query q (equal entries - fills - ordered by position, so
    // every rank is taken exactly once)
    const int64_t ob = ((int64_t)g * q_stride + q) * k;
    for (int c = tid; c < n; c += 256) {
        const float sc = ex[c];
        const int ic = cand[c];
        int rank = 0;
        for (int j2 = 0; j2 < n; ++j2)
            rank += (better(ex[j2], cand[j2], sc, ic) || (ex[j2] == sc && cand[j2] == ic && j2 < c)) ? 1 : 0;
        if (rank < k) {
            blk_s[ob + rank] = sc;
            blk_i[ob + rank] = ic;
        }
    }
    if (tid < k && tid >= n) {
        blk_s[ob + tid] = -FLT_MAX;
        blk_i[ob + tid] = -1;
    }
    if (tid == 0 && bad_s) atomicOr(flag, 1);
}

// The k-th largest of the keys a 256-thread block holds (8 per thread, 0 = none), ties counted with multiplicity, built
// from the top bit down like wave_kth_largest; the waves' counts meet in LDS (two alternating sets of four words, one
// barrier per bit).  Every thread returns the same value; 0 when fewer than k keys are set.
__device__ __forceinline__ unsigned block_kth_largest(const unsigned (&key)[8], int k, int* red /* [8] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned prefix = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned trial = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(key[i] >= trial));
        int* r4 = red + 4 * (bit & 1);
        if (lane == 0) r4[w] = cnt;
        __syncthreads();
        if (r4[0] + r4[1] + r4[2] + r4[3] >= k) prefix = trial;
    }
    return prefix;
}

// Rescoring for the threshold scan: the candidates of query q are the rows in its global list (approximate score at
// or above T = s_m - 2 eps, s_m the sample statistic the threshold was taken from), ALL such rows of the corpus unless
// something overflowed.  If at least k of them reach s_m, the k-th largest approximate score in the list is a_k, the
// k-th best approximate score of the whole corpus, and a_k >= s_m = T + 2 eps; every row of the exact top k has an
// approximate score >= a_k - 2 eps >= T, so it is in the list, and only the entries at or above the line a_k - 2 eps
// need their exact score.  Otherwise (fewer than k reach s_m, more than RS2_HI do, the list overflowed or the scan
// dropped survivors) the flag makes the exact scan answer.  Without a sample statistic the threshold is -FLT_MAX, the
// list is the whole corpus and every entry is scored.
// grid (RS2_G, queries): every block finds the line on its own (the list's scores at or above s_m, up to RS2_HI of
// them), takes the entries at list positions g, g + RS2_G, ... that reach it, scores them in chunks of RS_CB, ranks them
// and writes its best k as partial list g of the query, in the layout the merge kernel reads.  (Four blocks per query:
// finding the line costs a block some thirty barriers whatever it scores afterwards, and a 256-query pass with sixteen
// blocks per query is 4,096 blocks that each repeat it.)
#ifndef SDAG_RS2_G
#define SDAG_RS2_G 4
#endif
constexpr int RS2_G = SDAG_RS2_G;   // blocks per query
constexpr int THR_CAP_ = 4096;      // candidates per query the global lists hold
constexpr int RS2_MAX = THR_CAP_ / RS2_G;   // list entries one block looks at
constexpr int RS2_HI = 2048;   // scores at or above s_m a block can select among (8 per thread)
__global__ __launch_bounds__(256) void rescore_cand_kernel(const float* __restrict__ Q, const float* __restrict__ C, int d, int k,
                                                           const ThrArgs ta, int q_stride, float* __restrict__ blk_s,
                                                           int* __restrict__ blk_i) {
    __shared__ __attribute__((aligned(16))) float rows[RS_CB * RS_RW];
    __shared__ __attribute__((aligned(16))) float qs[RS_DC];
    __shared__ int cand[RS_CB];
    __shared__ float res_s[RS2_MAX];
    __shared__ int res_i[RS2_MAX];
    __shared__ int sel[RS2_MAX];
    __shared__ int red[8];
    __shared__ int n_hi_s, n_sel_s;
    const int g = blockIdx.x, q = blockIdx.y, tid = threadIdx.x;
    const int n_raw = ta.cand_n[q];
    const int n_all = n_raw < ta.cap ? n_raw : ta.cap;
    const float* ls = ta.cand_s + (int64_t)q * ta.cap;
    const int* li = ta.cand_i + (int64_t)q * ta.cap;
    const bool sampled = ta.pilot_i[(int64_t)q * ta.m + ta.m - 1] >= 0;
    const float s_m = ta.pilot_s[(int64_t)q * ta.m + ta.m - 1];
    if (tid == 0) { n_hi_s = 0; n_sel_s = 0; }
    __syncthreads();
    // ---- the line: a_k - 2 eps, a_k = the k-th largest of the list's scores that reach s_m
    float line = -FLT_MAX;
    bool bad = n_raw > ta.cap;
    if (sampled) {
        unsigned key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int e = tid; e < n_all; e += 256) {
            const float x = ls[e];
            if (x >= s_m) {
                const int pos = atomicAdd(&n_hi_s, 1);
                // thread (pos % 256) register (pos / 256) would need a shuffle: park the value in LDS instead
                if (pos < RS2_HI) rows[pos] = x;
            }
        }
        __syncthreads();
        const int n_hi = n_hi_s;
        if (n_hi < k || n_hi > RS2_HI) {
            bad = true;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (tid + 256 * i < n_hi) key[i] = float_key(rows[tid + 256 * i]);
            __syncthreads();  // (rows is reused below)
            line = key_float(block_kth_largest(key, k, red)) - 2.0f * ta.eps[q];
        }
    }
    if (bad && g == 0 && tid == 0) atomicOr(ta.flag, 1);
    // ---- this block's entries at or above the line
    if (!(bad && sampled))
        for (int e = g + tid * RS2_G; e < n_all; e += 256 * RS2_G)
            if (ls[e] >= line) sel[atomicAdd(&n_sel_s, 1)] = li[e];
    __syncthreads();
    const int n_mine = n_sel_s;
    constexpr int PER_ROW = RS_DC / 4, PIECES = RS_CB * PER_ROW / 256;
    for (int c0 = 0; c0 < n_mine; c0 += RS_CB) {
        const int n = n_mine - c0 < RS_CB ? n_mine - c0 : RS_CB;
        __syncthreads();  // the previous chunk's rows and candidates have been consumed
        if (tid < n) cand[tid] = sel[c0 + tid];
        __syncthreads();
        f32x4 pre[PIECES], preq;
        auto fetch = [&](int dc) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int e = tid + 256 * i, rr = e / PER_ROW, c4 = e % PER_ROW;
                if (rr < n && dc + 4 * c4 < d) pre[i] = *reinterpret_cast<const f32x4*>(C + (int64_t)cand[rr] * d + dc + 4 * c4);
            }
            if (tid < PER_ROW && dc + 4 * tid < d) preq = *reinterpret_cast<const f32x4*>(Q + (int64_t)q * d + dc + 4 * tid);
        };
        float acc = 0.f;
        fetch(0);
        for (int dc = 0; dc < d; dc += RS_DC) {
            const int dn = d - dc < RS_DC ? d - dc : RS_DC;  // a multiple of 64
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const int e = tid + 256 * i, rr = e / PER_ROW, c4 = e % PER_ROW;
                if (rr < n && dc + 4 * c4 < d) *reinterpret_cast<f32x4*>(rows + rr * RS_RW + 4 * c4) = pre[i];
            }
            if (tid < PER_ROW && dc + 4 * tid < d) *reinterpret_cast<f32x4*>(qs + 4 * tid) = preq;
            __syncthreads();
            if (dc + RS_DC < d) fetch(dc + RS_DC);
            if (tid < n) {
                const float* rp = rows + tid * RS_RW;
                for (int j = 0; j < dn; j += 4) {  // the one fp32 FMA chain over the dimensions in order
                    const f32x4 cv = *reinterpret_cast<const f32x4*>(rp + j);
                    const f32x4 qv = *reinterpret_cast<const f32x4*>(qs + j);
                    acc = __builtin_fmaf(cv[0], qv[0], acc);
                    acc = __builtin_fmaf(cv[1], qv[1], acc);
                    acc = __builtin_fmaf(cv[2], qv[2], acc);
                    acc = __builtin_fmaf(cv[3], qv[3], acc);
                }
            }
        }
        if (tid < n) {
            const bool ok = acc == acc;  // a NaN score is never returned: it becomes a fill entry
            res_s[c0 + tid] = ok ? acc : -FLT_MAX;
            res_i[c0 + tid] = ok ? cand[tid] : -1;
        }
    }
    __syncthreads();
    // ---- this block's best k, sorted (equal entries - fills - ordered by position: every rank is taken exactly once)
    const int64_t ob = ((int64_t)g * q_stride + q) * k;
    for (int c = tid; c < n_mine; c += 256) {
        const float sc = res_s[c];
        const int ic = res_i[c];
        int rank = 0;
        for (int j2 = 0; j2 < n_mine; ++j2)
            rank += (better(res_s[j2], res_i[j2], sc, ic) || (res_s[j2] == sc && res_i[j2] == ic && j2 < c)) ? 1 : 0;
        if (rank < k) {
            blk_s[ob + rank] = sc;
            blk_i[ob + rank] = ic;
        }
    }
    if (tid < k && tid >= n_mine) {
        blk_s[ob + tid] = -FLT_MAX;
        blk_i[ob + tid] = -1;
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

// The two kernels above for a whole batch of queries in one launch: block b owns query b (rows r_off[b] .. r_off[b+1]
// of R, candidates c_off[b] .. c_off[b+1] of Cand); same summation order per element, so the same bits.
__global__ __launch_bounds__(256) void centroid_batch_kernel(const float* __restrict__ R, const int* __restrict__ r_off,
                                                             const float* __restrict__ Cand, const int* __restrict__ c_off, int d,
                                                             float* __restrict__ centroid, float* __restrict__ sims) {
    const int b = blockIdx.x;
    const int r0 = r_off[b], nr = r_off[b + 1] - r0;
    const int c0 = c_off[b], nc = c_off[b + 1] - c0;
    float* cen = centroid + (int64_t)b * d;
    for (int j = threadIdx.x; j < d; j += 256) {
        float s = 0.f;
        for (int i = 0; i < nr; ++i) s += R[(int64_t)(r0 + i) * d + j];
        cen[j] = s / (float)nr;
    }
    __threadfence_block();
    __syncthreads();
    for (int c = threadIdx.x; c < nc; c += 256) {
        float s = 0.f;
        for (int j = 0; j < d; ++j) s = __builtin_fmaf(Cand[(int64_t)(c0 + c) * d + j], cen[j], s);
        sims[c0 + c] = s;
    }
}

// Query blocks per pass for a search of nq queries keeping ks entries per stripe: as many as the queries fill, the
// widest only while its shorter lists hold ks entries.
inline int pass_blocks(int nq, int ks) {
    if (nq <= QT) return 1;
    if (nq <= 2 * QT || ks > WIDE_KS) return 2;
    return MAX_NQB;
}

struct ScanPlan { int n_wg; int rows_per_wg; };
// Stripes of a scan with nqb query blocks per pass: what the chip holds at once (the narrow scan runs two workgroups per
// compute unit, the wide ones one of twice the waves).
inline ScanPlan plan_scan(int64_t N, int nqb = 1) {
    const int max_wg = nqb == 1 ? SCAN_WG : 256;
    int64_t tiles = (N + TR - 1) / TR;
    int64_t n_wg = tiles < max_wg ? tiles : max_wg;
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

namespace {
// slab_width: 0 = the queries are a row-major [nq][d] matrix, else they are stored slab-major for passes of slab_width
// queries (see query_prepare_kernel) and Q points at the first query of the launch inside its pass.
template <bool BF16, int NQB, int NW, int MODE>
int launch_scan_cfg(int n_wg, hipStream_t st, const void* Q, int nqt, const void* C, int64_t N, int d, int k, int rows_per_wg,
                    float* part_s, int* part_i, const int* run_flag, int slab_width, int tile_step, const ThrArgs& ta) {
    using CFG = ScanCfg<NQB, NW, MODE != 0>;
    const int es = BF16 ? 2 : 4;
    const int q_row_bytes = slab_width ? 128 : d * es, q_kc_bytes = slab_width ? 128 * slab_width : 128;
    SDAG_HIP_CHECK(hipFuncSetAttribute((const void*)dense_scan<BF16, NQB, NW, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, CFG::LDS));
    hipLaunchKernelGGL((dense_scan<BF16, NQB, NW, MODE>), dim3(n_wg), dim3(64 * NW), CFG::LDS, st, Q, nqt, C, N, d, k, rows_per_wg, part_s,
                       part_i, run_flag, q_row_bytes, q_kc_bytes, tile_step, ta);
    SDAG_HIP_CHECK(hipGetLastError());
    return SDAG_OK;
}

// One pass of the list scan over the corpus (or a sample of it) for nqt <= 32 nqb queries.
template <bool BF16>
int launch_scan(int nqb, int n_wg, hipStream_t st, const void* Q, int nqt, const void* C, int64_t N, int d, int k, int rows_per_wg,
                float* part_s, int* part_i, const int* run_flag, int slab_width, int tile_step = 1, const ThrArgs& none = ThrArgs{}) {
    switch (nqb) {
        case 1: return launch_scan_cfg<BF16, 1, 4, MODE_LISTS>(n_wg, st, Q, nqt, C, N, d, k, rows_per_wg, part_s, part_i, run_flag, slab_width, tile_step, none);
        case 2: return launch_scan_cfg<BF16, 2, 8, MODE_LISTS>(n_wg, st, Q, nqt, C, N, d, k, rows_per_wg, part_s, part_i, run_flag, slab_width, tile_step, none);
        case 4: return launch_scan_cfg<BF16, 4, 8, MODE_LISTS>(n_wg, st, Q, nqt, C, N, d, k, rows_per_wg, part_s, part_i, run_flag, slab_width, tile_step, none);
    }
    return fail(SDAG_ERR_ARG, "bad pass width");
}

// One pass of the threshold scan (bf16 copies) for nqt <= 32 nqb queries, or (tile_step > 1) its pilot over a sample.
template <int MODE>
int launch_thr_scan(int nqb, int n_wg, hipStream_t st, const void* Q16, int nqt, const void* C16, int64_t N, int d, int rows_per_wg,
                    int tile_step, const ThrArgs& ta) {
    switch (nqb) {
        case 2: return launch_scan_cfg<true, 2, 8, MODE>(n_wg, st, Q16, nqt, C16, N, d, 0, rows_per_wg, nullptr, nullptr, nullptr, 32 * nqb, tile_step, ta);
        case 4: return launch_scan_cfg<true, 4, 8, MODE>(n_wg, st, Q16, nqt, C16, N, d, 0, rows_per_wg, nullptr, nullptr, nullptr, 32 * nqb, tile_step, ta);
        case 8: return launch_scan_cfg<true, 8, 8, MODE>(n_wg, st, Q16, nqt, C16, N, d, 0, rows_per_wg, nullptr, nullptr, nullptr, 32 * nqb, tile_step, ta);
    }
    return fail(SDAG_ERR_ARG, "bad pass width");
}
}  // namespace

extern "C" int64_t sdag_dense_workspace_bytes(int nq, int64_t N, int d, int k) {
    (void)d;
    if (k <= 0 || k > MAX_K || N < 0 || nq < 0) return -1;
    const int nqb = pass_blocks(nq, k);
    const ScanPlan pl = plan_scan(N, nqb);
    return ((int64_t)pl.n_wg * QT * nqb * k * 8 + 255) & ~255ll;
}

namespace {
// The exact scan + merge; after_s / after_i (both or neither): a cursor per query, see ThrArgs.
int dense_topk_exact(const float* Q, int nq, const float* C, int64_t N, int d, int k, const float* after_s, const int64_t* after_i,
                     float* out_s, int64_t* out_i, int64_t id_base, void* workspace, int64_t workspace_bytes, void* stream) {
    if (nq < 0 || N < 0) return fail(SDAG_ERR_ARG, "negative nq or N");
    if (d <= 0 || d % KC) return fail(SDAG_ERR_ARG, "d=%d must be a positive multiple of %d", d, KC);
    if (k <= 0 || k > MAX_K) return fail(SDAG_ERR_ARG, "k=%d outside [1, %d]", k, MAX_K);
    if (N > INT_MAX) return fail(SDAG_ERR_ARG, "N too large for one shard");
    if (nq == 0) return SDAG_OK;
    if (!Q || !out_s || !out_i || (N > 0 && !C)) return fail(SDAG_ERR_ARG, "NULL pointer");
    if (((uintptr_t)Q | (uintptr_t)C) & 15) return fail(SDAG_ERR_ARG, "Q and C must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nqb = pass_blocks(nq, k), qtw = QT * nqb;
    const ScanPlan pl = plan_scan(N, nqb);
    const int64_t need = sdag_dense_workspace_bytes(nq, N, d, k);
    if (!workspace || workspace_bytes < need) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)need);
    float* part_s = (float*)workspace;
    int* part_i = (int*)(part_s + (int64_t)pl.n_wg * qtw * k);
    if ((( (int64_t)pl.n_wg * k + 63) / 64) * k > MRG_A) return fail(SDAG_ERR_ARG, "too many partial lists for the merge");
    for (int q0 = 0; q0 < nq; q0 += qtw) {
        const int nqt = nq - q0 < qtw ? nq - q0 : qtw;
        ThrArgs ta = {};
        if (after_s) {
            ta.after_s = after_s + q0;
            ta.after_i = after_i + q0;
            ta.after_base = id_base;
        }
        if (N > 0)
            if (int rc = launch_scan<false>(nqb, pl.n_wg, st, (const void*)(Q + (int64_t)q0 * d), nqt, (const void*)C, N, d, k,
                                            pl.rows_per_wg, part_s, part_i, (const int*)nullptr, 0, 1, ta))
                return rc;
        hipLaunchKernelGGL(topk_merge_kernel<int>, dim3(nqt), dim3(MRG_THREADS), 0, st, part_s, part_i, N > 0 ? pl.n_wg : 0, nqt, qtw,
                           k, id_base, out_s + (int64_t)q0 * k, out_i + (int64_t)q0 * k, (const int*)nullptr);
        SDAG_HIP_CHECK(hipGetLastError());
    }
    return SDAG_OK;
}
}  // namespace

extern "C" int sdag_dense_topk(const float* Q, int nq, const float* C, int64_t N, int d, int k,
                               float* out_s, int64_t* out_i, int64_t id_base,
                               void* workspace, int64_t workspace_bytes, void* stream) {
    return dense_topk_exact(Q, nq, C, N, d, k, nullptr, nullptr, out_s, out_i, id_base, workspace, workspace_bytes, stream);
}

extern "C" int sdag_dense_topk_after(const float* Q, int nq, const float* C, int64_t N, int d, int k,
                                     const float* after_s, const int64_t* after_i,
                                     float* out_s, int64_t* out_i, int64_t id_base,
                                     void* workspace, int64_t workspace_bytes, void* stream) {
    if (nq > 0 && (!after_s || !after_i)) return fail(SDAG_ERR_ARG, "NULL cursor");
    return dense_topk_exact(Q, nq, C, N, d, k, after_s, after_i, out_s, out_i, id_base, workspace, workspace_bytes, stream);
}

extern "C" int sdag_corpus_to_bf16(const float* C, int64_t N, int d, void* C16, float* max_row_norm, void* stream) {
    if (N < 0 || d <= 0 || d % 4) return fail(SDAG_ERR_ARG, "bad N or d (d must be a positive multiple of 4)");
    if (!max_row_norm || (N > 0 && (!C || !C16))) return fail(SDAG_ERR_ARG, "NULL pointer");
    if (((uintptr_t)C | (uintptr_t)C16) & 15) return fail(SDAG_ERR_ARG, "C and C16 must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SDAG_HIP_CHECK(hipMemsetAsync(max_row_norm, 0, sizeof(float), st));
    if (N > 0) {
        if ((N + 3) / 4 > 0x7fffffff) return fail(SDAG_ERR_ARG, "N too large");
        hipLaunchKernelGGL(corpus_prepare_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, C, N, d, (unsigned short*)C16,
                           (unsigned*)max_row_norm);
        SDAG_HIP_CHECK(hipGetLastError());
    }
    return SDAG_OK;
}

namespace {
// ---- plan of a two-stage search.  Up to 32 queries (or a small corpus): list scan + rescore_kernel, as described
// above.  More queries: per pass of up to 256 queries a pilot (the block maxima of every THR_STEP-th tile, of which the
// merge kernel keeps the m best per query), the threshold scan, rescore_cand_kernel; the exact list scan stands by in
// sub-passes of up to 128.
constexpr int THR_STEP = 32;            // the pilot scans one 128-row tile in THR_STEP
constexpr int THR_MIN_TILES = 8;        // ... and needs at least this many (else: the list scan)
constexpr int THR_CAP = THR_CAP_;
// Entries the pilot keeps per query: the number of rows of the whole corpus at or above the sample's m-th best score is
// about THR_STEP times a Gamma(m) variate; it has to reach k for the candidate list to be vouched for, and with
// m = 4 + ceil(k / 6) it fails to with a probability of some 1e-5 per query (k = 10: m = 6, 190 such rows expected).
inline int pilot_m(int k) { return 4 + (k + 5) / 6; }

struct Ws2 {
    bool thr;            // threshold-scan path
    int nqb, qtw;        // query blocks and queries per pass
    int n_flags;         // passes
    int m;               // the pilot's statistic: the m-th largest block maximum of the sample
    int64_t bm_parts;    // block maxima per query in groups of m (the merge kernel's lists)
    ScanPlan main, pilot, exact;  // stripes of the candidate scan, of the pilot, of the stand-by exact scan
    int ex_nqb;          // query blocks of an exact sub-pass
    int64_t part_s, part_i, flag, eps, q16, exact_l, rs, pil_s, pil_i, cand_s, cand_i, cand_n, total;
};
Ws2 plan_ws2(int nq, int64_t N, int d, int k) {
    const int ks = scan_keep(k);
    Ws2 w = {};
    const int64_t tiles = (N + TR - 1) / TR;
    const int64_t sampled = (tiles + THR_STEP - 1) / THR_STEP;
    w.thr = nq > QT && sampled >= THR_MIN_TILES;
    w.m = pilot_m(k);
    if (w.thr && (((sampled * 4 + w.m - 1) / w.m * w.m + 63) / 64) * w.m > MRG_A) w.thr = false;  // too many block maxima to merge
    if (w.thr) {
        w.qtw = nq <= 64 ? 64 : nq <= 128 ? 128 : 256;
        w.nqb = w.qtw / QT;
        w.main = plan_scan(N, w.nqb);
        w.pilot = plan_scan(sampled * TR, 2);  // (stripes over the sampled tiles)
        w.bm_parts = (sampled * 4 + w.m - 1) / w.m;
        w.ex_nqb = k > WIDE_KS ? 2 : (w.nqb < MAX_NQB ? w.nqb : MAX_NQB);
        w.exact = plan_scan(N, w.ex_nqb);
    } else {
        w.nqb = pass_blocks(nq, ks);
        w.qtw = QT * w.nqb;
        w.main = plan_scan(N, w.nqb);
        w.ex_nqb = w.nqb;
        w.exact = w.main;
    }
    const int qtw = w.qtw;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off += (bytes + 255) & ~255ll; return o; };
    // stripe lists of the candidate list scan, or the pilot's block maxima
    const int64_t list_entries = w.thr ? w.bm_parts * qtw * w.m : (int64_t)w.main.n_wg * qtw * ks;
    w.part_s = take(list_entries * 4);
    w.part_i = take(list_entries * 4);
    w.n_flags = (nq + qtw - 1) / qtw;  // one fallback flag per pass over the corpus
    w.flag = take((int64_t)(w.n_flags > 0 ? w.n_flags : 1) * 4);
    w.eps = take((int64_t)nq * 4);
    w.q16 = take((int64_t)w.n_flags * qtw * d * 2);  // whole passes (slab-major)
    w.exact_l = take((int64_t)w.exact.n_wg * QT * w.ex_nqb * k * 8);
    w.rs = take((int64_t)RS_G * qtw * k * 8);  // the rescoring blocks' partial lists
    if (w.thr) {
        w.pil_s = take((int64_t)nq * w.m * 4);
        w.pil_i = take((int64_t)nq * w.m * 8);
        w.cand_s = take((int64_t)nq * THR_CAP * 4);
        w.cand_i = take((int64_t)nq * THR_CAP * 4);
        w.cand_n = take((int64_t)nq * 4);
    }
    w.total = off;
    return w;
}
}  // namespace

extern "C" int64_t sdag_dense_topk2_workspace_bytes(int nq, int64_t N, int d, int k) {
    if (k <= 0 || k > MAX_K || N < 0 || nq < 0 || d <= 0) return -1;
    return plan_ws2(nq, N, d, k).total;
}

extern "C" int64_t sdag_dense_topk2_flag_offset(int nq, int64_t N, int d, int k) {
    if (k <= 0 || k > MAX_K || N < 0 || nq < 0 || d <= 0) return -1;
    const Ws2 w = plan_ws2(nq, N, d, k);
    return w.flag + (int64_t)(w.n_flags > 0 ? w.n_flags - 1 : 0) * 4;  // the last pass's flag
}

extern "C" int sdag_dense_topk2(const float* Q, int nq, const float* C, const void* C16, const float* max_row_norm, int64_t N,
                                int d, int k, float* out_s, int64_t* out_i, int64_t id_base, void* workspace,
                                int64_t workspace_bytes, void* stream) {
    if (nq < 0 || N < 0) return fail(SDAG_ERR_ARG, "negative nq or N");
    if (d <= 0 || d % KC16) return fail(SDAG_ERR_ARG, "d=%d must be a positive multiple of %d", d, KC16);
    if (k <= 0 || k > MAX_K) return fail(SDAG_ERR_ARG, "k=%d outside [1, %d]", k, MAX_K);
    if (N > INT_MAX) return fail(SDAG_ERR_ARG, "N too large for one shard");
    if (nq == 0) return SDAG_OK;
    if (!Q || !out_s || !out_i || !max_row_norm || (N > 0 && (!C || !C16))) return fail(SDAG_ERR_ARG, "NULL pointer");
    if (((uintptr_t)Q | (uintptr_t)C | (uintptr_t)C16) & 15) return fail(SDAG_ERR_ARG, "Q, C and C16 must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const Ws2 w = plan_ws2(nq, N, d, k);
    const int nqb = w.nqb, qtw = w.qtw;
    if (!workspace || workspace_bytes < w.total) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)w.total);
    unsigned char* ws = (unsigned char*)workspace;
    float* part_s = (float*)(ws + w.part_s);
    int* part_i = (int*)(ws + w.part_i);
    int* flags = (int*)(ws + w.flag);
    float* eps = (float*)(ws + w.eps);
    unsigned short* q16 = (unsigned short*)(ws + w.q16);
    const int ex_qtw = QT * w.ex_nqb;
    float* ex_s = (float*)(ws + w.exact_l);
    int* ex_i = (int*)(ex_s + (int64_t)w.exact.n_wg * ex_qtw * k);
    float* rs_s = (float*)(ws + w.rs);
    int* rs_i = (int*)(rs_s + (int64_t)RS_G * qtw * k);
    const int ks = scan_keep(k);
    if ((((int64_t)w.main.n_wg * k + 63) / 64) * k > MRG_A || (((int64_t)w.exact.n_wg * k + 63) / 64) * k > MRG_A)
        return fail(SDAG_ERR_ARG, "too many partial lists for the merge");
    int* cand_n = w.thr ? (int*)(ws + w.cand_n) : nullptr;
    hipLaunchKernelGGL(query_prepare_kernel, dim3(nq), dim3(256), 0, st, Q, d, max_row_norm, q16, eps, flags, w.n_flags, qtw, cand_n);
    SDAG_HIP_CHECK(hipGetLastError());
    for (int q0 = 0; q0 < nq; q0 += qtw) {
        const int nqt = nq - q0 < qtw ? nq - q0 : qtw;
        float* os = out_s + (int64_t)q0 * k;
        int64_t* oi = out_i + (int64_t)q0 * k;
        int* flag = flags + q0 / qtw;
        const unsigned short* q16p = q16 + (int64_t)q0 * d;  // this pass's slab-major block
        if (w.thr) {
            ThrArgs ta = {};
            ta.pilot_s = (float*)(ws + w.pil_s) + (int64_t)q0 * w.m;
            ta.pilot_i = (int64_t*)(ws + w.pil_i) + (int64_t)q0 * w.m;
            ta.m = w.m;
            ta.eps = eps + q0;
            ta.cand_s = (float*)(ws + w.cand_s) + (int64_t)q0 * THR_CAP;
            ta.cand_i = (int*)(ws + w.cand_i) + (int64_t)q0 * THR_CAP;
            ta.cand_n = cand_n + q0;
            ta.cap = THR_CAP;
            ta.flag = flag;
            // pilot: the largest approximate score of every 32-row block of one tile in THR_STEP, then the m-th largest of
            // those per query
            ta.bm_s = part_s;
            ta.bm_i = part_i;
            ta.bm_m = w.m;
            ta.bm_qtw = qtw;
            SDAG_HIP_CHECK(hipMemsetAsync(part_i, 0xff, (size_t)(w.bm_parts * qtw * w.m) * 4, st));  // -1: no entry
            if (int rc = launch_thr_scan<MODE_BMAX>(nqb, w.pilot.n_wg, st, (const void*)q16p, nqt, C16, N, d, w.pilot.rows_per_wg, THR_STEP, ta))
                return rc;
            hipLaunchKernelGGL(topk_merge_kernel<int>, dim3(nqt), dim3(MRG_THREADS), 0, st, part_s, part_i, (int)w.bm_parts, nqt, qtw, w.m,
                               (int64_t)0, const_cast<float*>(ta.pilot_s), const_cast<int64_t*>(ta.pilot_i), (const int*)nullptr);
            SDAG_HIP_CHECK(hipGetLastError());
            if (int rc = launch_thr_scan<MODE_THR>(nqb, w.main.n_wg, st, (const void*)q16p, nqt, C16, N, d, w.main.rows_per_wg, 1, ta)) return rc;
            hipLaunchKernelGGL(rescore_cand_kernel, dim3(RS2_G, nqt), dim3(256), 0, st, Q + (int64_t)q0 * d, C, d, k, ta, qtw, rs_s, rs_i);
            SDAG_HIP_CHECK(hipGetLastError());
        } else {
            if (N > 0)
                if (int rc = launch_scan<true>(nqb, w.main.n_wg, st, (const void*)q16p, nqt, C16, N, d, ks, w.main.rows_per_wg, part_s,
                                               part_i, (const int*)nullptr, qtw))
                    return rc;
            hipLaunchKernelGGL(rescore_kernel, dim3(RS_G, nqt), dim3(256), 0, st, Q + (int64_t)q0 * d, C, d, k, ks, part_s, part_i,
                               N > 0 ? w.main.n_wg : 0, qtw, eps + q0, rs_s, rs_i, flag);
            SDAG_HIP_CHECK(hipGetLastError());
        }
        hipLaunchKernelGGL(topk_merge_kernel<int>, dim3(nqt), dim3(MRG_THREADS), 0, st, rs_s, rs_i, w.thr ? RS2_G : RS_G, nqt, qtw, k, id_base, os, oi,
                           (const int*)nullptr);
        SDAG_HIP_CHECK(hipGetLastError());
        if (N > 0 && !SDAG_SCAN_EXP) {
            // only when the rescoring kernel could not vouch for its candidate set: the exact scan of the pass
            for (int s0 = 0; s0 < nqt; s0 += ex_qtw) {
                const int nqs = nqt - s0 < ex_qtw ? nqt - s0 : ex_qtw;
                if (int rc = launch_scan<false>(w.ex_nqb, w.exact.n_wg, st, (const void*)(Q + (int64_t)(q0 + s0) * d), nqs, (const void*)C,
                                                N, d, k, w.exact.rows_per_wg, ex_s, ex_i, (const int*)flag, 0))
                    return rc;
                hipLaunchKernelGGL(topk_merge_kernel<int>, dim3(nqs), dim3(MRG_THREADS), 0, st, ex_s, ex_i, w.exact.n_wg, nqs, ex_qtw, k,
                                   id_base, os + (int64_t)s0 * k, oi + (int64_t)s0 * k, (const int*)flag);
                SDAG_HIP_CHECK(hipGetLastError());
            }
        }
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
                       (int64_t)0, out_s, out_i, (const int*)nullptr);
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

extern "C" int sdag_centroid_sims_batch(const float* R, const int32_t* r_off, const float* Cand, const int32_t* c_off, int nb, int d,
                                        float* centroid, float* sims, void* stream) {
    if (nb < 0 || d <= 0) return fail(SDAG_ERR_ARG, "bad nb or d");
    if (nb == 0) return SDAG_OK;
    if (!R || !r_off || !Cand || !c_off || !centroid || !sims) return fail(SDAG_ERR_ARG, "NULL pointer");
    hipLaunchKernelGGL(centroid_batch_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, R, r_off, Cand, c_off, d, centroid, sims);
    SDAG_HIP_CHECK(hipGetLastError());
    return SDAG_OK;
}
