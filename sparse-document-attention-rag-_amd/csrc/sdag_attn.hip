// SDAG block-sparse attention prefill for gfx950 (MI355X), bf16 in, fp32 softmax/accumulate.
//
// Replaces the dense-masked attention the reference reaches from
//   model(input_ids=input_ids, attention_mask=attn[1,1,L,L], use_cache=True)
// (src/pipeline/sparse_attention_RAG/SDAG.py:176; transformers' LlamaAttention / eager_attention_forward:
// repeat_kv, Q K^T * scaling + mask, fp32 softmax, P V, output [B, L, Hq, D]).
//
// Work decomposition
//   The host deals the descriptor's units (a 32-row query block x a run of 64-key tiles) to work streams; one
//   persistent workgroup (4 waves) walks one stream for ONE key/value head, a step being one tile of one unit.  The
//   four waves take four query heads that share that key/value head (grouped-query attention), so one K/V tile
//   staged in LDS feeds four heads, and all four waves see the same mask.
//   blockIdx % (Hkv * head_groups) selects the key/value head, so with 8 KV heads every block that reads a
//   given head's K/V runs on the same XCD and shares that XCD's L2.
// Per step, per wave (the loop body is sdag_attn_step.inc)
//   S^T = K Q^T   (16 x v_mfma_f32_32x32x16_bf16 on top of accumulators initialised from the tile's lane mask;
//                  keys in the accumulator rows, the query on the lane, so the softmax row statistics are
//                  lane-local plus one exchange between the two 32-lane halves)
//   online softmax in fp32 registers with a deferred running maximum
//   O^T += V^T P^T (16 x v_mfma_f32_32x32x16_f16; the S^T accumulators, converted to fp16, are directly the B
//                  operand; V^T comes from LDS through ds_read_b64_tr_b16)
// The next tile's K rows go from global memory straight into an XOR-swizzled LDS image (LDS-DMA); its V rows pass
// through registers, where they are converted to fp16, into an image of the same shape; both images are
// conflict-free for the row reads of K and the transposed reads of V, double buffered, one barrier a step.
// Row blocks with long tile runs are split into several units; their partial (O, m, l) go to a workspace and a
// second kernel merges them.  Rows that allow nothing get mean(V) (uniform softmax over all L keys), from
// per-chunk column sums that extra blocks of the main launch compute.
#include <hip/hip_runtime.h>

#include <mutex>

#include "common.h"

// Scratch memory of sdag_attn_fwd_static (partial results of split row blocks and column sums of V): part of the
// library's device image, so the entry point has nothing to allocate.  Each GPU holds its own copy at its own
// address; in host code the name only stands for the symbol, the address comes from hipGetSymbolAddress.
static __device__ __align__(256) unsigned char scratch[SDAG_STATIC_SCRATCH_BYTES];

namespace sdag {
#ifdef SDAG_WGTIMES
// Diagnostic build only: per-workgroup wall time (100 MHz ticks), steps walked, start tick, place and shader cycles
// of the last launch.
__device__ unsigned g_wg_ticks[8192][6];
#endif
#ifdef SDAG_STAMPS
__device__ unsigned long long g_stamps[16];
#endif
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int D = 128;            // head dim
constexpr int RB = 32;            // query rows per workgroup (one MFMA tile)
constexpr int KT = 64;            // keys per tile
constexpr int TILE_BYTES = KT * D * 2;
constexpr int VSUM_CHUNKS = 32;   // column-sum partials per (batch, kv head)
constexpr float RESCALE_LOG2 = 8.0f;  // the running maximum may lag the true one by up to 2^8

struct AttnParams {
    const unsigned short* q;
    const unsigned short* k;
    const unsigned short* v;
    void* o;
    int q_sb, q_sh, q_sl;  // element strides (each below 2^31; products are taken in 64 bits)
    int k_sb, k_sh, k_sl;
    int v_sb, v_sh, v_sl;
    int o_sb, o_sl, o_sh;
    int B, Hq, Hkv;
    int G;        // query heads per kv head
    int HG;       // groups of 4 query heads per kv head
    int n_main_blocks;
    int out_f32;
    float c;      // scale * log2(e)
    const int32_t* desc;  // single descriptor shared by every batch element (btab == nullptr)
    const int32_t* btab;  // batch table + per-sequence descriptors (see common.h), or nullptr
    const int32_t* tab;   // whichever of the two is set: the table the streams and steps live in
    int n_streams;        // work streams of the table
    int off_streams, off_steps;  // word offsets of the stream offsets and the step records in the table
    int n_slots;          // partial-result slots of the shared descriptor
    int slots_per_b;      // shared descriptor: n_slots (slots of batch element b start at b * n_slots); batch table: 0
    float* part_o;   // [global slot][Hq][32][128]
    float* part_ml;  // [global slot][Hq][32][2]
    float* vsum;     // [B][Hkv][VSUM_CHUNKS][128]
};

// Byte offset of 16-byte chunk `ch` of row `row` inside a [64][128] bf16 tile image.  The XOR makes both
// ds_read_b128 row reads (32 lanes = 32 rows, one chunk column) and ds_read_b64_tr_b16 4-row block reads
// conflict-free.
__device__ __forceinline__ int tile_off(int row, int ch) {
    return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

#ifdef SDAG_STAMPS
// Diagnostic build only: per-phase cycle sums of wave 0 of block 0, read back with sdag_debug_stamps().
#define STAMP(i)                                                            \
    do {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                  \
        unsigned long long _t;                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); \
        st_acc[i] += _t - st_prev;                                          \
        st_prev = _t;                                                       \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#else
#define STAMP(i)
#endif

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Three-input maximum as ONE instruction.  Written in asm because fmaxf() on accumulator outputs makes the compiler
// add a canonicalising v_max per operand.  The first argument must already have been touched by compiler-visible
// code after the MFMAs that produced the others, so the MFMA -> VALU wait states have been inserted.
__device__ __forceinline__ float max3(float acc, float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(acc), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max3_abs(float acc, float a, float b) {
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(acc), "v"(a), "v"(b));
    return r;
}

// Combine a value with the one held by the lane 32 away.  v_permlane32_swap_b32 a, b trades lanes 32-63 of a with
// lanes 0-31 of b; started from two copies of x it leaves a = (x.lo | x.lo) and b = (x.hi | x.hi), so op(a, b) is
// the combination in every lane, with no LDS round trip.  (Inline asm with its own wait states: the compiler
// builtin of the same name folds both results into one register on ROCm 7.2.)
__device__ __forceinline__ float half_max(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ float half_sum(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

__device__ void vsum_block(const AttnParams& p, int vb, unsigned char* smem) {
    // vb -> (b, kvh, chunk); 256 threads: 32 column quads x 8 row lanes.
    const int chunk = vb % VSUM_CHUNKS;
    const int kvh = (vb / VSUM_CHUNKS) % p.Hkv;
    const int b = vb / (VSUM_CHUNKS * p.Hkv);
    const int32_t* desc = p.btab ? p.btab + p.btab[p.btab[BT_OFF_DESC] + b] : p.desc;
    if (desc[H_NUNIFORM] == 0) return;  // this sequence has no row that needs mean(V)
    const int L = desc[H_L];
    const int rows_per = (L + VSUM_CHUNKS - 1) / VSUM_CHUNKS;
    const int r0 = chunk * rows_per, r1 = min(L, r0 + rows_per);
    const int cq = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const unsigned short* base = p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh + cq * 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int r = r0 + rl; r < r1; r += 8) {
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
        const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(base + (int64_t)r * p.v_sl));  // read once
        s0 += __uint_as_float(w.x << 16);
        s1 += __uint_as_float(w.x & 0xffff0000u);
        s2 += __uint_as_float(w.y << 16);
        s3 += __uint_as_float(w.y & 0xffff0000u);
    }
    float* red = reinterpret_cast<float*>(smem);  // [8][128]
    red[rl * 128 + cq * 4 + 0] = s0;
    red[rl * 128 + cq * 4 + 1] = s1;
    red[rl * 128 + cq * 4 + 2] = s2;
    red[rl * 128 + cq * 4 + 3] = s3;
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += red[i * 128 + threadIdx.x];
        p.vsum[((int64_t)(b * p.Hkv + kvh) * VSUM_CHUNKS + chunk) * 128 + threadIdx.x] = t;
    }
}

#define AS4 __attribute__((address_space(4)))
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) unsigned int u32x16;

// 0 where the lane's bit of the 64-bit lane mask (words 2 j, 2 j + 1 of m) is set, -inf where it is clear: one
// v_cndmask with the mask in a scalar register pair.
__device__ __forceinline__ float mask_init(const u32x16& m, int j) {
    const uint64_t w = ((uint64_t)m[2 * j + 1] << 32) | m[2 * j];
    return __builtin_amdgcn_inverse_ballot_w64(w) ? 0.f : -INFINITY;
}

// One persistent workgroup (4 waves) per (work stream, key/value head slot): it walks the stream's steps, each a
// 32-row query block x one 64-key tile.  The steps of one unit (a row block and a run of its tiles) follow each
// other; at a unit's last step the next unit's Q rows are requested, and the K/V tile of the following step is
// always in flight, so the load -> compute chain is only exposed once per workgroup, not once per unit.
__global__ __launch_bounds__(256, 2) void sdag_attn_main(const AttnParams p) {
    // K0 K1 V0 V1 tile images, then two words of V-range tags (one per V buffer)
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES + 16];

    const int bid = blockIdx.x;
    if (bid >= p.n_main_blocks) {
        vsum_block(p, bid - p.n_main_blocks, smem);
        return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_hslots = p.Hkv * p.HG;
    const int hslot = bid % n_hslots;
    const int rest = bid / n_hslots;
    const int strm = rest % p.n_streams;
    const int bg = rest / p.n_streams;  // batch element when one descriptor serves the whole batch, else 0
    // The table is read-only for the whole launch and every index into it is wave-uniform: scalar loads.
    const AS4 int32_t* tab = (const AS4 int32_t*)p.tab;
    const int s_begin = tab[p.off_streams + strm], s_end = tab[p.off_streams + strm + 1];
    if (s_begin >= s_end) return;
    const AS4 i32x8* steps = (const AS4 i32x8*)(tab + p.off_steps);

    const int kvh = hslot % p.Hkv;
    const int hg = hslot / p.Hkv;
    const int hin = hg * 4 + wave;
    const bool head_ok = hin < p.G;
    const int hq = kvh * p.G + (head_ok ? hin : p.G - 1);
    unsigned* vtag = reinterpret_cast<unsigned*>(smem + 4 * TILE_BYTES);
    // LDS byte address of smem (what m0 takes for LDS-DMA)
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    if (tid < 2) vtag[tid] = 0;

    const int r = lane & 31;  // query row inside the block (S^T / O^T column)
    const int h = lane >> 5;

    // ---- Q fragments: B operand of S^T = K Q^T; lane (r, h) holds Q[row][16 ks + 8 h .. +7].
    bf16x8 qf[8];
    // Buffer loads: one descriptor per unit over this head's slice of sequence b (wave-uniform), a 32-bit offset per
    // lane (its row, its half of every 32-byte piece) and the piece as the instruction's immediate.  Every instruction
    // gathers 32 bytes from each of 32 rows: used once per workgroup, for its first unit.
    auto load_q = [&](int b, int row0, int L) {
        const unsigned qrow = (unsigned)p.q_sl * 2u;
        const auto qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.q + (int64_t)b * p.q_sb + (int64_t)hq * p.q_sh), 0,
                                                           (int)((unsigned)(L - 1) * qrow + (unsigned)(D * 2)), 0x00020000);
        const unsigned qoff = (unsigned)min(row0 + r, L - 1) * qrow + 16u * (unsigned)h;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(qrs, qoff + 32u * ks, 0, 0));
    };
    // The rows of every later unit arrive as WHOLE rows (lane -> row 4 j + (lane >> 4), 16-byte chunk lane & 15: four
    // 256-byte rows per instruction instead of 32 pieces of 32 bytes), raw in the registers of the fragments, which
    // the unit that ends no longer needs; rows past L read as zeros (the descriptor ends with the sequence's last
    // row).  q_fragments() then turns them into the fragments through the wave's own pieces of the tile buffer that
    // was just consumed (the image O is staged in at a unit's end): eight 16-byte writes, eight reads, no barrier.
    auto load_q_rows = [&](int b, int row0, int L) {
        const unsigned qrow = (unsigned)p.q_sl * 2u;
        const auto qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.q + (int64_t)b * p.q_sb + (int64_t)hq * p.q_sh), 0,
                                                           (int)((unsigned)(L - 1) * qrow + (unsigned)(D * 2)), 0x00020000);
        const unsigned qoff = (unsigned)(row0 + (lane >> 4)) * qrow + 16u * (unsigned)(lane & 15);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            qf[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(qrs, qoff + (unsigned)(4 * j) * qrow, 0, 0));
    };
    // ---- staging geometry: thread covers chunk (tid & 15) of rows (tid >> 4) + 16 i, i = 0..3.
    const int st_row = tid >> 4, st_ch = tid & 15;
    const int st_off = tile_off(st_row, st_ch);  // + 16 * 256 * i for the other rows (swizzle term unchanged)
    u32x4 vreg[4];
    // Next-tile rows: K goes straight from global memory into the LDS image (LDS-DMA, one 1 KiB piece = 4 tile rows
    // per wave instruction, the swizzle applied through the source address); V travels global -> registers -> LDS
    // because it is converted on the way.  All are buffer loads whose descriptor starts at the tile and ends with
    // the sequence's last row, so rows past L read as zeros and no lane needs its own clamp; the per-lane offset
    // (row inside the tile, 16-byte chunk) is the same for every tile.
    const int kd_ch = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (wave & 3));  // source chunk of this lane's LDS slot
    const unsigned koff0 = (unsigned)st_row * (unsigned)p.k_sl * 2u + (unsigned)kd_ch * 16u;
    const unsigned voff0 = (unsigned)st_row * (unsigned)p.v_sl * 2u + (unsigned)st_ch * 16u;
    const unsigned kstep = 16u * (unsigned)p.k_sl * 2u, vstep = 16u * (unsigned)p.v_sl * 2u;
    // Start of this key/value head's slice of sequence b_kv: recomputed only when a step belongs to another sequence
    // than the one before (the 64-bit products stay out of the per-step path).
    int b_kv = -1;
    const unsigned char* kbase = nullptr;
    const unsigned char* vbase = nullptr;
    auto set_kv_base = [&](int b) {
        b_kv = b;
        kbase = reinterpret_cast<const unsigned char*>(p.k + (int64_t)b * p.k_sb + (int64_t)kvh * p.k_sh);
        vbase = reinterpret_cast<const unsigned char*>(p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh);
    };
    // Buffer descriptor from the tile's first key to the end of row L - 1 of the slice (below 2^31 bytes: checked
    // on the host).
    auto tile_rsrc = [&](const unsigned char* base, unsigned row_bytes, int key0, int L) {
        const unsigned off = (unsigned)key0 * row_bytes;
        const unsigned bytes = (unsigned)(L - 1) * row_bytes + (unsigned)(D * 2) - off;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base + off), 0, (int)bytes, 0x00020000);
    };
    // The two descriptors of a tile are made apart from the loads (SDAG_TILE_RSRC, with its branch on a change of
    // sequence), so that the loads can sit in one basic block with the MFMAs they are interleaved with.
#define SDAG_TILE_RSRC(B_, KEY0_, L_)                                                                         \
    if ((B_) != b_kv) set_kv_base(B_);                                                                        \
    const auto krs = tile_rsrc(kbase, (unsigned)p.k_sl * 2u, (KEY0_), (L_));                                  \
    const auto vrs = tile_rsrc(vbase, (unsigned)p.v_sl * 2u, (KEY0_), (L_));
    // one quarter of a tile's V rows (16 rows) into registers
#define SDAG_V_LOAD(I_) \
    vreg[I_] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrs, voff0 + (I_) * vstep, 0, 0))
    // one of this wave's four 1 KiB pieces of a tile's K rows into LDS.  (asm: behind the builtin the compiler orders
    // every later LDS read after these writes with a full vmcnt(0), in the middle of the step; the wait that covers
    // them is the one at the step's end)
#define SDAG_K_DMA(BUF_, I_)                                                                                  \
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"                                 \
                 :: "s"((unsigned)((BUF_) * TILE_BYTES + (4 * (I_) + wave) * 1024) + smem_base),               \
                    "v"(koff0 + (I_) * kstep), "s"(krs) : "memory", "m0")
#define SDAG_STAGE_LOADS(BUF_)                                                                                \
    {                                                                                                         \
        SDAG_V_LOAD(0); SDAG_V_LOAD(1); SDAG_V_LOAD(2); SDAG_V_LOAD(3);                                       \
        SDAG_K_DMA(BUF_, 0); SDAG_K_DMA(BUF_, 1); SDAG_K_DMA(BUF_, 2); SDAG_K_DMA(BUF_, 3);                   \
    }
    // V goes to LDS as fp16: P V then runs on the f16 MFMA with 11-bit probabilities instead of 8-bit ones.
    // bf16 -> fp16 is exact for 2^-14 <= |v| < 2^16; round-toward-zero conversion saturates larger values
    // instead of producing infinities, and a tag tells the consumers to redo such a tile with a scale.
    float vmax = 0.f;
#define SDAG_STAGE_STORE(BUF_)                                                                                \
    {                                                                                                         \
        unsigned char* vd = smem + (2 + (BUF_)) * TILE_BYTES + st_off;                                        \
        vmax = 0.f;                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
            u32x4 hv;                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
                const unsigned w = vreg[i][j];                                                                \
                const float f0 = __uint_as_float(w << 16), f1 = __uint_as_float(w & 0xffff0000u);            \
                vmax = max3_abs(vmax, f0, f1);                                                                \
                hv[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(f0, f1));                     \
            }                                                                                                 \
            *reinterpret_cast<u32x4*>(vd + i * 16 * 256) = hv;                                                \
        }                                                                                                     \
    }
    // exponent field >= 143 means |v| >= 2^16, beyond fp16
    auto stage_tag = [&](int buf, unsigned seq) {
        const unsigned eb = (__float_as_uint(vmax) >> 23) & 0xffu;
        if (eb >= 143u) atomicMax(&vtag[buf], (seq << 8) | eb);
    };
    // Slow path for a V tile with values beyond the fp16 range: every thread reloads its part of the tile and
    // stores it scaled by 2^-e (the probabilities are scaled by 2^e to compensate).
    auto restage_v_scaled = [&](int buf, int b, int key0, int L, int e) {
        const float sc = __uint_as_float((unsigned)(127 - e) << 23);
        unsigned char* vd = smem + (2 + buf) * TILE_BYTES + st_off;
        const char* vh = reinterpret_cast<const char*>(p.v + (int64_t)b * p.v_sb + (int64_t)kvh * p.v_sh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kr = min(key0 + st_row + 16 * i, L - 1);
            const u32x4 raw = *reinterpret_cast<const u32x4*>(vh + ((int64_t)kr * p.v_sl + st_ch * 8) * 2);
            u32x4 hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned w = raw[j];
                hv[j] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(w << 16) * sc, __uint_as_float(w & 0xffff0000u) * sc));
            }
            *reinterpret_cast<u32x4*>(vd + i * 16 * 256) = hv;
        }
    };

    // ---- K read geometry: row kb*32 + r, chunk (2 ks + h) ^ f(r)
    const int kf = ((r & 3) << 2) | ((r >> 2) & 3);
    const int k_row_off = r * 256;
    // ---- V transposed-read geometry (ds_read_b64_tr_b16 delivers, per 16-lane group, a 4-key x 16-column
    // block column-major: lane i of the group supplies the address of key (i >> 2), columns 4 (i & 3) .. +3,
    // and receives column i).
    const int g16 = (lane >> 4) & 1, li = lane & 15, vq = li >> 2, vp = li & 3;
    const int cl = 2 * g16 + (vp >> 1);
    int v_off[4][2];
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
            v_off[db][hf] = (4 * h + vq) * 256 + (((((db ^ vq) << 2) | (cl ^ (2 * hf + h)))) << 4) + 8 * (vp & 1);

    f32x16 oacc[4];
    float m_run = -INFINITY, l_run = 0.f;
    const float c = p.c;

    // Initial S^T accumulators of a step from its lane-mask image (table offset MOFF_; 0 = every pair allowed: the
    // accumulators are left alone and the step's first products start from a constant zero instead).
#define SDAG_INIT_SCORES(MOFF_)                                                                               \
    {                                                                                                         \
        const int moff_ = (MOFF_);                                                                            \
        if (moff_) {                                                                                          \
            const AS4 u32x16* mp = (const AS4 u32x16*)(tab + moff_);                                          \
            {                                                                                                 \
                const u32x16 ma = mp[0], mb = mp[1];                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                               \
                    s0[i] = mask_init(ma, i);                                                                 \
                    s0[8 + i] = mask_init(mb, i);                                                             \
                }                                                                                             \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0); /* the second half reuses the first half's scalar registers */ \
            {                                                                                                 \
                const u32x16 mc = mp[2], md = mp[3];                                                          \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                               \
                    s1[i] = mask_init(mc, i);                                                                 \
                    s1[8 + i] = mask_init(md, i);                                                             \
                }                                                                                             \
            }                                                                                                 \
        }                                                                                                     \
    }
    // ---- software pipeline over the stream's steps: the record of the NEXT step is known one iteration ahead,
    // so its K/V rows are requested before the current tile is computed, and the record after that is fetched
    // meanwhile.
    i32x8 cur = steps[s_begin];
    i32x8 nxt = steps[min(s_begin + 1, s_end - 1)];
    load_q(cur[S_B] + bg, cur[S_ROW0], cur[S_L]);
    __syncthreads();  // vtag zeroed
    {
        SDAG_TILE_RSRC(cur[S_B] + bg, cur[S_KEY0], cur[S_L]);
        SDAG_STAGE_LOADS(0);
    }
    SDAG_STAGE_STORE(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the K pieces have landed
    stage_tag(0, 1u);
    // The mask enters as the initial accumulator: 0 where the pair is allowed, -inf where it is not (a finite
    // product added to -inf stays -inf).  Accumulator register i of block kb is key 32 kb + (i & 3) + 8 (i >> 2)
    // + 4 h; the tile's lane masks are stored in that order.  Each step prepares the accumulators of the next one
    // once its own scores have been turned into probabilities.
    f32x16 s0, s1;
    SDAG_INIT_SCORES(cur[S_MOFF]);
    __syncthreads();

#ifdef SDAG_STAMPS
    unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
    unsigned seq = 1u;
#ifdef SDAG_WGTIMES
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wg_c0 = __builtin_amdgcn_s_memtime();
#endif
    // The loop body is written once (sdag_attn_step.inc) and expanded for either tile buffer.
    for (int it = s_begin;;) {
#define SDAG_BUFC 0
#include "sdag_attn_step.inc"
#undef SDAG_BUFC
        ++seq;
        if (++it >= s_end) break;
#define SDAG_BUFC 1
#include "sdag_attn_step.inc"
#undef SDAG_BUFC
        ++seq;
        if (++it >= s_end) break;
    }

#ifdef SDAG_WGTIMES
    if (tid == 0 && blockIdx.x < 8192) {
        g_wg_ticks[blockIdx.x][0] = (unsigned)(__builtin_amdgcn_s_memrealtime() - wg_t0);
        const unsigned long long wg_dc = __builtin_amdgcn_s_memtime() - wg_c0;  // shader-clock cycles
        g_wg_ticks[blockIdx.x][4] = (unsigned)wg_dc;
        g_wg_ticks[blockIdx.x][5] = (unsigned)(wg_dc >> 32);
        g_wg_ticks[blockIdx.x][1] = (unsigned)(s_end - s_begin);
        g_wg_ticks[blockIdx.x][2] = (unsigned)(wg_t0 & 0xffffffffu);
        // HW_ID bits 8-15 (CU, shader array, shader engine) and XCC_ID bits 0-3
        g_wg_ticks[blockIdx.x][3] = ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xffu) | ((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu) << 8);
    }
#endif
#ifdef SDAG_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        for (int i = 0; i < 8; ++i) g_stamps[i] = st_acc[i];
        g_stamps[15] = (unsigned long long)(s_end - s_begin);
    }
#endif
}

// Merge the partial results of split row blocks, and write mean(V) into rows that allow nothing.
// grid: (n_split * B * Hq) blocks of 256 threads for the merge, then ceil(n_uniform * B * Hq / 8) blocks for
// the uniform rows (8 (row, head) pairs per block, 32 threads x 4 columns each).
__global__ __launch_bounds__(256) void sdag_attn_combine(const AttnParams p, int n_merge_blocks) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < n_merge_blocks) {
        const int hq = blockIdx.x % p.Hq;
        const int rest = blockIdx.x / p.Hq;
        int b, e, slot_base;
        const int32_t* desc;
        if (p.btab) {
            const int32_t* bt = p.btab;
            const int32_t* mr = bt + bt[BT_OFF_MERGE] + 2 * rest;
            b = mr[0];
            e = mr[1];
            desc = bt + bt[bt[BT_OFF_DESC] + b];
            slot_base = bt[bt[BT_OFF_SLOTBASE] + b];
        } else {
            b = rest % p.B;
            e = rest / p.B;
            desc = p.desc;
            slot_base = b * p.n_slots;
        }
        const int32_t* sp = desc + desc[H_OFF_SPLIT] + e * 4;
        const int first = slot_base + sp[1], n = sp[2];
        // thread -> row (tid >> 3), 16 columns starting at (tid & 7) * 16
        const int r = tid >> 3, c0 = (tid & 7) * 16;
        const int row = sp[0] + r;
        if (r >= sp[3]) return;
        float mmax = -INFINITY;
        for (int s = 0; s < n; ++s) {
            const int64_t prow = ((int64_t)(first + s) * p.Hq + hq) * RB + r;
            mmax = fmaxf(mmax, p.part_ml[prow * 2]);
        }
        if (mmax == -INFINITY) return;  // a row that allows nothing: handled below
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        float den = 0.f;
        for (int s = 0; s < n; ++s) {
            const int64_t prow = ((int64_t)(first + s) * p.Hq + hq) * RB + r;
            const float ms = p.part_ml[prow * 2];
            const float w = fast_exp2((ms - mmax) * p.c);  // ms = -inf -> 0
            den += w * p.part_ml[prow * 2 + 1];
            const f32x4* po = reinterpret_cast<const f32x4*>(p.part_o + prow * D + c0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 x = po[i];
                acc[4 * i] += w * x[0];
                acc[4 * i + 1] += w * x[1];
                acc[4 * i + 2] += w * x[2];
                acc[4 * i + 3] += w * x[3];
            }
        }
        const float inv = 1.0f / den;
        const int64_t obase = (int64_t)b * p.o_sb + (int64_t)row * p.o_sl + (int64_t)hq * p.o_sh + c0;
        if (p.out_f32) {
            float* op = reinterpret_cast<float*>(p.o) + obase;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 w = {acc[4 * i] * inv, acc[4 * i + 1] * inv, acc[4 * i + 2] * inv, acc[4 * i + 3] * inv};
                reinterpret_cast<f32x4*>(op)[i] = w;
            }
        } else {
            unsigned short* op = reinterpret_cast<unsigned short*>(p.o) + obase;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bf16x8 w;
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = (__bf16)(acc[8 * i + j] * inv);
                reinterpret_cast<bf16x8*>(op)[i] = w;
            }
        }
        return;
    }
    // ---- rows that allow nothing: softmax over all-equal scores = 1/L on every key -> mean of V over all L keys
    const int64_t pair = (int64_t)(blockIdx.x - n_merge_blocks) * 8 + (tid >> 5);
    int b, row, L;
    const int hq = pair % p.Hq;
    const int64_t rest = pair / p.Hq;
    if (p.btab) {
        const int32_t* bt = p.btab;
        if (rest >= bt[BT_NUNIFORM]) return;
        const int32_t* ur = bt + bt[BT_OFF_UNIFORM] + 2 * rest;
        b = ur[0];
        row = ur[1];
        L = bt[bt[bt[BT_OFF_DESC] + b] + H_L];
    } else {
        const int32_t* desc = p.desc;
        if (rest >= (int64_t)desc[H_NUNIFORM] * p.B) return;
        b = rest % p.B;
        row = desc[desc[H_OFF_UNIFORM] + rest / p.B];
        L = desc[H_L];
    }
    const int kvh = hq / p.G;
    const int c0 = (tid & 31) * 4;
    const float* vs = p.vsum + (int64_t)(b * p.Hkv + kvh) * VSUM_CHUNKS * 128 + c0;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < VSUM_CHUNKS; ++ch) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(vs + ch * 128);
        t += x;
    }
    const float invL = 1.0f / (float)L;
    const int64_t obase = (int64_t)b * p.o_sb + (int64_t)row * p.o_sl + (int64_t)hq * p.o_sh + c0;
    if (p.out_f32) {
        f32x4 w = {t[0] * invL, t[1] * invL, t[2] * invL, t[3] * invL};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.o) + obase) = w;
    } else {
        bf16x4 w = {(__bf16)(t[0] * invL), (__bf16)(t[1] * invL), (__bf16)(t[2] * invL), (__bf16)(t[3] * invL)};
        *reinterpret_cast<bf16x4*>(reinterpret_cast<unsigned short*>(p.o) + obase) = w;
    }
}

// Dense uint8 [L, L] image of a descriptor: one block per row block; zero the 32 rows, then paint every item.
__global__ __launch_bounds__(256) void sdag_mask_expand(const int32_t* desc, unsigned char* out) {
    const int L = desc[H_L];
    const int rb = blockIdx.x;
    const int32_t* rbinfo = desc + desc[H_OFF_RBINFO];
    const int r0 = rbinfo[2 * rb + 1], rows = (rb + 1 < desc[H_NRB] ? rbinfo[2 * rb + 3] : L) - r0;
    const int tid = threadIdx.x;
    for (int64_t i = tid; i < (int64_t)rows * L; i += 256) out[(int64_t)r0 * L + i] = 0;
    __syncthreads();
    const int n_units = desc[H_NUNITS];
    const int32_t* units = desc + desc[H_OFF_UNITS];
    const int32_t* items = desc + desc[H_OFF_ITEMS];
    const uint64_t* masks = reinterpret_cast<const uint64_t*>(desc + desc[H_OFF_MASKS]);
    for (int u = 0; u < n_units; ++u) {
        if (units[u * UNIT_WORDS] != rb) continue;
        for (int it = units[u * UNIT_WORDS + 1]; it < units[u * UNIT_WORDS + 2]; ++it) {
            const int key0 = items[it * 2], mi = items[it * 2 + 1];
            for (int e = tid; e < RB * KT; e += 256) {
                const int r = e >> 6, cidx = e & 63;
                const int col = key0 + cidx;
                if (r >= rows || col >= L) continue;
                const bool on = mi < 0 ? true : ((masks[(int64_t)mi * RB + r] >> cidx) & 1ull);
                if (on) out[(int64_t)(r0 + r) * L + col] = 1;
            }
        }
    }
}

}  // namespace
}  // namespace sdag

using namespace sdag;

namespace {

int64_t workspace_bytes(int64_t n_slots_total, int64_t n_uniform_total, int B, int Hq, int Hkv) {
    int64_t bytes = n_slots_total * Hq * RB * (D + 2) * (int64_t)sizeof(float);
    if (n_uniform_total > 0) bytes += (int64_t)B * Hkv * VSUM_CHUNKS * 128 * sizeof(float);
    return (bytes + 255) & ~255ll;
}

struct Tensors {
    const void *q, *k, *v;
    void* o;
    int64_t q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sl, o_sh;
};

int check_common(const Tensors& t, int B, int Hq, int Hkv, int L, int Dh, float scale, int out_dtype) {
    if (Dh != D) return fail(SDAG_ERR_ARG, "head_dim %d not supported (only 128)", Dh);
    if (B <= 0 || Hq <= 0 || Hkv <= 0 || L <= 0 || Hq % Hkv) return fail(SDAG_ERR_ARG, "bad B/Hq/Hkv/L");
    if (!(scale > 0.f)) return fail(SDAG_ERR_ARG, "scale must be positive");
    if (out_dtype != SDAG_OUT_BF16 && out_dtype != SDAG_OUT_F32) return fail(SDAG_ERR_ARG, "bad out_dtype");
    if (!t.q || !t.k || !t.v || !t.o) return fail(SDAG_ERR_ARG, "NULL tensor pointer");
    const int64_t strides[] = {t.q_sb, t.q_sh, t.q_sl, t.k_sb, t.k_sh, t.k_sl, t.v_sb, t.v_sh, t.v_sl, t.o_sb, t.o_sl, t.o_sh};
    for (int64_t st : strides)
        if (st < 0 || (st & 7) || st > 0x7fffffff) return fail(SDAG_ERR_ARG, "strides must be non-negative multiples of 8 elements below 2^31");
    if ((int64_t)L * t.q_sl * 2 >= (1ll << 32) || (int64_t)L * t.k_sl * 2 >= (1ll << 32) || (int64_t)L * t.v_sl * 2 >= (1ll << 32))
        return fail(SDAG_ERR_ARG, "one (batch, head) slice of Q, K or V must span less than 4 GiB");
    if ((int64_t)L * t.o_sl * 2 >= (1ll << 31)) return fail(SDAG_ERR_ARG, "one (batch, head) slice of O must span less than 2 GiB");
    if (((uintptr_t)t.q | (uintptr_t)t.k | (uintptr_t)t.v | (uintptr_t)t.o) & 15) return fail(SDAG_ERR_ARG, "tensor pointers must be 16-byte aligned");
    return SDAG_OK;
}

// Launch main (+ column-sum blocks) and, if anything needs it, the combine kernel.
int launch(AttnParams& p, const Tensors& t, int B, int Hq, int Hkv, float scale, int out_dtype, int64_t n_stream_blocks,
           int64_t n_slots_total, int64_t n_merge_entries, int64_t n_uniform_total, void* workspace, hipStream_t st) {
    p.q = (const unsigned short*)t.q; p.k = (const unsigned short*)t.k; p.v = (const unsigned short*)t.v; p.o = t.o;
    p.q_sb = (int)t.q_sb; p.q_sh = (int)t.q_sh; p.q_sl = (int)t.q_sl;
    p.k_sb = (int)t.k_sb; p.k_sh = (int)t.k_sh; p.k_sl = (int)t.k_sl;
    p.v_sb = (int)t.v_sb; p.v_sh = (int)t.v_sh; p.v_sl = (int)t.v_sl;
    p.o_sb = (int)t.o_sb; p.o_sl = (int)t.o_sl; p.o_sh = (int)t.o_sh;
    p.B = B; p.Hq = Hq; p.Hkv = Hkv;
    p.G = Hq / Hkv;
    p.HG = (p.G + 3) / 4;
    p.out_f32 = out_dtype == SDAG_OUT_F32;
    p.c = scale * 1.4426950408889634f;
    float* ws = (float*)workspace;
    p.part_o = ws;
    p.part_ml = ws + n_slots_total * Hq * RB * D;
    p.vsum = p.part_ml + n_slots_total * Hq * RB * 2;
    const int64_t main_blocks = n_stream_blocks * Hkv * p.HG;
    const int64_t vsum_blocks = n_uniform_total > 0 ? (int64_t)B * Hkv * VSUM_CHUNKS : 0;
    if (main_blocks + vsum_blocks > 0x7fffffff) return fail(SDAG_ERR_ARG, "grid too large");
    p.n_main_blocks = (int)main_blocks;
    if (main_blocks + vsum_blocks > 0) {
        hipLaunchKernelGGL(sdag_attn_main, dim3((unsigned)(main_blocks + vsum_blocks)), dim3(256), 0, st, p);
        SDAG_HIP_CHECK(hipGetLastError());
    }
    const int64_t n_merge = n_merge_entries * Hq;
    const int64_t cblocks = n_merge + (n_uniform_total * Hq + 7) / 8;
    if (cblocks > 0x7fffffff) return fail(SDAG_ERR_ARG, "grid too large");
    if (cblocks > 0) {
        hipLaunchKernelGGL(sdag_attn_combine, dim3((unsigned)cblocks), dim3(256), 0, st, p, (int)n_merge);
        SDAG_HIP_CHECK(hipGetLastError());
    }
    return SDAG_OK;
}

}  // namespace

extern "C" int64_t sdag_attn_workspace_bytes(const int32_t* d, int B, int Hq, int Hkv) {
    if (!desc_ok(d) || B <= 0 || Hq <= 0 || Hkv <= 0) return -1;
    return workspace_bytes((int64_t)d[H_NSLOTS] * B, (int64_t)d[H_NUNIFORM] * B, B, Hq, Hkv);
}

extern "C" int sdag_attn_fwd(const void* q, const void* k, const void* v, void* o,
                             int64_t q_sb, int64_t q_sh, int64_t q_sl,
                             int64_t k_sb, int64_t k_sh, int64_t k_sl,
                             int64_t v_sb, int64_t v_sh, int64_t v_sl,
                             int64_t o_sb, int64_t o_sl, int64_t o_sh,
                             int B, int Hq, int Hkv, int L, int Dh, float scale,
                             const int32_t* desc_host, const int32_t* desc_dev,
                             void* workspace, int64_t workspace_bytes_, int out_dtype, void* stream) {
    const Tensors t = {q, k, v, o, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sl, o_sh};
    if (int rc = check_common(t, B, Hq, Hkv, L, Dh, scale, out_dtype)) return rc;
    if (!desc_ok(desc_host)) return fail(SDAG_ERR_DESC, "desc_host is not a descriptor");
    if (!desc_dev) return fail(SDAG_ERR_ARG, "desc_dev is NULL");
    if (desc_host[H_L] != L) return fail(SDAG_ERR_DESC, "descriptor is for L=%d, call has L=%d", desc_host[H_L], L);
    const int64_t need = sdag_attn_workspace_bytes(desc_host, B, Hq, Hkv);
    if (need > 0 && (!workspace || workspace_bytes_ < need)) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)need);
    AttnParams p;
    p.desc = desc_dev;
    p.btab = nullptr;
    p.tab = desc_dev;
    p.n_streams = desc_host[H_NSTREAMS];
    p.off_streams = desc_host[H_OFF_STREAMS];
    p.off_steps = desc_host[H_OFF_STEPS];
    p.n_slots = desc_host[H_NSLOTS];
    p.slots_per_b = p.n_slots;
    return launch(p, t, B, Hq, Hkv, scale, out_dtype, desc_host[H_NSTEPS] ? (int64_t)p.n_streams * B : 0, (int64_t)p.n_slots * B,
                  (int64_t)desc_host[H_NSPLIT] * B, (int64_t)desc_host[H_NUNIFORM] * B, workspace, (hipStream_t)stream);
}

namespace {
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];  // address of `scratch` on each device, looked up on the first call there
std::mutex g_scratch_mu;

// Address of the current device's copy of `scratch`.
int static_scratch(void** out) {
    int dev = -1;
    SDAG_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(SDAG_ERR_ARG, "device %d outside [0, %d)", dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        SDAG_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(SDAG_ERR_HIP, "hipGetSymbolAddress returned NULL");
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return SDAG_OK;
}
}  // namespace

extern "C" int sdag_attn_fwd_static(const void* q, const void* k, const void* v, void* o,
                                    int64_t q_sb, int64_t q_sh, int64_t q_sl,
                                    int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                    int64_t v_sb, int64_t v_sh, int64_t v_sl,
                                    int64_t o_sb, int64_t o_sl, int64_t o_sh,
                                    int B, int Hq, int Hkv, int L, int Dh, float scale,
                                    const int32_t* desc_host, const int32_t* desc_dev, int out_dtype, void* stream) {
    if (!desc_ok(desc_host)) return fail(SDAG_ERR_DESC, "desc_host is not a descriptor");
    if (B <= 0 || Hq <= 0 || Hkv <= 0) return fail(SDAG_ERR_ARG, "bad B/Hq/Hkv/L");
    const int64_t need = sdag_attn_workspace_bytes(desc_host, B, Hq, Hkv);
    if (need > (int64_t)SDAG_STATIC_SCRATCH_BYTES)
        return fail(SDAG_ERR_CAPACITY, "this call needs %lld bytes of scratch, sdag_attn_fwd_static holds %lld: use sdag_attn_fwd with a workspace",
                    (long long)need, (long long)SDAG_STATIC_SCRATCH_BYTES);
    void* ws = nullptr;
    if (int rc = static_scratch(&ws)) return rc;
    return sdag_attn_fwd(q, k, v, o, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sl, o_sh, B, Hq, Hkv, L, Dh, scale,
                         desc_host, desc_dev, ws, (int64_t)SDAG_STATIC_SCRATCH_BYTES, out_dtype, stream);
}

extern "C" int64_t sdag_attn_batch_workspace_bytes(const int32_t* bt, int Hq, int Hkv) {
    if (!batch_ok(bt) || Hq <= 0 || Hkv <= 0) return -1;
    return workspace_bytes(bt[BT_NSLOTS], bt[BT_NUNIFORM], bt[BT_B], Hq, Hkv);
}

extern "C" int sdag_attn_fwd_batch(const void* q, const void* k, const void* v, void* o,
                                   int64_t q_sb, int64_t q_sh, int64_t q_sl,
                                   int64_t k_sb, int64_t k_sh, int64_t k_sl,
                                   int64_t v_sb, int64_t v_sh, int64_t v_sl,
                                   int64_t o_sb, int64_t o_sl, int64_t o_sh,
                                   int B, int Hq, int Hkv, int Lmax, int Dh, float scale,
                                   const int32_t* batch_host, const int32_t* batch_dev,
                                   void* workspace, int64_t workspace_bytes_, int out_dtype, void* stream) {
    const Tensors t = {q, k, v, o, q_sb, q_sh, q_sl, k_sb, k_sh, k_sl, v_sb, v_sh, v_sl, o_sb, o_sl, o_sh};
    if (int rc = check_common(t, B, Hq, Hkv, Lmax, Dh, scale, out_dtype)) return rc;
    if (!batch_ok(batch_host)) return fail(SDAG_ERR_DESC, "batch_host is not a batch table");
    if (!batch_dev) return fail(SDAG_ERR_ARG, "batch_dev is NULL");
    if (batch_host[BT_B] != B) return fail(SDAG_ERR_DESC, "batch table holds %d sequences, call has B=%d", batch_host[BT_B], B);
    if (batch_host[BT_LMAX] > Lmax) return fail(SDAG_ERR_DESC, "a sequence of the batch has L=%d > Lmax=%d", batch_host[BT_LMAX], Lmax);
    const int64_t need = sdag_attn_batch_workspace_bytes(batch_host, Hq, Hkv);
    if (need > 0 && (!workspace || workspace_bytes_ < need)) return fail(SDAG_ERR_CAPACITY, "workspace needs %lld bytes", (long long)need);
    AttnParams p;
    p.desc = nullptr;
    p.btab = batch_dev;
    p.tab = batch_dev;
    p.n_streams = batch_host[BT_NSTREAMS];
    p.off_streams = batch_host[BT_OFF_STREAMS];
    p.off_steps = batch_host[BT_OFF_STEPS];
    p.n_slots = 0;
    p.slots_per_b = 0;
    return launch(p, t, B, Hq, Hkv, scale, out_dtype, batch_host[BT_NSTEPS] ? batch_host[BT_NSTREAMS] : 0, batch_host[BT_NSLOTS], batch_host[BT_NMERGE],
                  batch_host[BT_NUNIFORM], workspace, (hipStream_t)stream);
}

extern "C" int sdag_mask_materialize(const int32_t* desc_host, const int32_t* desc_dev, uint8_t* out_dev, void* stream) {
    if (!desc_ok(desc_host)) return fail(SDAG_ERR_DESC, "desc_host is not a descriptor");
    if (!desc_dev || !out_dev) return fail(SDAG_ERR_ARG, "NULL pointer");
    const int nrb = desc_host[H_NRB];
    if (nrb == 0) return SDAG_OK;
    hipLaunchKernelGGL(sdag_mask_expand, dim3(nrb), dim3(256), 0, (hipStream_t)stream, desc_dev, out_dev);
    SDAG_HIP_CHECK(hipGetLastError());
    return SDAG_OK;
}

#ifdef SDAG_WGTIMES
extern "C" int sdag_debug_wg_ticks(unsigned* out) {
    SDAG_HIP_CHECK(hipDeviceSynchronize());
    SDAG_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sdag::g_wg_ticks), sizeof(unsigned) * 8192 * 6));
    return SDAG_OK;
}
#endif
#ifdef SDAG_STAMPS
extern "C" int sdag_debug_stamps(unsigned long long* out) {
    SDAG_HIP_CHECK(hipDeviceSynchronize());
    SDAG_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(sdag::g_stamps), sizeof(unsigned long long) * 16));
    return SDAG_OK;
}
#endif
