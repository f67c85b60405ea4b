// Issue-rate microbenchmarks for gfx950: cycles per instruction of the vector, matrix and LDS instructions the
// attention kernel is made of, at 1, 2 and 4 waves per SIMD, alone and interleaved.  Diagnostic only.
//   hipcc -O3 --offload-arch=gfx950 issue_rates.hip -o issue_rates && ./issue_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP32(x) REP16(x) REP16(x)

constexpr int ITER = 200;

struct Out { long long cyc; long long wall; };

template <int T>
__global__ __launch_bounds__(1024) void k(Out* out, int iters) {
    __shared__ float lds[8192];
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 ka = {a0, a1, a2, a3}, kb = {a4, a5, a6, a7};
    lds[threadIdx.x] = a0;
    unsigned la = (threadIdx.x & 63) * 16;
    f32x4 r0 = {}, r1 = {}, q0 = {}, q1 = {}, q2 = {}, q3 = {};
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, d0 = {}, d1 = {};
    unsigned long long msk = 0x5555555555555555ull + blockIdx.x;
    __syncthreads();
    long long w0 = wall_clock64();
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (T == 0) {  // 32 independent v_fma_f32 (8 chains)
            asm volatile(REP4("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                              "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 1) {  // 32 v_exp_f32
            asm volatile(REP4("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                              "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 2) {  // 32 v_cndmask with an SGPR-pair mask
            asm volatile(REP4("v_cndmask_b32 %0, %1, %2, %8\n v_cndmask_b32 %1, %2, %3, %8\n v_cndmask_b32 %2, %3, %4, %8\n v_cndmask_b32 %3, %4, %5, %8\n"
                              "v_cndmask_b32 %4, %5, %6, %8\n v_cndmask_b32 %5, %6, %7, %8\n v_cndmask_b32 %6, %7, %0, %8\n v_cndmask_b32 %7, %0, %1, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(msk));
        } else if constexpr (T == 3) {  // 32 v_cvt_pkrtz_f16_f32
            asm volatile(REP4("v_cvt_pkrtz_f16_f32 %0, %1, %2\n v_cvt_pkrtz_f16_f32 %1, %2, %3\n v_cvt_pkrtz_f16_f32 %2, %3, %4\n v_cvt_pkrtz_f16_f32 %3, %4, %5\n"
                              "v_cvt_pkrtz_f16_f32 %4, %5, %6\n v_cvt_pkrtz_f16_f32 %5, %6, %7\n v_cvt_pkrtz_f16_f32 %6, %7, %0\n v_cvt_pkrtz_f16_f32 %7, %0, %1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 4) {  // 32 v_max3_f32
            asm volatile(REP4("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %4\n v_max3_f32 %3, %3, %4, %5\n"
                              "v_max3_f32 %4, %4, %5, %6\n v_max3_f32 %5, %5, %6, %7\n v_max3_f32 %6, %6, %7, %0\n v_max3_f32 %7, %7, %0, %1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 5) {  // 32 v_pk_mul_f32 (two floats per instruction)
            asm volatile(REP8("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        } else if constexpr (T == 6) {  // 32 v_mov_b32
            asm volatile(REP4("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 7) {  // 32 independent MFMA 32x32x16 bf16 (4 accumulators)
            asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n"
                              "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n")
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(ka), "v"(kb));
        } else if constexpr (T == 8) {  // 32 MFMA on ONE accumulator (dependent chain, as in the kernel)
            asm volatile(REP32("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n") : "+v"(c0) : "v"(ka), "v"(kb));
        } else if constexpr (T == 9) {  // 2 chains, alternating
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n")
                         : "+v"(c0), "+v"(c1) : "v"(ka), "v"(kb));
        } else if constexpr (T == 10) {  // 16 x (MFMA on 2 chains + 4 fma): are the vector ops free under the MFMAs?
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                         : "+v"(c0), "+v"(c1), "+v"(ka), "+v"(kb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if constexpr (T == 11) {  // 16 x (MFMA + 8 fma)
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                               "v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f32 %11, %11, %11, %11\n")
                         : "+v"(c0), "+v"(c1), "+v"(ka), "+v"(kb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 12) {  // 16 x (MFMA + 16 fma)
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n"
                               "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                               "v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f32 %11, %11, %11, %11\n"
                               "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                               "v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f32 %11, %11, %11, %11\n")
                         : "+v"(c0), "+v"(c1), "+v"(ka), "+v"(kb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 13) {  // 32 ds_read_b128
            asm volatile(REP16("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(r0), "=&v"(r1) : "v"(la) : "memory");
        } else if constexpr (T == 14) {  // 32 ds_read_b64_tr_b16
            asm volatile(REP16("ds_read_b64_tr_b16 %0, %2\n ds_read_b64_tr_b16 %1, %2 offset:4096\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(d0), "=&v"(d1) : "v"(la / 2) : "memory");
        } else if constexpr (T == 15) {  // 16 x (MFMA + 2 ds_read_b64_tr + 4 fma): a P V-like slot
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n ds_read_b64_tr_b16 %12, %14\n ds_read_b64_tr_b16 %13, %14 offset:4096\n"
                               "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n") "s_waitcnt lgkmcnt(0)\n"
                         : "+v"(c0), "+v"(c1), "+v"(ka), "+v"(kb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                           "=&v"(d0), "=&v"(d1) : "v"(la / 2) : "memory");
        } else if constexpr (T == 16) {  // 32 v_dot2c_f32_f16
            asm volatile(REP4("v_dot2c_f32_f16 %0, %1, %2\n v_dot2c_f32_f16 %1, %2, %3\n v_dot2c_f32_f16 %2, %3, %4\n v_dot2c_f32_f16 %3, %4, %5\n"
                              "v_dot2c_f32_f16 %4, %5, %6\n v_dot2c_f32_f16 %5, %6, %7\n v_dot2c_f32_f16 %6, %7, %0\n v_dot2c_f32_f16 %7, %0, %1\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (T == 17) {  // 32 v_pk_fma_f32
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n v_pk_fma_f32 %3, %3, %0, %1\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
        } else if constexpr (T == 18) {  // 16 x (MFMA + 4 exp)
            asm volatile(REP16("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                         : "+v"(c0), "+v"(c1), "+v"(ka), "+v"(kb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        } else if constexpr (T == 19) {  // 64 MFMA 16x16x32 bf16 on 4 chains: the flops of 32 MFMA 32x32x16
            asm volatile(REP16("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n"
                               "v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n")
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(ka), "v"(kb));
        } else if constexpr (T == 20) {  // 16 x (2 MFMA 16x16x32 + 8 fma): the flops and vector work of T == 11
            asm volatile(REP16("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                               "v_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n v_fma_f32 %8, %8, %8, %8\n v_fma_f32 %9, %9, %9, %9\n v_fma_f32 %10, %10, %10, %10\n v_fma_f32 %11, %11, %11, %11\n")
                         : "+v"(q0), "+v"(q1), "+v"(ka), "+v"(kb), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0[0] + c1[1] + c2[2] + c3[3] + ka[0] + kb[1] + r0[0] + r1[1] + p0[0] + p1[1] + p2[0] + p3[1] + d0[0] + d1[1] + q0[0] + q1[1] + q2[2] + q3[3];
    if (sink == 12345.678f) lds[0] = sink;
    if ((threadIdx.x & 63) == 0) {
        out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = {t1 - t0, w1 - w0};
    }
    if (lds[1] == -1.f) out[0].cyc = 0;
}

template <int T>
void run(const char* name, int n_instr, int blocks) {
    Out* d;
    hipMalloc(&d, sizeof(Out) * 16 * 1024);
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256 * waves_per_simd;
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, ITER);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, d, ITER);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        std::vector<Out> h(blocks * threads / 64);
        hipMemcpy(h.data(), d, sizeof(Out) * h.size(), hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (auto& o : h) { cyc += o.cyc; wall += o.wall; }
        cyc /= h.size(); wall /= h.size();
        // per SIMD: waves_per_simd waves each issued ITER * n_instr instructions in `cyc` cycles
        printf("%-34s blocks=%3d waves/SIMD=%d: %7.2f cycles per instr and wave, %6.2f per instr and SIMD, clock %.0f MHz\n", name, blocks,
               waves_per_simd, cyc / (ITER * (double)n_instr), cyc / (ITER * (double)n_instr * waves_per_simd), cyc / wall * 100.0);
        printf("%-34s   wall ns per %d-instr iteration and wave: %.1f\n", "", n_instr, wall * 10.0 / ITER);
    }
    hipFree(d);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;
    if (argc > 2) {  // MFMA shapes only: equal flops per line pair, compare cycles x clock
        run<7>("mfma 32x32x16 bf16, 4 chains", 32, blocks);
        run<19>("mfma 16x16x32 bf16 x2, 4 chains", 32, blocks);
        run<11>("mfma 32x32x16 + 8 fma", 16, blocks);
        run<20>("2 mfma 16x16x32 + 8 fma", 16, blocks);
        return 0;
    }
    run<0>("v_fma_f32", 32, blocks);
    run<17>("v_pk_fma_f32", 32, blocks);
    run<5>("v_pk_mul_f32", 32, blocks);
    run<1>("v_exp_f32", 32, blocks);
    run<2>("v_cndmask_b32 (sgpr mask)", 32, blocks);
    run<3>("v_cvt_pkrtz_f16_f32", 32, blocks);
    run<4>("v_max3_f32", 32, blocks);
    run<16>("v_dot2c_f32_f16", 32, blocks);
    run<6>("v_mov_b32", 32, blocks);
    run<7>("mfma 32x32x16 bf16, 4 chains", 32, blocks);
    run<8>("mfma 32x32x16 bf16, 1 chain", 32, blocks);
    run<9>("mfma 32x32x16 bf16, 2 chains", 32, blocks);
    run<10>("mfma + 4 fma (per mfma)", 16, blocks);
    run<11>("mfma + 8 fma (per mfma)", 16, blocks);
    run<12>("mfma + 16 fma (per mfma)", 16, blocks);
    run<18>("mfma + 4 exp (per mfma)", 16, blocks);
    run<13>("ds_read_b128", 32, blocks);
    run<14>("ds_read_b64_tr_b16", 32, blocks);
    run<15>("mfma + 2 ds_read_tr + 4 fma", 16, blocks);
    return 0;
}
