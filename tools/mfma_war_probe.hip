// Hardware probe: how many wait states does gfx950 need between v_mfma_i32_16x16x64_i8 and a VALU write of one of its
// SrcC registers (SrcC != vDst)?  The instruction pair sits in inline assembly, where the compiler's hazard recogniser
// inserts nothing, so the distance is exactly what is written: K wait states (K = 0: back to back; else s_nop K-1),
// then v_mov_b32 of garbage into register R of SrcC. A wave's result is compared with the same MFMA left alone.
// Every SIMD is kept busy with MFMAs of other waves (the kernels' situation), 4 waves per SIMD.
//   for n in 0 1 2 3; do hipcc --offload-arch=gfx950 -O2 -DPRE_N=$n tools/mfma_war_probe.hip -o tools/bin/mfma_war_probe_p$n; done
// Output: one line per (R, K) with the number of wrong accumulator values over all lanes, waves and repetitions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

// PRE_N (compile-time, 0..3): independent MFMAs issued back to back in FRONT of the probed one, so that the probed
// instruction queues behind them in the matrix pipe -- the situation in the kernels' block loop
#ifndef PRE_N
#define PRE_N 0
#endif
#define PRE_A "v_mfma_i32_16x16x64_i8 v[36:39], v[20:23], v[24:27], v[36:39]\n\t"
#define PRE_B "v_mfma_i32_16x16x64_i8 v[40:43], v[24:27], v[20:23], v[40:43]\n\t"
#if PRE_N == 0
#define PRE_STR ""
#elif PRE_N == 1
#define PRE_STR PRE_A
#elif PRE_N == 2
#define PRE_STR PRE_A PRE_B
#else
#define PRE_STR PRE_A PRE_B PRE_A
#endif

// The accumulator sits in v[28:31] by explicit register numbers (an operand constraint cannot name one register of a tuple).
template <int R, int K>
__global__ void __launch_bounds__(256) probe_kernel(const int* __restrict__ in, int* __restrict__ out, int reps) {
    const int lane = threadIdx.x & 63;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    int wrong = 0;
    for (int it = 0; it < reps; ++it) {
        // operands: small signed bytes so nothing saturates; different per lane and iteration
        int a0 = in[(gid * 4 + 0 + it) & 4095], a1 = in[(gid * 4 + 1 + it) & 4095], a2 = in[(gid * 4 + 2 + it) & 4095],
            a3 = in[(gid * 4 + 3 + it) & 4095];
        int b0 = in[(gid * 4 + 7 + it) & 4095], b1 = in[(gid * 4 + 11 + it) & 4095], b2 = in[(gid * 4 + 13 + it) & 4095],
            b3 = in[(gid * 4 + 17 + it) & 4095];
        int c0 = lane * 3 + it, c1 = lane * 5 - it, c2 = -lane * 7 + 11 * it, c3 = lane * 11 + 1;
        int r0, r1, r2, r3, d0, d1, d2, d3;
        // reference: the MFMA alone, well separated
        asm volatile(
            "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v22, %6\n\tv_mov_b32 v23, %7\n\t"
            "v_mov_b32 v24, %8\n\tv_mov_b32 v25, %9\n\tv_mov_b32 v26, %10\n\tv_mov_b32 v27, %11\n\t"
            "v_mov_b32 v28, %12\n\tv_mov_b32 v29, %13\n\tv_mov_b32 v30, %14\n\tv_mov_b32 v31, %15\n\t"
            "s_nop 7\n\t"
            "v_mfma_i32_16x16x64_i8 v[32:35], v[20:23], v[24:27], v[28:31]\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v34\n\tv_mov_b32 %3, v35\n\t"
            : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(c0), "v"(c1), "v"(c2), "v"(c3)
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35");
        // probe: the same MFMA, SrcC register R overwritten K wait states later
#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n\t"
#define NOPS_2 "s_nop 1\n\t"
#define NOPS_3 "s_nop 2\n\t"
#define NOPS_4 "s_nop 3\n\t"
#define NOPS_5 "s_nop 4\n\t"
#define NOPS_6 "s_nop 5\n\t"
#define NOPS_7 "s_nop 6\n\t"
#define NOPS_8 "s_nop 7\n\t"
#define PROBE(NOPS, REG)     PROBE_P(PRE_STR, NOPS, REG)
#define PROBE_P(PRE, NOPS, REG)                                                                                             \
        asm volatile(                                                                                                 \
            "v_mov_b32 v20, %4\n\tv_mov_b32 v21, %5\n\tv_mov_b32 v22, %6\n\tv_mov_b32 v23, %7\n\t"                    \
            "v_mov_b32 v24, %8\n\tv_mov_b32 v25, %9\n\tv_mov_b32 v26, %10\n\tv_mov_b32 v27, %11\n\t"                  \
            "v_mov_b32 v28, %12\n\tv_mov_b32 v29, %13\n\tv_mov_b32 v30, %14\n\tv_mov_b32 v31, %15\n\t"                \
            "v_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t"                        \
            "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"                        \
            "s_nop 7\n\t" PRE                                                                                         \
            "v_mfma_i32_16x16x64_i8 v[32:35], v[20:23], v[24:27], v[28:31]\n\t" NOPS                                  \
            "v_mov_b32 " REG ", 0x12345678\n\t"                                                                       \
            "s_nop 15\n\ts_nop 15\n\t"                                                                                \
            "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v33\n\tv_mov_b32 %2, v34\n\tv_mov_b32 %3, v35\n\t"                    \
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)                                                                  \
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(c0), "v"(c1), "v"(c2),      \
              "v"(c3)                                                                                                 \
            : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", \
              "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43")
        if constexpr (K == 0) { if constexpr (R == 0) PROBE(NOPS_0, "v28"); else if constexpr (R == 1) PROBE(NOPS_0, "v29"); else if constexpr (R == 2) PROBE(NOPS_0, "v30"); else PROBE(NOPS_0, "v31"); }
        if constexpr (K == 1) { if constexpr (R == 0) PROBE(NOPS_1, "v28"); else if constexpr (R == 1) PROBE(NOPS_1, "v29"); else if constexpr (R == 2) PROBE(NOPS_1, "v30"); else PROBE(NOPS_1, "v31"); }
        if constexpr (K == 2) { if constexpr (R == 0) PROBE(NOPS_2, "v28"); else if constexpr (R == 1) PROBE(NOPS_2, "v29"); else if constexpr (R == 2) PROBE(NOPS_2, "v30"); else PROBE(NOPS_2, "v31"); }
        if constexpr (K == 3) { if constexpr (R == 0) PROBE(NOPS_3, "v28"); else if constexpr (R == 1) PROBE(NOPS_3, "v29"); else if constexpr (R == 2) PROBE(NOPS_3, "v30"); else PROBE(NOPS_3, "v31"); }
        if constexpr (K == 4) { if constexpr (R == 0) PROBE(NOPS_4, "v28"); else if constexpr (R == 1) PROBE(NOPS_4, "v29"); else if constexpr (R == 2) PROBE(NOPS_4, "v30"); else PROBE(NOPS_4, "v31"); }
        if constexpr (K == 5) { if constexpr (R == 0) PROBE(NOPS_5, "v28"); else if constexpr (R == 1) PROBE(NOPS_5, "v29"); else if constexpr (R == 2) PROBE(NOPS_5, "v30"); else PROBE(NOPS_5, "v31"); }
        if constexpr (K == 6) { if constexpr (R == 0) PROBE(NOPS_6, "v28"); else if constexpr (R == 1) PROBE(NOPS_6, "v29"); else if constexpr (R == 2) PROBE(NOPS_6, "v30"); else PROBE(NOPS_6, "v31"); }
        if constexpr (K == 7) { if constexpr (R == 0) PROBE(NOPS_7, "v28"); else if constexpr (R == 1) PROBE(NOPS_7, "v29"); else if constexpr (R == 2) PROBE(NOPS_7, "v30"); else PROBE(NOPS_7, "v31"); }
        if constexpr (K == 8) { if constexpr (R == 0) PROBE(NOPS_8, "v28"); else if constexpr (R == 1) PROBE(NOPS_8, "v29"); else if constexpr (R == 2) PROBE(NOPS_8, "v30"); else PROBE(NOPS_8, "v31"); }
        wrong += (d0 != r0) + (d1 != r1) + (d2 != r2) + (d3 != r3);
    }
    if (wrong) atomicAdd(out, wrong);
}

template <int R, int K>
static void run(const int* d_in, int* d_out, int blocks, int reps) {
    (void)hipMemset(d_out, 0, 4);
    hipLaunchKernelGGL((probe_kernel<R, K>), dim3(blocks), dim3(256), 0, 0, d_in, d_out, reps);
    int wrong = -1;
    (void)hipMemcpy(&wrong, d_out, 4, hipMemcpyDeviceToHost);
    printf("%d MFMAs in front, SrcC register %d overwritten after %d wait states: %d wrong values of %lld\n", PRE_N, R, K, wrong,
           (long long)blocks * 256 * reps * 4);
}

template <int R>
static void run_all_k(const int* d_in, int* d_out, int blocks, int reps) {
    run<R, 0>(d_in, d_out, blocks, reps); run<R, 1>(d_in, d_out, blocks, reps); run<R, 2>(d_in, d_out, blocks, reps);
    run<R, 3>(d_in, d_out, blocks, reps); run<R, 4>(d_in, d_out, blocks, reps); run<R, 5>(d_in, d_out, blocks, reps);
    run<R, 6>(d_in, d_out, blocks, reps); run<R, 7>(d_in, d_out, blocks, reps); run<R, 8>(d_in, d_out, blocks, reps);
}

int main() {
    std::vector<int> h(4096);
    srand(5);
    for (auto& v : h) {   // four signed bytes in -8..7 per word
        unsigned w = 0;
        for (int b = 0; b < 4; ++b) w |= (unsigned)((rand() % 16 - 8) & 0xff) << (8 * b);
        v = (int)w;
    }
    int *d_in, *d_out;
    if (hipMalloc(&d_in, 4096 * 4) != hipSuccess || hipMalloc(&d_out, 4) != hipSuccess) { printf("no device\n"); return 1; }
    (void)hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int blocks = 1024, reps = 64;   // 4 waves per SIMD on every CU
    run_all_k<0>(d_in, d_out, blocks, reps);
    run_all_k<1>(d_in, d_out, blocks, reps);
    run_all_k<2>(d_in, d_out, blocks, reps);
    run_all_k<3>(d_in, d_out, blocks, reps);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
