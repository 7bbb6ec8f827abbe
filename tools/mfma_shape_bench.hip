// Which int8 MFMA shape suits the kNN kernel's block loop? (tools only: nothing here ships)
//
// The kNN kernel's inner loop is: A fragments (train rows) from LDS, B fragments (queries) resident in registers, a chain
// of MFMAs per (tile, query block) and an epilogue that takes the maximum over the block's results, compares it with the
// query's bound and branches (rarely) into the insertion path. With v_mfma_i32_16x16x64_i8 a 128-d block is 2 MFMAs of 16
// cycles and 4 results per lane; with v_mfma_i32_32x32x32_i8 it is 4 MFMAs of 32 cycles and 16 results per lane. Per
// MI355X_MICROARCH.md an MFMA holds the SIMD's vector issue for 8 cycles whatever its length, so the short shape leaves 2
// vector-instruction slots per MFMA and the long one 6 -- but the guide also measures the 16x16 shape at a higher clock.
// This bench runs both loops (operands random, every CU busy, the kNN kernel's 40 KB of LDS per workgroup) and prints
// op/s, so the trade can be read off instead of argued.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_bench.hip -o tools/bin/mfma_shape_bench && tools/bin/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define TN 128                       // train rows per stage
#define STAGE_BYTES (TN * 128 + TN * 4)

// ---- 16x16x64: the kernel's pair loop (two 16-row tiles per step, 8 results per lane and query block) ------------------
template <int QB, int EPI, int WGS>
__global__ void __launch_bounds__(256, WGS)
loop16(const uint4* __restrict__ Q, const uint4* __restrict__ T, const int* __restrict__ Gq, int iters, int* __restrict__ sink) {
    __shared__ uint4 lds[40960 / 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < STAGE_BYTES / 16; i += 256) lds[i] = T[(blockIdx.x % 61) * (STAGE_BYTES / 16) + i];
    i32x4 qf[QB][2];
    int G[QB];
    const size_t qrow0 = ((size_t)blockIdx.x * 4 + wave) * (16 * QB);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int t = 0; t < 2; ++t) qf[qb][t] = __builtin_bit_cast(i32x4, Q[((qrow0 + qb * 16 + r) * 8 + 4 * t + q) & 0xfffff]);
        G[qb] = Gq[(qrow0 + qb * 16 + r) & 0xffff];
    }
    __syncthreads();
    const unsigned frag_off = (lane & 15) * 128u + ((((lane >> 4) ^ lane) & 7) << 4);
    int hits = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int tp = 0; tp < TN / 32; ++tp) {
            i32x4 afA[2], afB[2];
            const unsigned a0 = (unsigned)tp * 4096u + frag_off;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                afA[t] = __builtin_bit_cast(i32x4, *(const uint4*)((const char*)lds + (a0 ^ (unsigned)(t << 6))));
                afB[t] = __builtin_bit_cast(i32x4, *(const uint4*)((const char*)lds + ((a0 + 2048u) ^ (unsigned)(t << 6))));
            }
            const i32x4 tinA = __builtin_bit_cast(i32x4, lds[TN * 8 + tp * 8 + q]) >> 1;
            const i32x4 tinB = __builtin_bit_cast(i32x4, lds[TN * 8 + tp * 8 + 4 + q]) >> 1;
            i32x4 P[2][2];
            auto chain2 = [&](int qb, i32x4 (&x)[2]) {
                x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afA[0], qf[qb][0], tinA, 0, 0, 0);
                x[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afB[0], qf[qb][0], tinB, 0, 0, 0);
                x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afA[1], qf[qb][1], x[0], 0, 0, 0);
                x[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afB[1], qf[qb][1], x[1], 0, 0, 0);
            };
            chain2(0, P[0]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const int cs = qb & 1, ns = cs ^ 1;
                if (qb + 1 < QB) chain2(qb + 1, P[ns]);
                if (EPI) {
                    int m = max(max(P[cs][0][0], P[cs][0][1]), max(P[cs][0][2], P[cs][0][3]));
                    m = max(max(m, P[cs][1][0]), max(P[cs][1][1], max(P[cs][1][2], P[cs][1][3])));
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= G[qb]) != 0, 0)) {
                        hits += P[cs][0][0] + P[cs][1][3];
                        G[qb] += 1;
                    }
                } else {
                    asm volatile("" :: "v"(P[cs][0]), "v"(P[cs][1]));
                }
            }
        }
    }
    if (hits == 0x12345678) sink[threadIdx.x] = hits;
}

// ---- 32x32x32: one 32-row tile per step, 16 results per lane and query block of 32 --------------------------------------
template <int QB, int EPI, int WGS>
__global__ void __launch_bounds__(256, WGS)
loop32(const uint4* __restrict__ Q, const uint4* __restrict__ T, const int* __restrict__ Gq, int iters, int* __restrict__ sink) {
    __shared__ uint4 lds[40960 / 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < STAGE_BYTES / 16; i += 256) lds[i] = T[(blockIdx.x % 61) * (STAGE_BYTES / 16) + i];
    i32x4 qf[QB][4];
    int G[QB];
    const size_t qrow0 = ((size_t)blockIdx.x * 4 + wave) * (32 * QB);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int t = 0; t < 4; ++t) qf[qb][t] = __builtin_bit_cast(i32x4, Q[((qrow0 + qb * 32 + r) * 8 + 2 * t + h) & 0xfffff]);
        G[qb] = Gq[(qrow0 + qb * 32 + r) & 0xffff];
    }
    __syncthreads();
    // row r of the tile, 16-byte chunk 2t + h of its 8, swizzled by the row
    const unsigned frag_off = (unsigned)r * 128u;
    int hits = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int tile = 0; tile < TN / 32; ++tile) {
            i32x4 af[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                af[t] = __builtin_bit_cast(i32x4, *(const uint4*)((const char*)lds + tile * 4096u + frag_off + ((((unsigned)(2 * t + h)) ^ (r & 7)) << 4)));
            // C starts at T' >> 1 of the result row: acc register g holds row (g & 3) + 8 (g >> 2) + 4 h
            i32x16 tin;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const i32x4 w = __builtin_bit_cast(i32x4, lds[TN * 8 + tile * 8 + 2 * g4 + h]) >> 1;
                tin[4 * g4] = w[0]; tin[4 * g4 + 1] = w[1]; tin[4 * g4 + 2] = w[2]; tin[4 * g4 + 3] = w[3];
            }
            i32x16 P[2];
            auto chain = [&](int qb, i32x16& x) {
                x = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], qf[qb][0], tin, 0, 0, 0);
#pragma unroll
                for (int t = 1; t < 4; ++t) x = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[t], qf[qb][t], x, 0, 0, 0);
            };
            chain(0, P[0]);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const int cs = qb & 1, ns = cs ^ 1;
                if (qb + 1 < QB) chain(qb + 1, P[ns]);
                if (EPI) {
                    int m = P[cs][0];
#pragma unroll
                    for (int g = 1; g < 16; ++g) m = max(m, P[cs][g]);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= G[qb]) != 0, 0)) {
                        hits += P[cs][0] + P[cs][15];
                        G[qb] += 1;
                    }
                } else {
                    asm volatile("" :: "v"(P[cs]));
                }
            }
        }
    }
    if (hits == 0x12345678) sink[threadIdx.x] = hits;
}


// ---- 16x16x64 on QUADS of tiles: four 16-row tiles per step, 16 results per lane and query block, one max tree, one
// compare, one branch per eight MFMAs; DB = 1 keeps two accumulator sets (the next block's chains are out before this
// block's epilogue), DB = 0 one set (fewer registers: a fourth workgroup per CU with QB = 6) -------------------------------
template <int QB, int DB, int WGS>
__global__ void __launch_bounds__(256, WGS)
loop16q(const uint4* __restrict__ Q, const uint4* __restrict__ T, const int* __restrict__ Gq, int iters, int* __restrict__ sink) {
    __shared__ uint4 lds[39936 / 16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < STAGE_BYTES / 16; i += 256) lds[i] = T[(blockIdx.x % 61) * (STAGE_BYTES / 16) + i];
    i32x4 qf[QB][2];
    int G[QB];
    const size_t qrow0 = ((size_t)blockIdx.x * 4 + wave) * (16 * QB);
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int t = 0; t < 2; ++t) qf[qb][t] = __builtin_bit_cast(i32x4, Q[((qrow0 + qb * 16 + r) * 8 + 4 * t + q) & 0xfffff]);
        G[qb] = Gq[(qrow0 + qb * 16 + r) & 0xffff];
    }
    __syncthreads();
    const unsigned frag_off = (lane & 15) * 128u + ((((lane >> 4) ^ lane) & 7) << 4);
    int hits = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 1
        for (int tq = 0; tq < TN / 64; ++tq) {
            i32x4 af[4][2], tin[4];
            const unsigned a0 = (unsigned)tq * 8192u + frag_off;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    af[u][t] = __builtin_bit_cast(i32x4, *(const uint4*)((const char*)lds + ((a0 + 2048u * u) ^ (unsigned)(t << 6))));
                tin[u] = __builtin_bit_cast(i32x4, lds[TN * 8 + tq * 16 + 4 * u + q]) >> 1;
            }
            i32x4 P[DB ? 2 : 1][4];
            auto chain4 = [&](int qb, i32x4 (&x)[4]) {
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u][0], qf[qb][0], tin[u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[u][1], qf[qb][1], x[u], 0, 0, 0);
            };
            auto epi = [&](const i32x4 (&x)[4], int qb) {
                int m = x[0][0];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) m = max(m, x[u][g]);
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= G[qb]) != 0, 0)) {
                    hits += x[0][0] + x[3][3];
                    G[qb] += 1;
                }
            };
            if (DB) {
                chain4(0, P[0]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const int cs = qb & 1, ns = cs ^ 1;
                    if (qb + 1 < QB) chain4(qb + 1, P[ns]);
                    epi(P[cs], qb);
                }
            } else {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    chain4(qb, P[0]);
                    epi(P[0], qb);
                }
            }
        }
    }
    if (hits == 0x12345678) sink[threadIdx.x] = hits;
}

static double run(const char* name, void (*launch)(int, int, hipStream_t), double ops_per_wg_iter, int grid, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) launch(grid, iters, 0);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0, 0));
    for (int w = 0; w < reps; ++w) launch(grid, iters, 0);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double tops = ops_per_wg_iter * grid * iters / (ms * 1e-3) / 1e12;
    printf("%-34s %8.3f ms  %8.1f Top/s  %5.1f %% of 5 P\n", name, ms, tops, tops / 50.0);
    fflush(stdout);
    return tops;
}

static uint4 *dQ, *dT;
static int *dG, *dS;

#define LAUNCHER(fn, K) static void fn(int grid, int iters, hipStream_t st) { hipLaunchKernelGGL((K), dim3(grid), dim3(256), 0, st, dQ, dT, dG, iters, dS); }
LAUNCHER(l16_e1_3, (loop16<8, 1, 3>))
LAUNCHER(l16_e0_3, (loop16<8, 0, 3>))
LAUNCHER(l16_e1_2, (loop16<8, 1, 2>))
LAUNCHER(l16q8_db_2, (loop16q<8, 1, 2>))
LAUNCHER(l16q8_sb_3, (loop16q<8, 0, 3>))
LAUNCHER(l16q6_db_3, (loop16q<6, 1, 3>))
LAUNCHER(l16q6_sb_4, (loop16q<6, 0, 4>))
LAUNCHER(l16q6_sb_3, (loop16q<6, 0, 3>))
LAUNCHER(l32_e1_2, (loop32<4, 1, 2>))
LAUNCHER(l32_e0_2, (loop32<4, 0, 2>))
LAUNCHER(l32_e1_3, (loop32<4, 1, 3>))
LAUNCHER(l32_e0_3, (loop32<4, 0, 3>))
LAUNCHER(l32q3_e1_3, (loop32<3, 1, 3>))
LAUNCHER(l32q2_e1_3, (loop32<2, 1, 3>))
LAUNCHER(l32q3_e0_3, (loop32<3, 0, 3>))

int main(int argc, char** argv) {
    const int sift_like = argc > 1 ? atoi(argv[1]) : 0;   // 1: bytes x - 128 with small x, as SIFT rows are stored
    const size_t nq16 = (size_t)1 << 20, nt16 = (size_t)61 * STAGE_BYTES / 16;
    std::vector<unsigned char> hq(nq16 * 16), ht(nt16 * 16);
    srand(7);
    auto fill = [&](std::vector<unsigned char>& v) {
        for (auto& b : v) {
            if (sift_like) { int x = rand() % 64; x = (rand() & 3) ? x / 2 : x * 3; b = (unsigned char)((x > 255 ? 255 : x) - 128); }
            else b = (unsigned char)rand();
        }
    };
    fill(hq);
    fill(ht);
    // the T' words of every stage: plausible negative norms
    for (int s = 0; s < 61; ++s)
        for (int i = 0; i < TN; ++i) ((int*)(ht.data() + (size_t)s * STAGE_BYTES + TN * 128))[i] = -(1000000 + rand() % 500000);
    std::vector<int> hg(65536, 0x7fffffff);
    CHECK(hipMalloc((void**)&dQ, hq.size()));
    CHECK(hipMalloc((void**)&dT, ht.size()));
    CHECK(hipMalloc((void**)&dG, hg.size() * 4));
    CHECK(hipMalloc((void**)&dS, 4096));
    CHECK(hipMemcpy(dQ, hq.data(), hq.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dT, ht.data(), ht.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dG, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    // operations per workgroup and iteration: 4 waves x (queries per wave) x TN rows x 256
    printf("operands: %s, %d iterations of a %d-row stage per workgroup\n", sift_like ? "SIFT-like bytes" : "uniform random bytes", iters, TN);
    run("16x16x64 pairs, epilogue, 3 wg/CU", l16_e1_3, 4.0 * 128 * TN * 256, 768 * 2, iters);
    run("16x16x64 pairs, no epilogue, 3 wg/CU", l16_e0_3, 4.0 * 128 * TN * 256, 768 * 2, iters);
    run("16x16x64 pairs, epilogue, 2 wg/CU", l16_e1_2, 4.0 * 128 * TN * 256, 512 * 2, iters);
    run("16x16x64 quads QB8 2 sets, 2 wg/CU", l16q8_db_2, 4.0 * 128 * TN * 256, 512 * 2, iters);
    run("16x16x64 quads QB8 1 set, 3 wg/CU", l16q8_sb_3, 4.0 * 128 * TN * 256, 768 * 2, iters);
    run("16x16x64 quads QB6 2 sets, 3 wg/CU", l16q6_db_3, 4.0 * 96 * TN * 256, 768 * 2, iters);
    run("16x16x64 quads QB6 1 set, 3 wg/CU", l16q6_sb_3, 4.0 * 96 * TN * 256, 768 * 2, iters);
    run("16x16x64 quads QB6 1 set, 4 wg/CU", l16q6_sb_4, 4.0 * 96 * TN * 256, 1024 * 2, iters);
    run("32x32x32 QB4, epilogue, 2 wg/CU", l32_e1_2, 4.0 * 128 * TN * 256, 512 * 2, iters);
    run("32x32x32 QB4, no epilogue, 2 wg/CU", l32_e0_2, 4.0 * 128 * TN * 256, 512 * 2, iters);
    run("32x32x32 QB4, epilogue, 3 wg/CU", l32_e1_3, 4.0 * 128 * TN * 256, 768 * 2, iters);
    run("32x32x32 QB4, no epilogue, 3 wg/CU", l32_e0_3, 4.0 * 128 * TN * 256, 768 * 2, iters);
    run("32x32x32 QB3, epilogue, 3 wg/CU", l32q3_e1_3, 4.0 * 96 * TN * 256, 768 * 2, iters);
    run("32x32x32 QB3, no epilogue, 3 wg/CU", l32q3_e0_3, 4.0 * 96 * TN * 256, 768 * 2, iters);
    run("32x32x32 QB2, epilogue, 3 wg/CU", l32q2_e1_3, 4.0 * 64 * TN * 256, 768 * 2, iters);
    return 0;
}
