// Ablation harness for the blur kernel (tool, not product): times variants on a 16384x16384 f32 layer.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
template <int R> struct GK { float k[2 * R + 1]; };
__device__ __forceinline__ int reflect101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
// MODE 0: full; 1: fill only; 2: fill + row pass; 3: no fill
template <int R, int MODE>
__global__ void __launch_bounds__(256, 4)
blur_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog, int rows, int cols, GK<R> gk) {
    constexpr int N = 2 * R + 1, TH = 32, H2 = TH + 2 * R;      // H2 <= 58: one lane per tile row
    constexpr int XO = (R + 3) & ~3, P = 64 + 2 * XO;            // x origin of the tile = x0 - XO
    constexpr int PP = P + 4, TP = 68;                           // padded pitches: (PP/4) and (TP/4) are odd
    constexpr int NX4 = P / 4, NITEM = H2 * NX4;
    static_assert(H2 <= 64, "row pass maps one lane to one tile row");
    __shared__ float lds[H2 * PP + H2 * TP];
    float* in = lds;
    float* tmp = lds + H2 * PP;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * TH;
    // Tiles that touch the image border, or images whose rows are not 16-byte multiples, are filled
    // element-wise with the reflect-101 mapping.
    const bool interior = (cols & 3) == 0 && x0 - XO >= 0 && x0 - XO + P <= cols && y0 - R >= 0 && y0 - R + H2 <= rows;
    if (MODE == 3) {} else if (interior) {
        constexpr int NIT = (NITEM + 255) / 256;
        float4 a[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = min(tid + it * 256, NITEM - 1);
            const int row = item / NX4, x4 = item - row * NX4;
            a[it] = *(const float4*)(src + (size_t)(y0 - R + row) * cols + (x0 - XO + 4 * x4));
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * 256;
            if (item < NITEM) {
                const int row = item / NX4, x4 = item - row * NX4;
                *(float4*)&in[row * PP + 4 * x4] = a[it];
            }
        }
    } else {
        for (int f = tid; f < H2 * P; f += 256) {
            const int row = f / P, col = f - row * P;
            in[row * PP + col] = src[(size_t)reflect101(y0 - R + row, rows) * cols + reflect101(x0 - XO + col, cols)];
        }
    }
    __syncthreads();
    if (MODE == 1) { for (int j = 0; j < 8; ++j) { size_t o = (size_t)(y0 + wv * 8 + j) * cols + x0 + lane; dst[o] = in[(wv * 8 + j) * PP + lane]; dog[o] = in[(wv * 8 + j + 1) * PP + lane]; } return; }
    // row pass: lane = tile row, wave = group of 8 outputs (two rounds cover the 64 columns)
    if (lane < H2) {
        constexpr int A0 = (XO - R) & ~3, A1 = (XO + R + 8 + 3) & ~3, OFF = XO - R - A0;
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int xq = (wv + 4 * rnd) * 8;
            float v[A1 - A0];
#pragma unroll
            for (int i = 0; i < (A1 - A0) / 4; ++i) {
                const float4 q = *(const float4*)&in[lane * PP + xq + A0 + 4 * i];
                v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
            }
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int t = 0; t < N; ++t) {
                const float k = gk.k[t];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(k, v[OFF + t + j], acc[j]);
            }
            *(float4*)&tmp[lane * TP + xq] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *(float4*)&tmp[lane * TP + xq + 4] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    }
    __syncthreads();
    if (MODE == 2) { for (int j = 0; j < 8; ++j) { size_t o = (size_t)(y0 + wv * 8 + j) * cols + x0 + lane; dst[o] = tmp[(wv * 8 + j) * TP + lane]; dog[o] = tmp[(wv * 8 + j + 1) * TP + lane]; } return; }
    // column pass: lane = column, wave = 8 output rows
    {
        const int r8 = wv * 8;
        float v[8 + 2 * R];
#pragma unroll
        for (int i = 0; i < 8 + 2 * R; ++i) v[i] = tmp[(r8 + i) * TP + lane];
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float k = gk.k[t];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(k, v[t + j], acc[j]);
        }
        const int gx = x0 + lane;
        if (gx < cols) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gy = y0 + r8 + j;
                if (gy < rows) {
                    dst[(size_t)gy * cols + gx] = acc[j];
                    if (dog) dog[(size_t)gy * cols + gx] = acc[j] - in[(r8 + j + R) * PP + lane + XO];
                }
            }
        }
    }
}

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, float4* __restrict__ c, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 v = a[i]; b[i] = v; c[i] = v; }
}
template <int R, int MODE> float run(const float* s, float* d, float* g, int n) {
    GK<R> gk; for (int i = 0; i < 2 * R + 1; ++i) gk.k[i] = 1.f / (2 * R + 1);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid(n / 64, n / 32);
    hipLaunchKernelGGL((blur_kernel<R, MODE>), grid, dim3(256), 0, 0, s, d, g, n, n, gk);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((blur_kernel<R, MODE>), grid, dim3(256), 0, 0, s, d, g, n, n, gk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
    const int n = 16384; const size_t px = (size_t)n * n;
    float *s, *d, *g; hipMalloc(&s, px * 4); hipMalloc(&d, px * 4); hipMalloc(&g, px * 4);
    hipMemset(s, 0x3c, px * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(copy_kernel, dim3(px / 4 / 256), dim3(256), 0, 0, (const float4*)s, (float4*)d, (float4*)g, px / 4);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel, dim3(px / 4 / 256), dim3(256), 0, 0, (const float4*)s, (float4*)d, (float4*)g, px / 4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("copy 1r+2w: %.3f ms  %.2f TB/s\n", ms / 3, px * 12.0 / (ms / 3 * 1e-3) / 1e12);
    printf("R=13 full %.3f | load-only %.3f | load+row %.3f | compute-only %.3f ms\n", run<13, 0>(s, d, g, n), run<13, 1>(s, d, g, n), run<13, 2>(s, d, g, n), run<13, 3>(s, d, g, n));
    printf("R=5  full %.3f | load-only %.3f | load+row %.3f | compute-only %.3f ms\n", run<5, 0>(s, d, g, n), run<5, 1>(s, d, g, n), run<5, 2>(s, d, g, n), run<5, 3>(s, d, g, n));
    printf("R=8  full %.3f | load-only %.3f | load+row %.3f | compute-only %.3f ms\n", run<8, 0>(s, d, g, n), run<8, 1>(s, d, g, n), run<8, 2>(s, d, g, n), run<8, 3>(s, d, g, n));
    return 0;
}
