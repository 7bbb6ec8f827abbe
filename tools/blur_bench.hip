// Ablation / exactness harness for the marching blur kernel (tool, not product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/blur_bench tools/blur_bench.hip
// Checks blur_march_kernel bit-for-bit against a naive two-pass kernel (same ascending-tap fma chains) on
// odd-sized and large images, f32 and u8-upsampled sources, with and without the decimated output, then
// times it on a 16384x16384 layer and on a batch-like 1280x960 x 32 layer for each radius / strip / march.
#include "../image_matching_amd/csrc/blur.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct GKrt { int r; float k[41]; };
__global__ void ref_rows(const float* src, float* tmp, int rows, int cols, GKrt gk) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    float acc = 0.f;
    for (int t = 0; t < 2 * gk.r + 1; ++t) acc = fmaf(gk.k[t], src[(size_t)y * cols + reflect101(x + t - gk.r, cols)], acc);
    tmp[(size_t)y * cols + x] = acc;
}
__global__ void ref_cols(const float* tmp, float* dst, int rows, int cols, GKrt gk) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    float acc = 0.f;
    for (int t = 0; t < 2 * gk.r + 1; ++t) acc = fmaf(gk.k[t], tmp[(size_t)reflect101(y + t - gk.r, rows) * cols + x], acc);
    dst[(size_t)y * cols + x] = acc;
}
__global__ void ref_up(const uint8_t* src, int rows, int cols, float* dst) {   // the formula of createInitialImage
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= 2 * cols || dy >= 2 * rows) return;
    float fx = (float)((dx + 0.5) * 0.5 - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
    float fy = (float)((dy + 0.5) * 0.5 - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= rows - 1) { fy = 0; sy = rows - 1; }
    const int sx1 = min(sx + 1, cols - 1), sy1 = min(sy + 1, rows - 1);
    const uint8_t* s0 = src + (size_t)sy * cols;
    const uint8_t* s1 = src + (size_t)sy1 * cols;
    const float a1 = fx, a0 = 1.f - a1;
    const float h0 = s0[sx] * a0 + s0[sx1] * a1;
    const float h1 = s1[sx] * a0 + s1[sx1] * a1;
    dst[(size_t)dy * (2 * cols) + dx] = h0 * (1.f - fy) + h1 * fy;
}
__global__ void fill_f32(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) * (255.f / 65535.f);
}
__global__ void fill_image(float* p, int cols, size_t n, unsigned seed) {   // textured, image-like content
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % cols), y = (int)(i / cols);
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = 127.f + 60.f * __sinf(x * 0.013f) * __cosf(y * 0.017f) + 40.f * __sinf((x + 2 * y) * 0.11f) + (float)(h & 15);
    if (((x >> 7) + (y >> 6)) & 1) v *= 0.6f;
    p[i] = fminf(fmaxf(rintf(v), 0.f), 255.f);
}
__global__ void fill_u8(uint8_t* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (uint8_t)(h >> 7);
}
__global__ void cmp_kernel(const float* a, const float* b, size_t n, unsigned long long* bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (__float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(bad, 1ull);
}
__global__ void cmp_dec(const float* full, const float* dec, int rows, int cols, unsigned long long* bad) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols / 2 || y >= rows / 2) return;
    if (__float_as_uint(full[(size_t)(2 * y) * cols + 2 * x]) != __float_as_uint(dec[(size_t)y * (cols / 2) + x])) atomicAdd(bad, 1ull);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

static void gauss(int R, float* k) {   // any normalised, asymmetric-free kernel will do for the check
    double sigma = (2 * R + 1 - 1) / 8.0, sum = 0;
    std::vector<double> v(2 * R + 1);
    for (int i = 0; i <= 2 * R; ++i) { v[i] = exp(-(double)(i - R) * (i - R) / (2 * sigma * sigma)); sum += v[i]; }
    for (int i = 0; i <= 2 * R; ++i) k[i] = (float)(v[i] / sum);
}

static unsigned long long* d_bad;
static unsigned long long count_bad(const float* a, const float* b, size_t n) {
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(cmp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n, d_bad);
    unsigned long long h;
    CK(hipMemcpy(&h, d_bad, 8, hipMemcpyDeviceToHost));
    return h;
}

template <int R, int W>
static int check(int rows, int cols, int nt, float* src, float* tmp, float* ref, float* dst, float* dec, uint8_t* u8) {
    float k[2 * R + 1];
    gauss(R, k);
    GKrt g; g.r = R; memcpy(g.k, k, sizeof(k));
    const size_t n = (size_t)rows * cols;
    const dim3 grid((cols + 63) / 64, (rows + 3) / 4);
    int fails = 0;
    // f32 source (+ decimated output)
    hipLaunchKernelGGL(fill_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, src, n, 77u + R);
    hipLaunchKernelGGL(ref_rows, grid, dim3(256), 0, 0, (const float*)src, tmp, rows, cols, g);
    hipLaunchKernelGGL(ref_cols, grid, dim3(256), 0, 0, (const float*)tmp, ref, rows, cols, g);
    CK(hipMemset(dst, 0xff, n * 4));
    blur_march_launch<R, W, false>(0, src, dst, dec, rows, cols, 0, k, 1, 0, 0, nt);
    CK(hipDeviceSynchronize());
    unsigned long long bad = count_bad(ref, dst, n);
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(cmp_dec, grid, dim3(256), 0, 0, (const float*)ref, (const float*)dec, rows, cols, d_bad);
    unsigned long long bad_dec;
    CK(hipMemcpy(&bad_dec, d_bad, 8, hipMemcpyDeviceToHost));
    if (bad || bad_dec) { printf("  MISMATCH f32 R=%d W=%d %dx%d nt=%d: %llu px, dec %llu px\n", R, W, rows, cols, nt, bad, bad_dec); ++fails; }
    // u8 source upsampled on the fly (rows, cols must be even here)
    if (!(rows & 1) && !(cols & 1)) {
        const int ur = rows / 2, uc = cols / 2;
        hipLaunchKernelGGL(fill_u8, dim3((unsigned)(((size_t)ur * uc + 255) / 256)), dim3(256), 0, 0, u8, (size_t)ur * uc, 5u + R);
        hipLaunchKernelGGL(ref_up, grid, dim3(256), 0, 0, (const uint8_t*)u8, ur, uc, src);
        hipLaunchKernelGGL(ref_rows, grid, dim3(256), 0, 0, (const float*)src, tmp, rows, cols, g);
        hipLaunchKernelGGL(ref_cols, grid, dim3(256), 0, 0, (const float*)tmp, ref, rows, cols, g);
        CK(hipMemset(dst, 0xff, n * 4));
        blur_march_launch<R, W, true>(0, u8, dst, nullptr, rows, cols, uc, k, 1, 0, 0, nt);
        CK(hipDeviceSynchronize());
        bad = count_bad(ref, dst, n);
        if (bad) { printf("  MISMATCH u8-up R=%d W=%d %dx%d nt=%d: %llu px\n", R, W, rows, cols, nt, bad); ++fails; }
    }
    return fails;
}

template <int R, int W, bool UP, int ABL = 0>
static float timeit(int rows, int cols, int batch, int nt, const void* src, float* dst, float* dec, size_t src_zs, size_t zs) {
    float k[2 * R + 1];
    gauss(R, k);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    blur_march_launch<R, W, UP, ABL>(0, src, dst, dec, rows, cols, cols / 2, k, batch, src_zs, zs, nt);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) blur_march_launch<R, W, UP, ABL>(0, src, dst, dec, rows, cols, cols / 2, k, batch, src_zs, zs, nt);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / 3;
}

template <int R>
static int run_radius(float* src, float* tmp, float* ref, float* dst, float* dec, uint8_t* u8) {
    int fails = 0;
    const int shapes[][2] = {{123, 157}, {240, 320}, {66, 64}, {960, 1280}, {1000, 1004}, {34, 36}, {2048, 4096}};
    for (auto& s : shapes) {
        if (s[0] <= R || s[1] <= R) continue;
        fails += check<R, 64>(s[0], s[1], 1, src, tmp, ref, dst, dec, u8);
        fails += check<R, 64>(s[0], s[1], 3, src, tmp, ref, dst, dec, u8);
        fails += check<R, 128>(s[0], s[1], 2, src, tmp, ref, dst, dec, u8);
    }
    printf("R=%d exactness: %s\n", R, fails ? "FAILED" : "ok");
    const int n = 16384;
    const double gb = (double)n * n * 8 / 1e9;
    for (int nt : {4, 8, 16, 32}) {
        const float a = timeit<R, 64, false>(n, n, 1, nt, src, dst, nullptr, 0, 0);
        const float b = timeit<R, 128, false>(n, n, 1, nt, src, dst, nullptr, 0, 0);
        printf("R=%2d 16384^2 nt=%2d  W64 %.3f ms (%.2f TB/s of 8 B/px)  W128 %.3f ms (%.2f TB/s)\n", R, nt, a, gb / a, b, gb / b);
    }
    {
        const float a1 = timeit<R, 64, false, 1>(n, n, 1, 16, src, dst, nullptr, 0, 0);
        const float a2 = timeit<R, 64, false, 2>(n, n, 1, 16, src, dst, nullptr, 0, 0);
        const float a3 = timeit<R, 64, false, 3>(n, n, 1, 16, src, dst, nullptr, 0, 0);
        printf("R=%2d 16384^2 nt=16  W64 ablation: centre-tap row pass %.3f | centre-tap column pass %.3f | both %.3f ms\n", R, a1, a2, a3);
    }
    {
        const float a = timeit<R, 64, false>(n, n, 1, 8, src, dst, dec, 0, 0);
        printf("R=%2d 16384^2 nt= 8  W64 + decimated output %.3f ms\n", R, a);
        const float u = timeit<R, 64, true>(n, n, 1, 8, u8, dst, nullptr, 0, 0);
        const float u2 = timeit<R, 128, true>(n, n, 1, 8, u8, dst, nullptr, 0, 0);
        printf("R=%2d 16384^2 nt= 8  u8-upsampled source W64 %.3f ms  W128 %.3f ms\n", R, u, u2);
    }
    for (int m : {8192, 4096, 2048}) {
        printf("R=%2d %5d^2 W64:", R, m);
        for (int nt : {1, 2, 4, 8, 16}) printf("  nt=%d %.4f", nt, timeit<R, 64, false>(m, m, 1, nt, src, dst, nullptr, 0, 0));
        printf(" ms\n");
    }
    const size_t zs = (size_t)960 * 1280 * 4;
    for (int nt : {1, 2, 4}) {
        const float a = timeit<R, 64, false>(960, 1280, 32, nt, src, dst, nullptr, zs, zs);
        const float b = timeit<R, 64, false>(960, 1280, 1, nt, src, dst, nullptr, zs, zs);
        printf("R=%2d 960x1280 nt=%d  x32 %.4f ms   x1 %.4f ms\n", R, nt, a, b);
    }
    return fails;
}

// the five blurs of an octave back to back, layers `gap` bytes apart in one allocation (the product's arena layout)
template <int R>
static void one(const float* a, float* b, int n) {
    float k[2 * R + 1];
    gauss(R, k);
    blur_march_launch<R, 64, false>(0, a, b, nullptr, n, n, 0, k, 1, 0, 0, 0);
}
static void pipeline(size_t gap_extra, bool sync_between, bool image_like = false) {
    const int n = 16384;
    const size_t px = (size_t)n * n, gap = px * 4 + gap_extra;
    char* arena;
    CK(hipMalloc(&arena, gap * 6));
    if (image_like) hipLaunchKernelGGL(fill_image, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, 0, (float*)arena, n, px, 3u);
    else hipLaunchKernelGGL(fill_f32, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, 0, (float*)arena, px, 3u);
    hipEvent_t e[6];
    for (auto& x : e) CK(hipEventCreate(&x));
    auto L = [&](int i) { return (float*)(arena + gap * i); };
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e[0]));
        one<5>(L(0), L(1), n); CK(hipEventRecord(e[1])); if (sync_between) CK(hipDeviceSynchronize());
        one<6>(L(1), L(2), n); CK(hipEventRecord(e[2])); if (sync_between) CK(hipDeviceSynchronize());
        one<8>(L(2), L(3), n); CK(hipEventRecord(e[3])); if (sync_between) CK(hipDeviceSynchronize());
        one<10>(L(3), L(4), n); CK(hipEventRecord(e[4])); if (sync_between) CK(hipDeviceSynchronize());
        one<13>(L(4), L(5), n); CK(hipEventRecord(e[5]));
        CK(hipDeviceSynchronize());
        float t[5];
        for (int i = 0; i < 5; ++i) CK(hipEventElapsedTime(&t[i], e[i], e[i + 1]));
        if (image_like) printf("image-like ");
        printf("pipeline gap 2^30+%zu %s: R5 %.3f  R6 %.3f  R8 %.3f  R10 %.3f  R13 %.3f ms\n", gap_extra,
               sync_between ? "host sync between" : "queued back to back", t[0], t[1], t[2], t[3], t[4]);
    }
    CK(hipFree(arena));
}

int main(int argc, char** argv) {
    if (argc > 1) {
        CK(hipMalloc(&d_bad, 8));
        pipeline(0, true);
        pipeline(0, false);
        pipeline((1 << 20) + 8192 + 256, true);
        pipeline((1 << 20) + 8192 + 256, false);
        pipeline((1 << 20) + 8192 + 256, false, true);
        return 0;
    }
    const int n = 16384;
    const size_t px = (size_t)n * n;
    float *src, *tmp, *ref, *dst, *dec;
    uint8_t* u8;
    CK(hipMalloc(&src, px * 4)); CK(hipMalloc(&tmp, px * 4)); CK(hipMalloc(&ref, px * 4)); CK(hipMalloc(&dst, px * 4));
    CK(hipMalloc(&dec, px)); CK(hipMalloc(&u8, px / 4)); CK(hipMalloc(&d_bad, 8));
    int fails = 0;
    fails += run_radius<5>(src, tmp, ref, dst, dec, u8);
    fails += run_radius<6>(src, tmp, ref, dst, dec, u8);
    fails += run_radius<8>(src, tmp, ref, dst, dec, u8);
    fails += run_radius<10>(src, tmp, ref, dst, dec, u8);
    fails += run_radius<13>(src, tmp, ref, dst, dec, u8);
    fails += check<13, 64>(n, n, 16, src, tmp, ref, dst, dec, u8);
    fails += check<5, 64>(n, n, 16, src, tmp, ref, dst, dec, u8);
    printf("16384^2 exactness (R=13, R=5): %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
