// AKAZE extraction: the detector the checked-in reference runs, match_finder.py:210-211,220
//   cv2.AKAZE_create(descriptor_type=AKAZE_DESCRIPTOR_KAZE, threshold=0.0005, nOctaves=4, nOctaveLayers=5,
//                    diffusivity=KAZE_DIFF_PM_G2).detectAndCompute(img, None)
// "Spec A" (oracle/akaze_oracle.c) stage by stage. The nonlinear scale space is a chain of small f32 stencils over
// planes resident in HBM (HBM-bound; 4 B read + 4 B write per pixel and stage, 12 B per FED step):
//   gauss_kernel<N>      N x N Gaussian, BORDER_REPLICATE (9 x 9 from the u8 image, 5 x 5 per level)
//   flow_kernel          Scharr pair of Lsmooth -> conductivity g2 (or |grad| + its maximum, for the contrast factor)
//   fed_kernel           one explicit diffusion step Lt' = Lt + tau/2 * div(g grad Lt)
//   area_kernel          resize(INTER_AREA) at a new octave (2 x 2 boxes, or the general decimation table)
//   deriv1_kernel        Lx, Ly of Lsmooth by the Scharr pair stretched to +- sigma_size
//   det_kernel           Lxx, Lxy, Lyy from Lx, Ly -> Ldet
//   candidates_kernel    strict 3 x 3 maxima above the threshold inside the level's border, with their 3 x 3 response
// Every stencil stages a 64 x 16 tile plus halo in LDS (coalesced 256-byte row pieces), forms the row pass into a
// second LDS plane and the column pass from it: one ascending-tap fma chain per pass, the oracle's order, so the
// planes are bit-identical to the oracle's. The candidate filtering of A4 is an order-dependent scan (first neighbour
// in row-major order wins) over a few thousand points: it runs on the host on the downloaded candidates, as does the
// 2 x 2 sub-pixel solve; orientation and the M-SURF descriptor run one wavefront per keypoint.
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <new>
#include <vector>
#include "common.h"
#include "preproc.h"

#define AK_MAX_LEVELS 32
#define AK_MAX_STEPS 256
#define AK_TW 64
#define AK_TH 16
#define AK_MAXR 6
#define AK_CAND_FLOATS 12   // level, x, y, 9 responses

struct ak_level {
    int rows, cols, octave, sublevel, sigma_size, border, nsteps;
    float esigma, etime, ratio;
    float tau[AK_MAX_STEPS];
    float *Lt, *Lsmooth, *Lx, *Ly, *Ldet;   // device planes
};

struct imatch_akaze_state {
    char* arena;
    size_t arena_bytes;
    int h, w, omax, nsub;
    int n_levels;
    ak_level lv[AK_MAX_LEVELS];
    float *tmpA, *tmpB;      // full-size scratch planes
    uint8_t* img;            // (CLAHE) input
    void* clahe_scratch;
    unsigned* hmax;          // device: bits of the largest gradient modulus
    int* hist;               // device: 300 bins
    float* kc;               // device: contrast factor per octave
    float* tau_dev;          // device [levels][AK_MAX_STEPS]: the FED cycles
    int* n_cand;             // device counter
    float* cand;             // device [cap][AK_CAND_FLOATS]; its own allocation, grown when a scan overflows it
    int cand_cap;
    int cand_hint;           // candidates of the previous call + a margin: how many records the first download takes
    int early_hint;          // the same for the candidates of the early levels (see akaze_run)
    hipStream_t side;        // default priority; the early levels' response and candidates run here beside the small levels
    hipEvent_t ev_fork, ev_early;
    float* kp_dev;           // device [kp_cap][8]: x, y, size, angle, response, octave, level, pad; own allocation
    float* desc_dev;         // device [kp_cap][64], behind kp_dev
    int kp_cap;
    // last result (host)
    std::vector<float> kp, desc;
    int n_kp;
    uint64_t serial;         // number of the extraction whose descriptors sit in desc_dev (0: none); see imatch_akaze_last
};

void imatch_akaze_state_free(imatch_ctx* ctx) {
    if (!ctx || !ctx->akaze) return;
    if (ctx->akaze->arena) (void)hipFree(ctx->akaze->arena);
    if (ctx->akaze->cand) (void)hipFree(ctx->akaze->cand);
    if (ctx->akaze->kp_dev) (void)hipFree(ctx->akaze->kp_dev);
    if (ctx->akaze->side) {
        (void)hipStreamSynchronize(ctx->akaze->side);
        (void)hipStreamDestroy(ctx->akaze->side);
        (void)hipEventDestroy(ctx->akaze->ev_fork);
        (void)hipEventDestroy(ctx->akaze->ev_early);
    }
    delete ctx->akaze;
    ctx->akaze = nullptr;
}

// ---- fed.cpp (host) ------------------------------------------------------------------------------------------------
static bool is_prime(int n) {
    if (n <= 1) return false;
    if (n == 2 || n == 3 || n == 5 || n == 7) return true;
    if (n % 2 == 0 || n % 3 == 0 || n % 5 == 0 || n % 7 == 0) return false;
    const int upper = (int)(sqrt((double)n + 1.0));
    for (int d = 11; d <= upper; d += 2)
        if (n % d == 0) return false;
    return true;
}

static int fed_tau(float T, float tau_max, float* tau) {
    const int n = (int)ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f);
    if (n <= 0) return 0;
    if (n > AK_MAX_STEPS) return -1;
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    float tauh[AK_MAX_STEPS];
    const float c = 1.0f / (4.0f * (float)n + 2.0f), d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) {
        const float h = cosf((float)M_PI * (2.0f * (float)k + 1.0f) * c);
        tauh[k] = d / (h * h);
    }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}

// ---- tile helpers ----------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ak_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
__device__ __forceinline__ int ak_border(int p, int len, int reflect) {
    return reflect ? ak_reflect101(p, len) : min(max(p, 0), len - 1);
}

struct ak_taps {
    int n;
    int off[9];
    float k[9];
};

#define AK_IN_W (AK_TW + 2 * AK_MAXR + 1)
#define AK_IN_H (AK_TH + 2 * AK_MAXR)

// stage the tile of `src` around (x0, y0) with halo R into LDS (f32; u8 sources are scaled by 1/255 as convertTo does)
template <typename T>
__device__ __forceinline__ void ak_load_tile(const T* __restrict__ src, int rows, int cols, int x0, int y0, int R, int reflect,
                                             float (*in)[AK_IN_W]) {
    const int tw = AK_TW + 2 * R, th = AK_TH + 2 * R;
    for (int i = threadIdx.x; i < tw * th; i += 256) {
        const int ly = i / tw, lx = i - ly * tw;
        const int gy = ak_border(y0 - R + ly, rows, reflect), gx = ak_border(x0 - R + lx, cols, reflect);
        const T v = src[(size_t)gy * cols + gx];
        in[ly][lx] = sizeof(T) == 1 ? (float)v * (float)(1.0 / 255.0) : (float)v;
    }
}

// row pass of the staged tile: tmp[ly][x] for ly in [0, TH + 2R), x in [0, TW)
__device__ __forceinline__ void ak_row_pass(const float (*in)[AK_IN_W], int R, const ak_taps& kx, float (*tmp)[AK_TW + 1]) {
    const int th = AK_TH + 2 * R;
    for (int i = threadIdx.x; i < th * AK_TW; i += 256) {
        const int ly = i >> 6, x = i & 63;
        float acc = kx.k[0] * in[ly][x + R + kx.off[0]];
        for (int t = 1; t < kx.n; ++t) acc = fmaf(kx.k[t], in[ly][x + R + kx.off[t]], acc);
        tmp[ly][x] = acc;
    }
}

__device__ __forceinline__ float ak_col_pass(const float (*tmp)[AK_TW + 1], int R, const ak_taps& ky, int y, int x) {
    float acc = ky.k[0] * tmp[y + R + ky.off[0]][x];
    for (int t = 1; t < ky.n; ++t) acc = fmaf(ky.k[t], tmp[y + R + ky.off[t]][x], acc);
    return acc;
}

// ---- kernels -----------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
ak_gauss_kernel(const T* __restrict__ src, int rows, int cols, ak_taps k, float* __restrict__ dst, float* __restrict__ dst2) {
    __shared__ float in[AK_IN_H][AK_IN_W];
    __shared__ float tmp[AK_IN_H][AK_TW + 1];
    const int x0 = blockIdx.x * AK_TW, y0 = blockIdx.y * AK_TH, R = k.n / 2;
    ak_load_tile(src, rows, cols, x0, y0, R, 0, in);
    __syncthreads();
    ak_row_pass(in, R, k, tmp);
    __syncthreads();
    const int x = threadIdx.x & 63;
    if (x0 + x >= cols) return;
    for (int y = threadIdx.x >> 6; y < AK_TH && y0 + y < rows; y += 4) {
        const float v = ak_col_pass(tmp, R, k, y, x);
        dst[(size_t)(y0 + y) * cols + x0 + x] = v;
        if (dst2) dst2[(size_t)(y0 + y) * cols + x0 + x] = v;
    }
}

// Scharr pair (unnormalised [-1 0 1] x [3 10 3], BORDER_REFLECT_101). mode 0: conductivity g2 with k = kc[octave];
// mode 1: gradient modulus of the interior pixels, and its maximum (all moduli are >= 0: their bit patterns order).
__global__ void __launch_bounds__(256)
ak_flow_kernel(const float* __restrict__ src, int rows, int cols, const float* __restrict__ kc, int octave, int mode,
               float* __restrict__ dst, unsigned* __restrict__ hmax) {
    __shared__ float in[AK_TH + 2][AK_TW + 2 + 1];
    const int x0 = blockIdx.x * AK_TW, y0 = blockIdx.y * AK_TH;
    for (int i = threadIdx.x; i < (AK_TH + 2) * (AK_TW + 2); i += 256) {
        const int ly = i / (AK_TW + 2), lx = i - ly * (AK_TW + 2);
        in[ly][lx] = src[(size_t)ak_reflect101(y0 - 1 + ly, rows) * cols + ak_reflect101(x0 - 1 + lx, cols)];
    }
    __syncthreads();
    const int x = threadIdx.x & 63;
    float k2inv = 0.f;
    if (mode == 0) {
        const float k = kc[octave];
        k2inv = 1.0f / (k * k);
    }
    float wmax = 0.f;
    for (int y = threadIdx.x >> 6; y < AK_TH; y += 4) {
        const int gx_ = x0 + x, gy_ = y0 + y;
        if (gx_ >= cols || gy_ >= rows) continue;
        // row passes of the three rows, then the column passes (ascending taps)
        const float d0 = in[y][x + 2] - in[y][x], d1 = in[y + 1][x + 2] - in[y + 1][x], d2 = in[y + 2][x + 2] - in[y + 2][x];
        const float gx = fmaf(3.f, d2, fmaf(10.f, d1, 3.f * d0));
        const float s0 = fmaf(3.f, in[y][x + 2], fmaf(10.f, in[y][x + 1], 3.f * in[y][x]));
        const float s2 = fmaf(3.f, in[y + 2][x + 2], fmaf(10.f, in[y + 2][x + 1], 3.f * in[y + 2][x]));
        const float gy = s2 - s0;
        if (mode == 0) {
            dst[(size_t)gy_ * cols + gx_] = 1.0f / (1.0f + ((gx * gx + gy * gy) * k2inv));
        } else {
            const float m = sqrtf(gx * gx + gy * gy);
            dst[(size_t)gy_ * cols + gx_] = m;
            if (gx_ >= 1 && gx_ < cols - 1 && gy_ >= 1 && gy_ < rows - 1) wmax = fmaxf(wmax, m);
        }
    }
    if (mode == 1) {
        for (int o = 32; o; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        if ((threadIdx.x & 63) == 0 && wmax > 0.f) atomicMax(hmax, __float_as_uint(wmax));
    }
}

__global__ void __launch_bounds__(256)
ak_hist_kernel(const float* __restrict__ modg, int rows, int cols, const unsigned* __restrict__ hmax, int nbins,
               int* __restrict__ hist) {
    __shared__ int h[512];
    for (int i = threadIdx.x; i < nbins; i += 256) h[i] = 0;
    __syncthreads();
    const float hm = __uint_as_float(*hmax);
    if (hm > 0.f) {
        const float mul = (float)(nbins - 1) / hm;
        const int total = (rows - 2) * (cols - 2);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const int y = i / (cols - 2) + 1, x = i % (cols - 2) + 1;
            atomicAdd(&h[(int)(modg[(size_t)y * cols + x] * mul)], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 256)
        if (h[i]) atomicAdd(hist + i, h[i]);
}

__global__ void ak_kc_kernel(const int* __restrict__ hist, const unsigned* __restrict__ hmax, int nbins, int total, float perc,
                             int n_octaves, float* __restrict__ kc) {
    if (threadIdx.x || blockIdx.x) return;
    const float hm = __uint_as_float(*hmax);
    float k = 0.03f;
    if (hm > 0.f) {
        const int nthreshold = (int)((total - hist[0]) * perc);
        int nelements = 0;
        for (int b = 1; b < nbins; ++b) {
            if (nelements >= nthreshold) { k = hm * b / nbins; break; }
            nelements += hist[b];
        }
    }
    for (int o = 0; o < n_octaves; ++o) {
        kc[o] = k;
        k *= 0.75f;
    }
}

__global__ void __launch_bounds__(256)
ak_fed_kernel(const float* __restrict__ Lt, const float* __restrict__ Lf, int rows, int cols, float step, float* __restrict__ out) {
    __shared__ float t[AK_TH + 2][AK_TW + 2 + 1];
    __shared__ float f[AK_TH + 2][AK_TW + 2 + 1];
    const int x0 = blockIdx.x * AK_TW, y0 = blockIdx.y * AK_TH;
    for (int i = threadIdx.x; i < (AK_TH + 2) * (AK_TW + 2); i += 256) {
        const int ly = i / (AK_TW + 2), lx = i - ly * (AK_TW + 2);
        const int gy = min(max(y0 - 1 + ly, 0), rows - 1), gx = min(max(x0 - 1 + lx, 0), cols - 1);
        t[ly][lx] = Lt[(size_t)gy * cols + gx];
        f[ly][lx] = Lf[(size_t)gy * cols + gx];
    }
    __syncthreads();
    const int x = threadIdx.x & 63, lx = x + 1;
    const int gx = x0 + x;
    if (gx >= cols) return;
    for (int y = threadIdx.x >> 6; y < AK_TH; y += 4) {
        const int gy = y0 + y, ly = y + 1;
        if (gy >= rows) break;
        const bool ex = gx == 0 || gx == cols - 1, ey = gy == 0 || gy == rows - 1;
        const float tc = t[ly][lx], fc = f[ly][lx];
        float s;
        if (ex && ey) {
            s = 0.f;
        } else if (!ex && !ey) {
            s = (fc + f[ly][lx + 1]) * (t[ly][lx + 1] - tc) + (fc + f[ly][lx - 1]) * (t[ly][lx - 1] - tc) +
                (fc + f[ly + 1][lx]) * (t[ly + 1][lx] - tc) + (fc + f[ly - 1][lx]) * (t[ly - 1][lx] - tc);
        } else if (ey) {
            const int lv = gy == 0 ? ly + 1 : ly - 1;
            s = (fc + f[ly][lx + 1]) * (t[ly][lx + 1] - tc) + (fc + f[ly][lx - 1]) * (t[ly][lx - 1] - tc) +
                (fc + f[lv][lx]) * (t[lv][lx] - tc);
        } else {
            const int ln = gx == 0 ? lx + 1 : lx - 1;
            s = (fc + f[ly][ln]) * (t[ly][ln] - tc) + (fc + f[ly + 1][lx]) * (t[ly + 1][lx] - tc) +
                (fc + f[ly - 1][lx]) * (t[ly - 1][lx] - tc);
        }
        const float stepv = s * step;
        out[(size_t)gy * cols + gx] = tc + stepv;
    }
}

// resize(INTER_AREA) of an f32 plane at a new octave: exact 2 x 2 boxes in ResizeAreaFast's vector order, otherwise the
// decimation table of resize.cpp:computeResizeAreaTab recomputed per output pixel (as preproc.hip does for u8): a
// fractional head, whole cells and a fractional tail per axis, row sums then beta-weighted rows in the table's order.
struct ak_span {
    int s_head, s1, s2, s_tail;
    float a_head, a_mid, a_tail;
};

__device__ __forceinline__ ak_span ak_area_entry(int d, double scale, int ssize) {
    ak_span e;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, ssize - f1);
    int sx1 = (int)ceil(f1), sx2 = (int)floor(f2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    e.s_head = e.s_tail = -1;
    e.a_head = e.a_tail = 0.f;
    if (sx1 - f1 > 1e-3) { e.s_head = sx1 - 1; e.a_head = (float)((sx1 - f1) / cell); }
    e.s1 = sx1; e.s2 = sx2;
    e.a_mid = (float)(1.0 / cell);
    if (f2 - sx2 > 1e-3) { e.s_tail = sx2; e.a_tail = (float)(fmin(fmin(f2 - sx2, 1.), cell) / cell); }
    return e;
}

__device__ __forceinline__ float ak_area_row(const float* __restrict__ S, const ak_span& ex) {
    float buf = 0.f;
    if (ex.s_head >= 0) buf += S[ex.s_head] * ex.a_head;
    for (int sx = ex.s1; sx < ex.s2; ++sx) buf += S[sx] * ex.a_mid;
    if (ex.s_tail >= 0) buf += S[ex.s_tail] * ex.a_tail;
    return buf;
}

// fast != 0: srows == 2 rows and scols == 2 cols exactly
__device__ __forceinline__ float ak_area_pixel(const float* __restrict__ src, int srows, int scols, int x, int y, int fast,
                                               double scale_x, double scale_y) {
    if (fast) {
        const float* s = src + (size_t)(2 * y) * scols + 2 * x;
        return ((s[0] + s[1]) + (s[scols] + s[scols + 1])) * 0.25f;
    }
    const ak_span ex = ak_area_entry(x, scale_x, scols), ey = ak_area_entry(y, scale_y, srows);
    float sum = 0.f;
    if (ey.s_head >= 0) sum += ey.a_head * ak_area_row(src + (size_t)ey.s_head * scols, ex);
    for (int sy = ey.s1; sy < ey.s2; ++sy) sum += ey.a_mid * ak_area_row(src + (size_t)sy * scols, ex);
    if (ey.s_tail >= 0) sum += ey.a_tail * ak_area_row(src + (size_t)ey.s_tail * scols, ex);
    return sum;
}

__global__ void __launch_bounds__(256)
ak_area_kernel(const float* __restrict__ src, int srows, int scols, float* __restrict__ dst, int rows, int cols, int fast,
               double scale_x, double scale_y) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    dst[(size_t)y * cols + x] = ak_area_pixel(src, srows, scols, x, y, fast, scale_x, scale_y);
}

// One whole evolution level of a SMALL plane (rows * cols <= AK_SMALL_PX: 120 x 160 and below, i.e. octaves 2 and 3 of a
// 640 x 480 frame, which hold 139 of the 192 diffusion steps) in ONE workgroup: the level's image and its conductivity
// stay in LDS (2 x 76 KB) across the start (copy or halving), the 5 x 5 Gaussian, the Scharr pair and every FED step.
// Replaces 3 + nsteps launches of a few microseconds of work each; same expressions, so the planes are unchanged.
#define AK_SMALL_PX 19200
#define AK_SMALL_LDS_FLOATS 40000   // 2 px + 4 cols must fit: 160 000 bytes of the CU's 160 KB
#define AK_SMALL_PT 19      // pixels per thread at 1024 threads
typedef float ak_f2 __attribute__((ext_vector_type(2)));
template <bool PAIRS>   // PAIRS (even cols): a thread steps two neighbouring pixels at a time, 64-bit LDS accesses and packed f32 math
__global__ void __launch_bounds__(1024)
ak_level_small_kernel(const float* __restrict__ prev, int prows, int pcols, int halve, int fast, double scale_x, double scale_y,
                      int rows, int cols, ak_taps g5,
                      const float* __restrict__ kc, int octave, int nsteps, const float* __restrict__ tau,
                      float* __restrict__ Lt, float* __restrict__ Lsmooth) {
    extern __shared__ float ak_lds[];
    const int px = rows * cols;
    // two planes, each with a guard row of `cols` zeros before and after it, so that the step below can read all four
    // neighbours of any pixel without clamping (a dropped neighbour has a zero coefficient)
    float* A = ak_lds + cols;
    float* F = ak_lds + px + 3 * cols;
    const int tid = threadIdx.x;
    for (int i = tid; i < cols; i += 1024) {
        ak_lds[i] = 0.f; ak_lds[px + cols + i] = 0.f; ak_lds[px + 2 * cols + i] = 0.f; ak_lds[2 * px + 3 * cols + i] = 0.f;
    }
    for (int p = tid; p < px; p += 1024) {
        const int y = p / cols, x = p - y * cols;
        if (halve) {
            A[p] = ak_area_pixel(prev, prows, pcols, x, y, fast, scale_x, scale_y);
        } else {
            A[p] = prev[p];
        }
    }
    __syncthreads();
    for (int p = tid; p < px; p += 1024) {   // Gaussian row pass -> F
        const int y = p / cols, x = p - y * cols;
        float acc = g5.k[0] * A[y * cols + min(max(x + g5.off[0], 0), cols - 1)];
        for (int t = 1; t < g5.n; ++t) acc = fmaf(g5.k[t], A[y * cols + min(max(x + g5.off[t], 0), cols - 1)], acc);
        F[p] = acc;
    }
    __syncthreads();
    for (int p = tid; p < px; p += 1024) {   // column pass -> Lsmooth (global: the derivative kernels read it too)
        const int y = p / cols, x = p - y * cols;
        float acc = g5.k[0] * F[min(max(y + g5.off[0], 0), rows - 1) * cols + x];
        for (int t = 1; t < g5.n; ++t) acc = fmaf(g5.k[t], F[min(max(y + g5.off[t], 0), rows - 1) * cols + x], acc);
        Lsmooth[p] = acc;
    }
    __threadfence_block();
    __syncthreads();
    {
        const float k = kc[octave];
        const float k2inv = 1.0f / (k * k);
        for (int p = tid; p < px; p += 1024) {   // Scharr pair of Lsmooth -> conductivity in F
            const int y = p / cols, x = p - y * cols;
            const int xm = ak_reflect101(x - 1, cols), xp = ak_reflect101(x + 1, cols);
            const float* r0 = Lsmooth + (size_t)ak_reflect101(y - 1, rows) * cols;
            const float* r1 = Lsmooth + (size_t)y * cols;
            const float* r2 = Lsmooth + (size_t)ak_reflect101(y + 1, rows) * cols;
            const float d0 = r0[xp] - r0[xm], d1 = r1[xp] - r1[xm], d2 = r2[xp] - r2[xm];
            const float gx = fmaf(3.f, d2, fmaf(10.f, d1, 3.f * d0));
            const float s0 = fmaf(3.f, r0[xp], fmaf(10.f, r0[x], 3.f * r0[xm]));
            const float s2 = fmaf(3.f, r2[xp], fmaf(10.f, r2[x], 3.f * r2[xm]));
            const float gy = s2 - s0;
            F[p] = 1.0f / (1.0f + ((gx * gx + gy * gy) * k2inv));
        }
    }
    __syncthreads();
    // The conductivity does not change during the cycle: each thread keeps the four sums (g_c + g_n) of its pixels in
    // registers (0 where OpenCV's edge rules drop the neighbour, all four 0 at the corners: the dropped terms are then
    // exact zeros at the places the source leaves them out) and F's LDS becomes the second image plane, so a step is
    // five LDS reads per pixel and one barrier.
    if (PAIRS) {
        // pixel pair q = tid + 1024 j holds pixels 2q, 2q + 1 (same row: cols is even). Per step and pair: three 64-bit and
        // two 32-bit LDS reads, one 64-bit write and 13 packed instructions -- the scalar form's subtractions, products and
        // sums, in its order, for both pixels at once (the scalar form: ten reads, two writes, 26 instructions).
        constexpr int NPAIR = (AK_SMALL_PT + 1) / 2;
        ak_f2 cR[NPAIR], cL[NPAIR], cD[NPAIR], cU[NPAIR];
        const int npair = px >> 1, hc = cols >> 1;
        {
            const int qs = 1024 / hc, rs = 1024 - qs * hc;   // (y, pair column) of pair tid + 1024 j by stepping
            int y = tid / hc, xh = tid - y * hc;
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                const int q = min(tid + j * 1024, npair - 1);
                const int pc = 2 * q;
                const bool y0 = y == 0, y1 = y == rows - 1;
                const bool xa0 = xh == 0, xb1 = xh == hc - 1;          // first pixel at x = 0, second at x = cols - 1
                const float fa = F[pc], fb = F[pc + 1];
                const float ra = fa + fb, la = fa + F[pc - 1], da = fa + F[pc + cols], ua = fa + F[pc - cols];
                const float rb = fb + F[pc + 2], lb = fb + fa, db = fb + F[pc + 1 + cols], ub = fb + F[pc + 1 - cols];
                const bool ca = xa0 && (y0 || y1), cb = xb1 && (y0 || y1);
                cR[j] = ak_f2{ca ? 0.f : ra, (xb1 || cb) ? 0.f : rb};
                cL[j] = ak_f2{(xa0 || ca) ? 0.f : la, cb ? 0.f : lb};
                cD[j] = ak_f2{(y1 || ca) ? 0.f : da, (y1 || cb) ? 0.f : db};
                cU[j] = ak_f2{(y0 || ca) ? 0.f : ua, (y0 || cb) ? 0.f : ub};
                xh += rs; y += qs;
                if (xh >= hc) { xh -= hc; ++y; }
                asm volatile("" ::: "memory");
            }
        }
        __syncthreads();
        float* B = F;
        for (int st = 0; st < nsteps; ++st) {
            const float stepf = tau[st] * 0.5f;
            const ak_f2 step = {stepf, stepf};
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                const int q = tid + j * 1024;
                if (q < npair) {
                    const int p = 2 * q;
                    const ak_f2 tc = *(const ak_f2*)(A + p);
                    const ak_f2 td = *(const ak_f2*)(A + p + cols), tu = *(const ak_f2*)(A + p - cols);
                    const ak_f2 tr = {tc.y, A[p + 2]}, tl = {A[p - 1], tc.x};
                    const ak_f2 sum = cR[j] * (tr - tc) + cL[j] * (tl - tc) + cD[j] * (td - tc) + cU[j] * (tu - tc);
                    const ak_f2 stepv = sum * step;
                    *(ak_f2*)(B + p) = tc + stepv;
                }
            }
            __syncthreads();
            float* t_ = A; A = B; B = t_;
        }
    } else {
        float cR[AK_SMALL_PT], cL[AK_SMALL_PT], cD[AK_SMALL_PT], cU[AK_SMALL_PT];
        {
            // (y, x) of pixel tid + 1024 j by stepping: 1024 = qs * cols + rs
            const int qs = 1024 / cols, rs = 1024 - qs * cols;
            int y = tid / cols, x = tid - y * cols;
#pragma unroll
            for (int j = 0; j < AK_SMALL_PT; ++j) {
                const int p = tid + j * 1024;
                const int pc = min(p, px - 1);   // threads beyond the plane compute on its last pixel and never store
                const bool x0 = x == 0, x1 = x == cols - 1, y0 = y == 0, y1 = y == rows - 1;
                const bool corner = (x0 || x1) && (y0 || y1);
                const float fc = F[pc];
                // F has guard rows too: the loads need no conditions, the selects drop what OpenCV's edge rules drop
                const float r = fc + F[pc + 1], l = fc + F[pc - 1], d = fc + F[pc + cols], u = fc + F[pc - cols];
                cR[j] = (x1 || corner) ? 0.f : r;
                cL[j] = (x0 || corner) ? 0.f : l;
                cD[j] = (y1 || corner) ? 0.f : d;
                cU[j] = (y0 || corner) ? 0.f : u;
                x += rs; y += qs;
                if (x >= cols) { x -= cols; ++y; }
                asm volatile("" ::: "memory");   // one pixel's operands in flight at a time: 76 registers hold the sums
            }
        }
        __syncthreads();
        float* B = F;
        for (int st = 0; st < nsteps; ++st) {
            const float step = tau[st] * 0.5f;
#pragma unroll
            for (int j = 0; j < AK_SMALL_PT; ++j) {
                const int p = tid + j * 1024;
                if (p < px) {
                    // a dropped neighbour has a zero coefficient: whatever finite value sits there stands in for it
                    const float tc = A[p];
                    const float tr = A[p + 1], tl = A[p - 1], td = A[p + cols], tu = A[p - cols];
                    const float sum = cR[j] * (tr - tc) + cL[j] * (tl - tc) + cD[j] * (td - tc) + cU[j] * (tu - tc);
                    const float stepv = sum * step;
                    B[p] = tc + stepv;
                }
            }
            __syncthreads();
            float* t_ = A; A = B; B = t_;
        }
    }
    for (int p = tid; p < px; p += 1024) Lt[p] = A[p];
}

// The detector response of ALL levels in three launches (blockIdx.z = level; blocks beyond a level's tiles leave): the
// levels are independent here, and a launch per level and stage was 60 dependent launches of a few microseconds.
struct ak_levels_all {
    int n;
    int rows[AK_MAX_LEVELS], cols[AK_MAX_LEVELS], s[AK_MAX_LEVELS], border[AK_MAX_LEVELS];
    const float* Lsmooth[AK_MAX_LEVELS];
    float* Lx[AK_MAX_LEVELS];
    float* Ly[AK_MAX_LEVELS];
    float* Ldet[AK_MAX_LEVELS];
};

// the Scharr pair stretched to +- s (compute_derivative_kernels): [-1 0 1] and [n, w n, n], n = 1 / (2 s (w + 2)), w = 10/3
__device__ __forceinline__ void ak_deriv_taps(int s, ak_taps& kd, ak_taps& ks) {
    const float w = 10.0f / 3.0f, norm = 1.0f / (2.0f * s * (w + 2.0f));
    kd.n = 2; kd.off[0] = -s; kd.off[1] = s; kd.k[0] = -1.f; kd.k[1] = 1.f;
    ks.n = 3; ks.off[0] = -s; ks.off[1] = 0; ks.off[2] = s; ks.k[0] = norm; ks.k[1] = w * norm; ks.k[2] = norm;
}

// Lx = D_x(Lsmooth), Ly = D_y(Lsmooth) (sepFilter2D, BORDER_REFLECT_101)
__global__ void __launch_bounds__(256)
ak_deriv1_kernel(ak_levels_all L, int z0) {
    __shared__ float in[AK_IN_H][AK_IN_W];
    __shared__ float tmp[AK_IN_H][AK_TW + 1];
    const int lv = blockIdx.z + z0;
    const int rows = L.rows[lv], cols = L.cols[lv], s = L.s[lv];
    const int x0 = blockIdx.x * AK_TW, y0 = blockIdx.y * AK_TH;
    if (x0 >= cols || y0 >= rows) return;
    ak_taps kd, ks;
    ak_deriv_taps(s, kd, ks);
    float* __restrict__ Lx = L.Lx[lv];
    float* __restrict__ Ly = L.Ly[lv];
    ak_load_tile(L.Lsmooth[lv], rows, cols, x0, y0, s, 1, in);
    __syncthreads();
    const int x = threadIdx.x & 63;
    const bool live = x0 + x < cols;
    ak_row_pass(in, s, kd, tmp);
    __syncthreads();
    if (live)
        for (int y = threadIdx.x >> 6; y < AK_TH && y0 + y < rows; y += 4)
            Lx[(size_t)(y0 + y) * cols + x0 + x] = ak_col_pass(tmp, s, ks, y, x);
    __syncthreads();
    ak_row_pass(in, s, ks, tmp);
    __syncthreads();
    if (live)
        for (int y = threadIdx.x >> 6; y < AK_TH && y0 + y < rows; y += 4)
            Ly[(size_t)(y0 + y) * cols + x0 + x] = ak_col_pass(tmp, s, kd, y, x);
}

// Ldet = (D_x(Lx) D_y(Ly) - D_y(Lx)^2) s^4
__global__ void __launch_bounds__(256)
ak_det_kernel(ak_levels_all L, int z0) {
    __shared__ float in[AK_IN_H][AK_IN_W];
    __shared__ float tmp[AK_IN_H][AK_TW + 1];
    const int lv = blockIdx.z + z0;
    const int rows = L.rows[lv], cols = L.cols[lv], s = L.s[lv];
    const int x0 = blockIdx.x * AK_TW, y0 = blockIdx.y * AK_TH;
    if (x0 >= cols || y0 >= rows) return;
    ak_taps kd, ks;
    ak_deriv_taps(s, kd, ks);
    const int x = threadIdx.x & 63, yb = threadIdx.x >> 6;
    float lxx[4], lxy[4], lyy[4];
    ak_load_tile((const float*)L.Lx[lv], rows, cols, x0, y0, s, 1, in);
    __syncthreads();
    ak_row_pass(in, s, kd, tmp);
    __syncthreads();
    for (int j = 0; j < 4; ++j) lxx[j] = ak_col_pass(tmp, s, ks, yb + 4 * j, x);
    __syncthreads();
    ak_row_pass(in, s, ks, tmp);
    __syncthreads();
    for (int j = 0; j < 4; ++j) lxy[j] = ak_col_pass(tmp, s, kd, yb + 4 * j, x);
    __syncthreads();
    ak_load_tile((const float*)L.Ly[lv], rows, cols, x0, y0, s, 1, in);
    __syncthreads();
    ak_row_pass(in, s, ks, tmp);
    __syncthreads();
    for (int j = 0; j < 4; ++j) lyy[j] = ak_col_pass(tmp, s, kd, yb + 4 * j, x);
    const float s4 = (float)(s * s * s * s);
    float* __restrict__ Ldet = L.Ldet[lv];
    if (x0 + x < cols)
        for (int j = 0; j < 4; ++j) {
            const int gy = y0 + yb + 4 * j;
            if (gy < rows) Ldet[(size_t)gy * cols + x0 + x] = (lxx[j] * lyy[j] - lxy[j] * lxy[j]) * s4;
        }
}

// A workgroup scans a 64 x 32 window (8 rows per thread) and reserves the slots of ALL its hits with one global atomic:
// atomics on the single counter complete at about one per 20 ns, and a launch finds tens of thousands of candidates.
#define AK_CAND_ROWS 8
__global__ void __launch_bounds__(256)
ak_candidates_kernel(ak_levels_all L, int z0, float thr, int* __restrict__ n_cand, float* __restrict__ cand, int cap) {
    __shared__ int blk_count, blk_base;
    const int level = blockIdx.z + z0;
    const int rows = L.rows[level], cols = L.cols[level], border = L.border[level];
    const float* __restrict__ Ldet = L.Ldet[level];
    if (border + 1 >= rows) return;   // "if border is too big we shouldn't search any keypoints" (uniform per block)
    const int x = border + blockIdx.x * 64 + (threadIdx.x & 63);
    const int yb = border + blockIdx.y * (4 * AK_CAND_ROWS) + (threadIdx.x >> 6) * AK_CAND_ROWS;
    if ((int)blockIdx.x * 64 >= cols - 2 * border || (int)blockIdx.y * (4 * AK_CAND_ROWS) >= rows - 2 * border) return;   // uniform
    if (threadIdx.x == 0) blk_count = 0;
    __syncthreads();
    unsigned hits = 0;
    if (x < cols - border)
        for (int r = 0; r < AK_CAND_ROWS; ++r) {
            const int y = yb + r;
            if (y >= rows - border) break;
            const float* c = Ldet + (size_t)y * cols + x;
            const float v = *c;
            if (v > thr && v > c[-1] && v > c[1] && v > c[-cols - 1] && v > c[-cols] && v > c[-cols + 1] && v > c[cols - 1] &&
                v > c[cols] && v > c[cols + 1])
                hits |= 1u << r;
        }
    int mine = 0;
    const int n_mine = __popc(hits);
    if (n_mine) mine = atomicAdd(&blk_count, n_mine);   // LDS
    __syncthreads();
    if (threadIdx.x == 0 && blk_count) blk_base = atomicAdd(n_cand, blk_count);
    __syncthreads();
    int slot = blk_base + mine;
    for (int r = 0; r < AK_CAND_ROWS; ++r) {
        if (!(hits >> r & 1)) continue;
        if (slot < cap) {
            const int y = yb + r;
            const float* c = Ldet + (size_t)y * cols + x;
            float* o = cand + (size_t)slot * AK_CAND_FLOATS;
            o[0] = (float)level; o[1] = (float)x; o[2] = (float)y;
            o[3] = c[-cols - 1]; o[4] = c[-cols]; o[5] = c[-cols + 1];
            o[6] = c[-1]; o[7] = *c; o[8] = c[1];
            o[9] = c[cols - 1]; o[10] = c[cols]; o[11] = c[cols + 1];
        }
        ++slot;
    }
}

// ---- orientation + descriptor: one wavefront per keypoint -----------------------------------------------------------
struct ak_levels_dev {
    int n;
    int rows[AK_MAX_LEVELS], cols[AK_MAX_LEVELS];
    float ratio[AK_MAX_LEVELS];
    const float* Lx[AK_MAX_LEVELS];
    const float* Ly[AK_MAX_LEVELS];
};

__device__ __forceinline__ float ak_angle_deg(float y, float x) {
    float a = atan2f(y, x) * (float)(180.0 / M_PI);
    if (a < 0) a += 360.f;
    if (a >= 360.f) a -= 360.f;
    return a;
}

__constant__ signed char ak_ori_di[112], ak_ori_dj[112];
__constant__ float ak_ori_w[112];

__global__ void __launch_bounds__(256)
ak_orient_kernel(float* __restrict__ kp, int n, ak_levels_dev L) {
    __shared__ float resX[4][112], resY[4][112];
    __shared__ unsigned char bin[4][112], order[4][112];
    __shared__ unsigned char start[4][44];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + w;
    const bool live = i < n;   // dead waves still reach the barriers
    float* k = kp + (size_t)(live ? i : 0) * 8;
    const int lvl = live ? (int)k[6] : 0;
    const float ratio = L.ratio[lvl];
    const int rows = L.rows[lvl], cols = L.cols[lvl];
    const int scale = __float2int_rn(0.5f * k[2] / ratio);
    const int x0 = __float2int_rn(k[0] / ratio), y0 = __float2int_rn(k[1] / ratio);
    // the 109 samples with di^2 + dj^2 < 36, di-major, with OpenSURF's gauss25 weights (table filled on the host)
    for (int r = lane; r < 109 && live; r += 64) {
        const int di = ak_ori_di[r], dj = ak_ori_dj[r];
        const int y = min(max(y0 + dj * scale, 0), rows - 1), x = min(max(x0 + di * scale, 0), cols - 1);
        const float wgt = ak_ori_w[r];
        const float rx = wgt * L.Lx[lvl][(size_t)y * cols + x], ry = wgt * L.Ly[lvl][(size_t)y * cols + x];
        resX[w][r] = rx;
        resY[w][r] = ry;
        const float ang = ak_angle_deg(ry, rx) * (float)(M_PI / 180.0);
        int b = (int)(ang / (float)(2.0 * M_PI / 42));
        if (b < 0 || b >= 42) b = 0;
        bin[w][r] = (unsigned char)b;
    }
    __syncthreads();
    if (live) {   // counting sort by slice, stable in the sample index -- every lane ranks its own samples (no serial pass)
        for (int s = lane; s < 109; s += 64) {
            const int b = bin[w][s];
            int r = 0;
            for (int j = 0; j < 109; ++j) {
                const int bj = bin[w][j];
                r += (bj < b) || (bj == b && j < s);
            }
            order[w][r] = (unsigned char)s;
        }
        if (lane < 43) {
            int c = 0;
            for (int j = 0; j < 109; ++j) c += bin[w][j] < lane;
            start[w][lane] = (unsigned char)c;
        }
    }
    __syncthreads();
    // lane sn sums the window of 7 slices starting at slice sn, in the oracle's order
    float sx = 0.f, sy = 0.f, norm = -1.f;
    if (lane < 42 && live) {
        for (int q = 0; q < 7; ++q) {
            const int b = (lane + q) % 42;   // sn..41 then 0..: the wrapped windows add the tail first, as the source does
            for (int p = start[w][b]; p < start[w][b + 1]; ++p) {
                sx += resX[w][order[w][p]];
                sy += resY[w][order[w][p]];
            }
        }
        norm = sx * sx + sy * sy;
    }
    // first window with the strictly largest norm
    float best = norm;
    for (int o = 32; o; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    const unsigned long long m = __ballot(norm == best && lane < 42);
    const int win = m ? __ffsll((long long)m) - 1 : 0;
    const float bx = __shfl(sx, win), by = __shfl(sy, win);
    if (lane == 0 && live) k[3] = ak_angle_deg(by, bx);
}

__global__ void __launch_bounds__(256)
ak_msurf_kernel(const float* __restrict__ kp, int n, ak_levels_dev L, float* __restrict__ desc) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;   // no barriers below
    const float* k = kp + (size_t)i * 8;
    const int lvl = (int)k[6], octave = (int)k[5];
    const int rows = L.rows[lvl], cols = L.cols[lvl];
    const float* Lx = L.Lx[lvl];
    const float* Ly = L.Ly[lvl];
    const float ratio = (float)(1 << octave);
    const int scale = __float2int_rn(0.5f * k[2] / ratio);
    const float angle = k[3] * (float)(M_PI / 180.f);
    const float yf = k[1] / ratio, xf = k[0] / ratio, co = cosf(angle), si = sinf(angle);
    float vals[4] = {0.f, 0.f, 0.f, 0.f};   // lane t < 64 keeps descriptor value t: (subregion t / 4, component t % 4)
    float len = 0.f;
    for (int sr = 0; sr < 16; ++sr) {
        const int bi = sr >> 2, bj = sr & 3;
        const int i0 = -12 + 5 * bi, j0 = -12 + 5 * bj;     // i = -8 - 4 + 9 b - 4 b: the source's running index
        const float cx = 0.5f + bi, cy = 0.5f + bj;
        const int ky = i0 + 5, kx = j0 + 5;
        const float xs = xf + (-kx * scale * si + ky * scale * co);
        const float ys = yf + (kx * scale * co + ky * scale * si);
        float dx = 0.f, dy = 0.f, mdx = 0.f, mdy = 0.f;
        for (int t = lane; t < 81; t += 64) {
            const int kk = i0 + t / 9, l = j0 + t % 9;
            const float sample_y = yf + (l * scale * co + kk * scale * si);
            const float sample_x = xf + (-l * scale * si + kk * scale * co);
            const float sg = 2.5f * scale;
            const float ex = xs - sample_x, ey = ys - sample_y;
            const float gauss_s1 = expf(-(ex * ex + ey * ey) / (2.0f * sg * sg));
            const int y1 = (int)floorf(sample_y), x1 = (int)floorf(sample_x);
            const int y2 = y1 + 1, x2 = x1 + 1;
            if (x1 < 0 || y1 < 0 || x2 >= cols || y2 >= rows) continue;
            const float fx = sample_x - x1, fy = sample_y - y1;
            const size_t a = (size_t)y1 * cols + x1, b = (size_t)y1 * cols + x2, c = (size_t)y2 * cols + x1, d = (size_t)y2 * cols + x2;
            const float rx = (1.0f - fx) * (1.0f - fy) * Lx[a] + fx * (1.0f - fy) * Lx[b] + (1.0f - fx) * fy * Lx[c] + fx * fy * Lx[d];
            const float ry = (1.0f - fx) * (1.0f - fy) * Ly[a] + fx * (1.0f - fy) * Ly[b] + (1.0f - fx) * fy * Ly[c] + fx * fy * Ly[d];
            const float rry = gauss_s1 * (rx * co + ry * si);
            const float rrx = gauss_s1 * (-rx * si + ry * co);
            dx += rrx; dy += rry; mdx += fabsf(rrx); mdy += fabsf(rry);
        }
        for (int o = 32; o; o >>= 1) {
            dx += __shfl_xor(dx, o); dy += __shfl_xor(dy, o); mdx += __shfl_xor(mdx, o); mdy += __shfl_xor(mdy, o);
        }
        const float ux = cx - 2.0f, uy = cy - 2.0f;
        const float g2 = expf(-(ux * ux + uy * uy) / (2.0f * 1.5f * 1.5f));
        len += (dx * dx + dy * dy + mdx * mdx + mdy * mdy) * g2 * g2;
        if ((lane >> 2) == sr) {
            const int comp = lane & 3;
            vals[0] = (comp == 0 ? dx : comp == 1 ? dy : comp == 2 ? mdx : mdy) * g2;
        }
    }
    const float len_inv = 1.0f / sqrtf(len);
    desc[(size_t)i * 64 + lane] = vals[0] * len_inv;
}

// ---- host side ---------------------------------------------------------------------------------------------------------
static void gauss_taps(int n, double sigma, ak_taps* k) {
    const int n2 = (n - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double vals[16], sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        vals[i] = exp((double)(x * x) * scale2x);
        sum += vals[i];
    }
    sum = sum * 2 + 1.0;
    const double mul1 = 1.0 / sum;
    k->n = n;
    for (int i = 0; i < n2; i++) k->k[i] = k->k[n - 1 - i] = (float)(vals[i] * mul1);
    k->k[n2] = (float)mul1;
    for (int i = 0; i < n; ++i) k->off[i] = i - n2;
}

// The candidate list and the keypoint / descriptor block are sized from the image and grow on demand (a noise image has
// up to one 3 x 3 maximum per four pixels and level).
static int akaze_grow_lists(imatch_ctx* ctx, int cand_cap, int kp_cap) {
    imatch_akaze_state* S = ctx->akaze;
    if (cand_cap > S->cand_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (S->cand) (void)hipFree(S->cand);
        S->cand = nullptr; S->cand_cap = 0;
        if (hipMalloc((void**)&S->cand, (size_t)cand_cap * AK_CAND_FLOATS * 4) != hipSuccess)
            return imatch_set_err(ctx, IMATCH_ENOMEM, "AKAZE candidate list of %d records", cand_cap);
        S->cand_cap = cand_cap;
    }
    if (kp_cap > S->kp_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (S->kp_dev) (void)hipFree(S->kp_dev);
        S->kp_dev = nullptr; S->kp_cap = 0;
        if (hipMalloc((void**)&S->kp_dev, (size_t)kp_cap * (8 + 64) * 4) != hipSuccess)
            return imatch_set_err(ctx, IMATCH_ENOMEM, "AKAZE keypoint block of %d records", kp_cap);
        S->desc_dev = S->kp_dev + (size_t)kp_cap * 8;
        S->kp_cap = kp_cap;
    }
    return IMATCH_OK;
}

static int akaze_prepare(imatch_ctx* ctx, int h, int w, int omax, int nsub) {
    imatch_akaze_state* S = ctx->akaze;
    if (S && S->h == h && S->w == w && S->omax == omax && S->nsub == nsub) return IMATCH_OK;
    if (S) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        imatch_akaze_state_free(ctx);
    }
    S = new (std::nothrow) imatch_akaze_state();
    if (!S) return imatch_set_err(ctx, IMATCH_ENOMEM, "host allocation failed");
    S->arena = nullptr; S->cand = nullptr; S->kp_dev = nullptr; S->desc_dev = nullptr; S->n_levels = 0; S->n_kp = 0;
    S->side = nullptr; S->ev_fork = nullptr; S->ev_early = nullptr; S->early_hint = 0;
    ctx->akaze = S;
    S->h = h; S->w = w; S->omax = omax; S->nsub = nsub;
    const float soffset = 1.6f, derivative_factor = 1.5f, smax = 12.0f * sqrtf(2.0f);
    for (int i = 0, power = 1; i <= omax - 1; ++i, power *= 2) {
        const float rfactor = 1.0f / power;
        const int lh = (int)(h * rfactor), lw = (int)(w * rfactor);
        if ((lw < 80 || lh < 40) && i != 0) break;
        for (int j = 0; j < nsub; ++j) {
            ak_level* e = &S->lv[S->n_levels++];
            e->rows = lh; e->cols = lw;
            e->esigma = soffset * powf(2.f, (float)j / (float)nsub + i);
            e->sigma_size = (int)lrint(e->esigma * derivative_factor / power);
            e->etime = 0.5f * (e->esigma * e->esigma);
            e->octave = i; e->sublevel = j; e->ratio = (float)power;
            e->border = (int)lrint(smax * e->sigma_size) + 1;
            e->nsteps = 0;
            if (e->sigma_size > AK_MAXR || e->sigma_size < 1) {
                imatch_akaze_state_free(ctx);
                return imatch_set_err(ctx, IMATCH_ESIZE, "AKAZE: derivative scale %d beyond the tile halo", e->sigma_size);
            }
        }
    }
    for (int i = 1; i < S->n_levels; ++i) {
        S->lv[i].nsteps = fed_tau(S->lv[i].etime - S->lv[i - 1].etime, 0.25f, S->lv[i].tau);
        if (S->lv[i].nsteps < 0) {
            imatch_akaze_state_free(ctx);
            return imatch_set_err(ctx, IMATCH_ESIZE, "AKAZE: FED cycle longer than %d steps", AK_MAX_STEPS);
        }
    }
    // arena
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t px0 = up((size_t)h * w * 4);
    size_t total = 2 * px0 + up((size_t)h * w) + up(4096 * 1280) + 4096 + up((size_t)AK_MAX_LEVELS * AK_MAX_STEPS * 4);
    for (int i = 0; i < S->n_levels; ++i) total += 5 * up((size_t)S->lv[i].rows * S->lv[i].cols * 4);
    S->cand_cap = 0;
    S->kp_cap = 0;
    if (hipMalloc((void**)&S->arena, total) != hipSuccess) {
        S->arena = nullptr;
        imatch_akaze_state_free(ctx);
        return imatch_set_err(ctx, IMATCH_ENOMEM, "AKAZE arena of %zu bytes", total);
    }
    S->arena_bytes = total;
    char* p = S->arena;
    auto take = [&](size_t b) { char* r = p; p += up(b); return r; };
    S->tmpA = (float*)take(px0); S->tmpB = (float*)take(px0);
    S->img = (uint8_t*)take((size_t)h * w);
    S->clahe_scratch = take(4096 * 1280);
    S->hmax = (unsigned*)take(256);
    S->hist = (int*)take(2048);
    S->kc = (float*)take(256);
    S->n_cand = (int*)take(256);
    S->tau_dev = (float*)take((size_t)AK_MAX_LEVELS * AK_MAX_STEPS * 4);
    for (int i = 0; i < S->n_levels; ++i) {
        ak_level* e = &S->lv[i];
        const size_t b = (size_t)e->rows * e->cols * 4;
        e->Lt = (float*)take(b); e->Lsmooth = (float*)take(b); e->Lx = (float*)take(b); e->Ly = (float*)take(b);
        e->Ldet = (float*)take(b);
    }
    const int rc = akaze_grow_lists(ctx, std::max(4096, (int)std::min<int64_t>((int64_t)h * w / 16, 1 << 26)), 4096);
    if (rc) {
        imatch_akaze_state_free(ctx);
        return rc;
    }
    for (int i = 0; i < S->n_levels; ++i)
        if (S->lv[i].nsteps > 0)
            HIP_TRY(ctx, hipMemcpy(S->tau_dev + (size_t)i * AK_MAX_STEPS, S->lv[i].tau, (size_t)S->lv[i].nsteps * 4,
                                   hipMemcpyHostToDevice));
    {
        signed char di[112] = {0}, dj[112] = {0};
        float wt[112] = {0};
        int k = 0;
        for (int i = -6; i <= 6; ++i)
            for (int j = -6; j <= 6; ++j)
                if (i * i + j * j < 36) {
                    const double g = exp(-(double)(i * i + j * j) / 12.5) / (2.0 * M_PI * 6.25);
                    di[k] = (signed char)i; dj[k] = (signed char)j;
                    wt[k++] = (float)(floor(g * 1e8 + 0.5) / 1e8);   // the table's 8 decimals
                }
        HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(ak_ori_di), di, sizeof di));
        HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(ak_ori_dj), dj, sizeof dj));
        HIP_TRY(ctx, hipMemcpyToSymbol(HIP_SYMBOL(ak_ori_w), wt, sizeof wt));
    }
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)ak_level_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     AK_SMALL_LDS_FLOATS * 4));
    HIP_TRY(ctx, hipFuncSetAttribute((const void*)ak_level_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     AK_SMALL_LDS_FLOATS * 4));
    return IMATCH_OK;
}

// Candidates of one level in scan order (row-major), as parallel arrays with a row index: A4 walks them in the order
// OpenCV's scan meets them, and its neighbour search looks at 2r rows only.
struct ak_level_pts {
    int rows = 0;
    std::vector<int> row_start;       // rows + 1
    std::vector<int> x, y;
    std::vector<float> v;             // 9 responses per point: rows y-1, y, y+1
    std::vector<uint8_t> alive;
    // the first alive point in row-major order inside [y - r, y + r) x [x - r, x + r) with squared distance <= r^2
    int find(int px, int py, int r) const {
        const int y0 = std::max(py - r, 0), y1 = std::min(py + r, rows);
        for (int yy = y0; yy < y1; ++yy)
            for (int p = row_start[yy]; p < row_start[yy + 1]; ++p) {
                if (x[p] >= px + r) break;
                if (!alive[p] || x[p] < px - r) continue;
                const int dx = x[p] - px, dy = yy - py;
                if (dx * dx + dy * dy <= r * r) return p;
            }
        return -1;
    }
};

// raw: records of AK_CAND_FLOATS in any order, ids: the `count` records of one level -> its points sorted by (y, x):
// counting sort on y, insertion by x inside a row (rows hold a handful of points). Each level is built by its own task.
static void bucket_level(const float* raw, const int* ids, int count, int rows, ak_level_pts& P) {
    P.rows = rows;
    P.row_start.assign((size_t)rows + 1, 0);
    for (int i = 0; i < count; ++i) P.row_start[(int)raw[(size_t)ids[i] * AK_CAND_FLOATS + 2] + 1]++;
    P.x.resize(count); P.y.resize(count); P.v.resize((size_t)count * 9); P.alive.assign(count, 0);
    for (int y = 0; y < rows; ++y) P.row_start[y + 1] += P.row_start[y];
    std::vector<int> fill(P.row_start.begin(), P.row_start.end() - 1);
    for (int i = 0; i < count; ++i) {
        const float* r = raw + (size_t)ids[i] * AK_CAND_FLOATS;
        const int yy = (int)r[2], xx = (int)r[1];
        int p = fill[yy]++;
        while (p > P.row_start[yy] && P.x[p - 1] > xx) {
            P.x[p] = P.x[p - 1];
            memcpy(&P.v[(size_t)p * 9], &P.v[(size_t)(p - 1) * 9], 36);
            --p;
        }
        P.x[p] = xx;
        memcpy(&P.v[(size_t)p * 9], r + 3, 36);
    }
    for (int yy = 0; yy < rows; ++yy)
        for (int p = P.row_start[yy]; p < P.row_start[yy + 1]; ++p) P.y[p] = yy;
}

static int akaze_run(imatch_ctx* ctx, const uint8_t* img_dev, int h, int w, int stride, const imatch_akaze_params* P) {
    int rc = akaze_prepare(ctx, h, w, P->n_octaves, P->n_octave_layers);
    if (rc) return rc;
    imatch_akaze_state* S = ctx->akaze;
    hipStream_t st = ctx->stream;
    const uint8_t* src = img_dev;
    int sstride = stride;
    if (P->clahe) {
        rc = imatch_clahe_dev(ctx, img_dev, h, w, stride, P->clahe_clip, P->clahe_tiles_x, P->clahe_tiles_y, S->img, S->clahe_scratch);
        if (rc) return rc;
        src = S->img;
        sstride = w;
    } else if (stride != w) {
        HIP_TRY(ctx, hipMemcpy2DAsync(S->img, w, img_dev, stride, w, h, hipMemcpyDeviceToDevice, st));
        src = S->img;
        sstride = w;
    }
    (void)sstride;
    auto grid = [](int rows, int cols) { return dim3((cols + AK_TW - 1) / AK_TW, (rows + AK_TH - 1) / AK_TH); };
    ak_taps g9, g5;
    gauss_taps(9, 1.6f, &g9);
    gauss_taps(5, 1.0, &g5);
    // A2: first level, contrast factor
    ak_level* e0 = &S->lv[0];
    hipLaunchKernelGGL((ak_gauss_kernel<uint8_t>), grid(h, w), dim3(256), 0, st, src, h, w, g9, e0->Lsmooth, e0->Lt);
    if (S->n_levels > 1) {
        HIP_TRY(ctx, hipMemsetAsync(S->hmax, 0, 256 + 2048, st));   // hmax and the histogram are adjacent
        hipLaunchKernelGGL((ak_gauss_kernel<uint8_t>), grid(h, w), dim3(256), 0, st, src, h, w, g5, S->tmpA, (float*)nullptr);
        hipLaunchKernelGGL(ak_flow_kernel, grid(h, w), dim3(256), 0, st, S->tmpA, h, w, S->kc, 0, 1, S->tmpB, S->hmax);
        const int total = (h - 2) * (w - 2);
        hipLaunchKernelGGL(ak_hist_kernel, dim3(std::min(1024, (total + 255) / 256)), dim3(256), 0, st, S->tmpB, h, w, S->hmax,
                           300, S->hist);
        hipLaunchKernelGGL(ak_kc_kernel, dim3(1), dim3(64), 0, st, S->hist, S->hmax, 300, total, 0.7f, P->n_octaves, S->kc);
    }
    static const bool small_pairs = !(getenv("IMATCH_AKAZE_PAIRS") && atoi(getenv("IMATCH_AKAZE_PAIRS")) == 0);   // A/B runs
    // The response / candidate launches take a range of levels (blockIdx.z + z0 = level)
    ak_levels_all LA;
    {
        LA.n = S->n_levels;
        for (int i = 0; i < AK_MAX_LEVELS; ++i) {
            const ak_level* e = &S->lv[i < S->n_levels ? i : 0];
            LA.rows[i] = e->rows; LA.cols[i] = e->cols; LA.s[i] = e->sigma_size; LA.border[i] = e->border;
            LA.Lsmooth[i] = e->Lsmooth; LA.Lx[i] = e->Lx; LA.Ly[i] = e->Ly; LA.Ldet[i] = e->Ldet;
        }
    }
    auto launch_group = [&](hipStream_t sq, int z0, int z1) {
        int max_rows = 1, max_cols = 1, max_iw = 0, max_ih = 0;
        for (int i = z0; i < z1; ++i) {
            const ak_level* e = &S->lv[i];
            max_rows = std::max(max_rows, e->rows); max_cols = std::max(max_cols, e->cols);
            max_iw = std::max(max_iw, e->cols - 2 * e->border); max_ih = std::max(max_ih, e->rows - 2 * e->border);
        }
        const dim3 g((max_cols + AK_TW - 1) / AK_TW, (max_rows + AK_TH - 1) / AK_TH, z1 - z0);
        hipLaunchKernelGGL(ak_deriv1_kernel, g, dim3(256), 0, sq, LA, z0);
        hipLaunchKernelGGL(ak_det_kernel, g, dim3(256), 0, sq, LA, z0);
        if (max_iw > 0 && max_ih > 0)
            hipLaunchKernelGGL(ak_candidates_kernel, dim3((max_iw + 63) / 64, (max_ih + 4 * AK_CAND_ROWS - 1) / (4 * AK_CAND_ROWS), z1 - z0), dim3(256), 0, sq, LA,
                               z0, (float)P->threshold, S->n_cand, S->cand, S->cand_cap);
    };
    auto is_small = [&](int i) {
        const ak_level* e = &S->lv[i];
        return e->rows * e->cols <= AK_SMALL_PX && 2 * e->rows * e->cols + 4 * e->cols <= AK_SMALL_LDS_FLOATS && e->rows >= 2 && e->cols >= 2;
    };
    // A frame's host time is mostly the candidate filter (0.3 ms), and the last ten levels are single-workgroup launches
    // that leave the chip empty (0.5 ms): the levels BEFORE the first of those ("early") get their response and candidates
    // on a second stream while that chain runs, and the host filters them meanwhile. IMATCH_AKAZE_OVERLAP=0: one stream.
    static const bool overlap_env = !(getenv("IMATCH_AKAZE_OVERLAP") && atoi(getenv("IMATCH_AKAZE_OVERLAP")) == 0);
    int first_small = S->n_levels;
    for (int i = 1; i < S->n_levels; ++i)
        if (is_small(i)) { first_small = i; break; }
    const bool overlap = overlap_env && first_small >= 2 && first_small < S->n_levels;
    if (overlap && !S->side) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&S->side, hipStreamNonBlocking));   // default priority (see sift.hip on why)
        HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_early, hipEventDisableTiming));
    }
    // pinned block: [early count | total count] [early records (a guess)] [all records (a guess)]
    const int guessE = overlap ? std::min(S->cand_cap, S->early_hint > 0 ? S->early_hint : 8192) : 0;
    const int guess = std::min(S->cand_cap, S->cand_hint > 0 ? S->cand_hint : 8192);
    const size_t cnt_b = 256, guessE_b = (size_t)guessE * AK_CAND_FLOATS * 4, guess_b = (size_t)guess * AK_CAND_FLOATS * 4;
    rc = imatch_hs_reserve(ctx, cnt_b + guessE_b + guess_b);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(S->n_cand, 0, 4, st));
    for (int i = 1; i < S->n_levels; ++i) {
        ak_level* e = &S->lv[i];
        ak_level* p = &S->lv[i - 1];
        const dim3 g = grid(e->rows, e->cols);
        const size_t bytes = (size_t)e->rows * e->cols * 4;
        // resize(): scale = 1 / (dsize / ssize) per axis; the 2 x 2 box path when both are exactly 2
        const double sc_x = 1. / ((double)e->cols / p->cols), sc_y = 1. / ((double)e->rows / p->rows);
        const int fast = fabs(sc_x - 2.0) < 2.220446049250313e-16 && fabs(sc_y - 2.0) < 2.220446049250313e-16;
        if (overlap && i == first_small) {
            // Fork: the levels before this one are complete. Their response, candidates and the candidates' way to the host
            // go to the second stream, beside the chain of single-workgroup levels that starts here; the host filters them
            // while that chain runs (below).
            HIP_TRY(ctx, hipEventRecord(S->ev_fork, st));
            HIP_TRY(ctx, hipStreamWaitEvent(S->side, S->ev_fork, 0));
            launch_group(S->side, 0, first_small);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipMemcpyAsync(ctx->hs, S->n_cand, 4, hipMemcpyDeviceToHost, S->side));
            HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->hs + cnt_b, S->cand, guessE_b, hipMemcpyDeviceToHost, S->side));
            HIP_TRY(ctx, hipEventRecord(S->ev_early, S->side));
        }
        if (e->rows * e->cols <= AK_SMALL_PX && 2 * e->rows * e->cols + 4 * e->cols <= AK_SMALL_LDS_FLOATS && e->rows >= 2 &&
            e->cols >= 2) {
            if ((e->cols & 1) == 0 && small_pairs)
                hipLaunchKernelGGL(ak_level_small_kernel<true>, dim3(1), dim3(1024), ((size_t)2 * e->rows * e->cols + 4 * e->cols) * 4, st, p->Lt, p->rows,
                                   p->cols, e->octave > p->octave ? 1 : 0, fast, sc_x, sc_y, e->rows, e->cols, g5, S->kc, e->octave, e->nsteps,
                                   S->tau_dev + (size_t)i * AK_MAX_STEPS, e->Lt, e->Lsmooth);
            else
                hipLaunchKernelGGL(ak_level_small_kernel<false>, dim3(1), dim3(1024), ((size_t)2 * e->rows * e->cols + 4 * e->cols) * 4, st, p->Lt, p->rows,
                                   p->cols, e->octave > p->octave ? 1 : 0, fast, sc_x, sc_y, e->rows, e->cols, g5, S->kc, e->octave, e->nsteps,
                                   S->tau_dev + (size_t)i * AK_MAX_STEPS, e->Lt, e->Lsmooth);
            continue;
        }
        // one launch per step, alternating between e->Lt and tmpA so that the last step writes e->Lt; the first step
        // reads the previous level's Lt directly (or its resized copy at a new octave). Dead end kept on record: several
        // steps per launch with a halo as wide as the step count (temporal blocking in LDS, bit-identical) was no faster
        // on a frame (1.58-1.63 against 1.55-1.60 ms) nor on a 2048^2 crop (7.0-7.3 against 6.8-6.9 ms): the ring that
        // is recomputed costs what the saved launches and HBM passes gain.
        const int n_launch = e->nsteps;
        auto out_of = [&](int k) { return ((n_launch - k) & 1) ? S->tmpA : e->Lt; };   // k = 1 .. n_launch
        const float* cur = p->Lt;
        if (e->octave > p->octave) {
            float* half = n_launch == 0 ? e->Lt : (out_of(1) == e->Lt ? S->tmpA : e->Lt);
            hipLaunchKernelGGL(ak_area_kernel, dim3((e->cols + 63) / 64, (e->rows + 3) / 4), dim3(256), 0, st, p->Lt, p->rows,
                               p->cols, half, e->rows, e->cols, fast, sc_x, sc_y);
            cur = half;
        } else if (n_launch == 0) {
            HIP_TRY(ctx, hipMemcpyAsync(e->Lt, p->Lt, bytes, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL((ak_gauss_kernel<float>), g, dim3(256), 0, st, cur, e->rows, e->cols, g5, e->Lsmooth, (float*)nullptr);
        hipLaunchKernelGGL(ak_flow_kernel, g, dim3(256), 0, st, e->Lsmooth, e->rows, e->cols, S->kc, e->octave, 0, S->tmpB,
                           (unsigned*)nullptr);
        for (int k = 1; k <= n_launch; ++k) {
            float* o = out_of(k);
            hipLaunchKernelGGL(ak_fed_kernel, g, dim3(256), 0, st, cur, S->tmpB, e->rows, e->cols, e->tau[k - 1] * 0.5f, o);
            cur = o;
        }
    }
    // A3 + candidates of the other levels (of all levels without the overlap)
    if (overlap) {
        HIP_TRY(ctx, hipStreamWaitEvent(st, S->ev_early, 0));   // the early records are the first ones of the list: nothing is appended before they are counted
        launch_group(st, first_small, S->n_levels);
    } else {
        launch_group(st, 0, S->n_levels);
    }
    HIP_TRY(ctx, hipGetLastError());
    static const bool timing = getenv("IMATCH_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    // candidates -> host
    // (the count and a guess of the records -- the previous call's count plus a margin -- in ONE synchronisation; the rest,
    // if the guess was short, in a second one)
    HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->hs + 64, S->n_cand, 4, hipMemcpyDeviceToHost, st));
    if (guess) HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->hs + cnt_b + guessE_b, S->cand, guess_b, hipMemcpyDeviceToHost, st));
    const auto t_enq = now();
    std::vector<ak_level_pts> LP((size_t)S->n_levels);
    // levels in parallel on the context's worker pool; even a frame's 6 k candidates gain (0.77 -> 0.41 ms), so the
    // inline path is only a measurement aid (IMATCH_AKAZE_PAR_MIN = candidate count below which it is taken)
    static const int par_min = getenv("IMATCH_AKAZE_PAR_MIN") ? atoi(getenv("IMATCH_AKAZE_PAR_MIN")) : 0;
    int n_cand = 0;
    auto pfor = [&](int cnt, const std::function<void(int)>& fn) {
        if (n_cand < par_min) for (int i = 0; i < cnt; ++i) fn(i);
        else imatch_parallel_for(ctx, cnt, fn);
    };
    // same level: row-major scan, the first kept neighbour decides (levels are independent of one another here);
    // record numbers grouped by level (one counting pass), so that a level's task touches its own records only
    auto same_level = [&](const float* raw, int n_rec, int l0, int l1) {
        std::vector<int> lvl_start((size_t)S->n_levels + 1, 0), lvl_ids((size_t)n_rec);
        for (int i = 0; i < n_rec; ++i) lvl_start[(int)raw[(size_t)i * AK_CAND_FLOATS] + 1]++;
        for (int l = 0; l < S->n_levels; ++l) lvl_start[l + 1] += lvl_start[l];
        {
            std::vector<int> fill(lvl_start.begin(), lvl_start.end() - 1);
            for (int i = 0; i < n_rec; ++i) lvl_ids[fill[(int)raw[(size_t)i * AK_CAND_FLOATS]]++] = i;
        }
        pfor(l1 - l0, [&](int k) {
            const int l = l0 + k;
            ak_level_pts& P = LP[l];
            bucket_level(raw, lvl_ids.data() + lvl_start[l], lvl_start[l + 1] - lvl_start[l], S->lv[l].rows, P);
            const int r = S->lv[l].sigma_size;
            for (int p = 0; p < (int)P.x.size(); ++p) {
                const int nb = P.find(P.x[p], P.y[p], r);
                if (nb >= 0) {
                    if (P.v[(size_t)p * 9 + 4] > P.v[(size_t)nb * 9 + 4]) P.alive[nb] = 0;
                    else continue;
                }
                P.alive[p] = 1;
            }
        });
    };
    bool early_done = false;
    if (overlap) {   // the early levels' candidates are here long before the chain of small levels ends
        HIP_TRY(ctx, hipEventSynchronize(S->ev_early));
        const int nE = *(const int*)ctx->hs;
        S->early_hint = nE + nE / 4 + 256;
        if (nE <= guessE && nE <= S->cand_cap) {
            n_cand = nE;   // (pfor's threshold)
            same_level((const float*)((char*)ctx->hs + cnt_b), nE, 0, first_small);
            early_done = true;
        }
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    n_cand = *(const int*)((char*)ctx->hs + 64);
    bool have_guess = true;
    if (n_cand > S->cand_cap) {
        have_guess = false;   // the scan counted them all: grow the list and scan again (the planes are still there)
        early_done = false;
        rc = akaze_grow_lists(ctx, n_cand + n_cand / 8, 0);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(S->n_cand, 0, 4, st));
        launch_group(st, 0, S->n_levels);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(&n_cand, S->n_cand, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (n_cand > S->cand_cap) return imatch_set_err(ctx, IMATCH_ESIZE, "AKAZE: candidate list overflow");
    }
    const float* raw = nullptr;   // the records stay in the pinned block until the levels are built from them
    const auto t_gpu = now();
    S->cand_hint = n_cand + n_cand / 4 + 256;
    if (n_cand) {
        const size_t rb = (size_t)n_cand * AK_CAND_FLOATS * 4;
        size_t at = cnt_b + guessE_b;
        if (!have_guess || n_cand > guess) {   // everything again at the right width (the block may move: the early levels are built already)
            rc = imatch_hs_reserve(ctx, cnt_b + rb);
            if (rc) return rc;
            at = cnt_b;
            HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->hs + at, S->cand, rb, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
        raw = (const float*)((char*)ctx->hs + at);
    }
    const auto t_dl = now();
    if (early_done) {
        same_level(raw, n_cand, first_small, S->n_levels);   // the early levels are built already
    } else {
        for (auto& P : LP) P = ak_level_pts();
        same_level(raw, n_cand, 0, S->n_levels);
    }
    // Against the level below, then against the level above. OpenCV runs the level pairs one after the other, but pair
    // (l, l-1) only READS level l's flags as the phase before left them (the pass that changes them comes later) and is
    // the only one that WRITES level l-1's: with a snapshot of the flags per phase the pairs are independent.
    std::vector<std::vector<uint8_t>> snap((size_t)S->n_levels);
    for (int l = 0; l < S->n_levels; ++l) snap[l] = LP[l].alive;
    pfor(S->n_levels - 1, [&](int i) {
        const int l = i + 1;
        const int dr = (int)S->lv[l].ratio / (int)S->lv[l - 1].ratio, r = S->lv[l].sigma_size;
        const ak_level_pts& P = LP[l];
        ak_level_pts& Q = LP[l - 1];
        for (int p = 0; p < (int)P.x.size(); ++p)
            if (snap[l][p]) {
                const int nb = Q.find(P.x[p] * dr, P.y[p] * dr, r);
                if (nb >= 0 && P.v[(size_t)p * 9 + 4] > Q.v[(size_t)nb * 9 + 4]) Q.alive[nb] = 0;
            }
    });
    for (int l = 0; l < S->n_levels; ++l) snap[l] = LP[l].alive;
    pfor(S->n_levels - 1, [&](int l) {
        const int dr = (int)S->lv[l + 1].ratio / (int)S->lv[l].ratio, r = S->lv[l].sigma_size;
        const ak_level_pts& P = LP[l];
        ak_level_pts& Q = LP[l + 1];
        for (int p = 0; p < (int)P.x.size(); ++p)
            if (snap[l][p]) {
                const int nb = Q.find(P.x[p] / dr, P.y[p] / dr, r);
                if (nb >= 0 && P.v[(size_t)p * 9 + 4] > Q.v[(size_t)nb * 9 + 4]) Q.alive[nb] = 0;
            }
    });
    // A5
    S->kp.clear();
    std::vector<float> kdev;
    for (int l = 0; l < S->n_levels; ++l) {
        const ak_level* e = &S->lv[l];
        const ak_level_pts& P = LP[l];
        for (int p = 0; p < (int)P.x.size(); ++p) {
            if (!P.alive[p]) continue;
            const float* v = &P.v[(size_t)p * 9];   // 0 1 2 / 3 4 5 / 6 7 8 = rows y-1, y, y+1
            const float Dx = 0.5f * (v[5] - v[3]), Dy = 0.5f * (v[7] - v[1]);
            const float Dxx = v[5] + v[3] - 2.0f * v[4], Dyy = v[7] + v[1] - 2.0f * v[4];
            const float Dxy = 0.25f * (v[8] + v[0] - v[2] - v[6]);
            float dx = 0.f, dy = 0.f;
            double det = (double)Dxx * Dyy - (double)Dxy * Dxy;
            if (det != 0) {
                det = 1. / det;
                dx = (float)(((double)(-Dx) * Dyy - (double)(-Dy) * Dxy) * det);
                dy = (float)(((double)(-Dy) * Dxx - (double)(-Dx) * Dxy) * det);
            }
            if (fabsf(dx) > 1.0f || fabsf(dy) > 1.0f) continue;
            const float rec[8] = {(P.x[p] + dx) * e->ratio + .5f * (e->ratio - 1.0f), (P.y[p] + dy) * e->ratio + .5f * (e->ratio - 1.0f),
                                  2.0f * (e->esigma * 1.5f), 0.f, v[4], (float)e->octave, (float)l, 0.f};
            kdev.insert(kdev.end(), rec, rec + 8);
        }
    }
    const auto t_filter = now();
    const int n = (int)(kdev.size() / 8);
    S->n_kp = n;
    S->serial = 0;   // desc_dev is about to be rewritten (or the lists regrown): no extraction's descriptors are there until the end
    S->kp.assign((size_t)n * 7, 0.f);
    S->desc.assign((size_t)n * 64, 0.f);
    if (n == 0) return IMATCH_OK;
    rc = akaze_grow_lists(ctx, 0, n + n / 4);
    if (rc) return rc;
    // A6, A7 on the device
    ak_levels_dev L;
    L.n = S->n_levels;
    for (int i = 0; i < AK_MAX_LEVELS; ++i) {
        const bool ok = i < S->n_levels;
        L.rows[i] = ok ? S->lv[i].rows : 1; L.cols[i] = ok ? S->lv[i].cols : 1; L.ratio[i] = ok ? S->lv[i].ratio : 1.f;
        L.Lx[i] = ok ? S->lv[i].Lx : nullptr; L.Ly[i] = ok ? S->lv[i].Ly : nullptr;
    }
    // records up and results down through the context's pinned block (pageable copies take the slow staged path)
    const size_t kb = (size_t)n * 8 * 4, db = (size_t)n * 64 * 4;
    rc = imatch_hs_reserve(ctx, kb + db);
    if (rc) return rc;
    memcpy(ctx->hs, kdev.data(), kb);
    HIP_TRY(ctx, hipMemcpyAsync(S->kp_dev, ctx->hs, kb, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(ak_orient_kernel, dim3((n + 3) / 4), dim3(256), 0, st, S->kp_dev, n, L);
    hipLaunchKernelGGL(ak_msurf_kernel, dim3((n + 3) / 4), dim3(256), 0, st, S->kp_dev, n, L, S->desc_dev);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs, S->kp_dev, kb, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync((char*)ctx->hs + kb, S->desc_dev, db, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(kdev.data(), ctx->hs, kb);
    memcpy(S->desc.data(), (char*)ctx->hs + kb, db);
    for (int i = 0; i < n; ++i) memcpy(S->kp.data() + (size_t)i * 7, kdev.data() + (size_t)i * 8, 28);
    static uint64_t next_serial = 0;   // process-wide: a serial names one extraction on whatever context
    S->serial = __atomic_add_fetch(&next_serial, 1, __ATOMIC_RELAXED);
    if (timing)
        fprintf(stderr, "[imatch] akaze host phases: wait for GPU %.3f ms, candidates down %.3f ms (%d), filter + sub-pixel %.3f ms, "
                "orientation + descriptors + download %.3f ms (%d)\n", ms(t_enq, t_gpu), ms(t_gpu, t_dl), n_cand, ms(t_dl, t_filter),
                ms(t_filter, now()), n);
    return IMATCH_OK;
}

extern "C" {

int imatch_akaze_extract_dev(imatch_ctx* ctx, const uint8_t* img_dev, int h, int w, int stride, const imatch_akaze_params* params,
                             int64_t* n_out) {
    if (!ctx || !img_dev || !params || !n_out || h < 3 || w < 3 || stride < w || params->n_octaves < 1 ||
        params->n_octave_layers < 1 || params->n_octaves * params->n_octave_layers > AK_MAX_LEVELS ||
        (params->clahe && (params->clahe_tiles_x <= 0 || params->clahe_tiles_y <= 0 ||
                           params->clahe_tiles_x * params->clahe_tiles_y > 4096)))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = akaze_run(ctx, img_dev, h, w, stride, params);
    if (rc) return rc;
    *n_out = ctx->akaze->n_kp;
    return IMATCH_OK;
}

int imatch_akaze_extract(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, const imatch_akaze_params* params,
                         int64_t* n_out) {
    if (!ctx || !img || h < 3 || w < 3 || stride < w) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = imatch_ws_reserve(ctx, (size_t)h * w);
    if (rc) return rc;
    rc = imatch_stage_upload(ctx, img, h, w, stride, (uint8_t*)ctx->ws);
    if (rc) return rc;
    return imatch_akaze_extract_dev(ctx, (const uint8_t*)ctx->ws, h, w, w, params, n_out);
}

int imatch_akaze_download(imatch_ctx* ctx, float* kp, float* desc) {
    if (!ctx || !ctx->akaze) return imatch_set_err(ctx, IMATCH_EINVAL, "no AKAZE extraction on this context");
    const imatch_akaze_state* S = ctx->akaze;
    if (kp && S->n_kp) memcpy(kp, S->kp.data(), (size_t)S->n_kp * 7 * 4);
    if (desc && S->n_kp) memcpy(desc, S->desc.data(), (size_t)S->n_kp * 64 * 4);
    return IMATCH_OK;
}

int imatch_akaze_last(imatch_ctx* ctx, uint64_t* serial, int64_t* n) {
    if (!ctx || !serial) return imatch_set_err(ctx, IMATCH_EINVAL, "null argument");
    *serial = (ctx->akaze && ctx->akaze->n_kp > 0) ? ctx->akaze->serial : 0;
    if (n) *n = *serial ? ctx->akaze->n_kp : 0;
    return IMATCH_OK;
}

int imatch_akaze_last_dev(imatch_ctx* ctx, uint64_t serial, const float** desc_dev, int64_t* n) {
    if (!ctx || !ctx->akaze || serial == 0 || ctx->akaze->serial != serial || ctx->akaze->n_kp <= 0) return IMATCH_EINVAL;
    *desc_dev = ctx->akaze->desc_dev;
    *n = ctx->akaze->n_kp;
    return IMATCH_OK;
}

int imatch_akaze_image(imatch_ctx* ctx, uint8_t* out, int h, int w) {
    if (!ctx || !ctx->akaze || !out || ctx->akaze->h != h || ctx->akaze->w != w)
        return imatch_set_err(ctx, IMATCH_EINVAL, "no AKAZE extraction of that size on this context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(out, ctx->akaze->img, (size_t)h * w, hipMemcpyDeviceToHost));
    return IMATCH_OK;
}

int imatch_akaze_debug_levels(imatch_ctx* ctx) { return ctx && ctx->akaze ? ctx->akaze->n_levels : 0; }

int imatch_akaze_debug_layer(imatch_ctx* ctx, int kind, int level, float* out, int* rows, int* cols) {
    if (!ctx || !ctx->akaze || level < 0 || level >= ctx->akaze->n_levels || kind < 0 || kind > 5)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    const ak_level* e = &ctx->akaze->lv[level];
    if (kind == 5) {   // the contrast factor of the level's octave, as a 1 x 1 plane
        if (rows) *rows = 1;
        if (cols) *cols = 1;
        if (out) HIP_TRY(ctx, hipMemcpy(out, ctx->akaze->kc + e->octave, 4, hipMemcpyDeviceToHost));
        return IMATCH_OK;
    }
    if (rows) *rows = e->rows;
    if (cols) *cols = e->cols;
    if (!out) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float* src = kind == 0 ? e->Lt : kind == 1 ? e->Lsmooth : kind == 2 ? e->Lx : kind == 3 ? e->Ly : e->Ldet;
    HIP_TRY(ctx, hipMemcpy(out, src, (size_t)e->rows * e->cols * 4, hipMemcpyDeviceToHost));
    return IMATCH_OK;
}

}  // extern "C"
