// Pre-processing kernels: CLAHE and INTER_AREA resize for 8-bit gray images.
// Replace cv2.createCLAHE(...).apply (reference match_finder.py:200-203) and
// cv2.resize(..., INTER_AREA) (reference utils.py:17-22). Both are byte-streaming, HBM-bound
// kernels; arithmetic follows OpenCV's order (unfused f32 products and sums), so results are
// bit-identical to the oracle.
#include "common.h"
#include "preproc.h"
#include <stdlib.h>

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ uint8_t sat_u8(float v) {
    int i = __float2int_rn(v);  // round half to even, as cvRound
    return (uint8_t)min(max(i, 0), 255);
}

// ---- CLAHE ---------------------------------------------------------------------------------
// grid (tiles_x*tiles_y, slices): each block histograms a band of rows of one tile
__global__ void __launch_bounds__(256)
clahe_hist_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, int tw, int th, int tiles_x,
                  int* __restrict__ hist, size_t src_zs, size_t zs) {
    // batched: image blockIdx.z, its source src_zs bytes and its scratch / output zs bytes further on
    src += (size_t)blockIdx.z * src_zs;
    hist = (int*)((char*)hist + (size_t)blockIdx.z * zs);
    __shared__ int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile % tiles_x;
    const int band = (th + gridDim.y - 1) / gridDim.y;
    const int y0 = blockIdx.y * band, y1 = min(th, y0 + band);
    for (int y = y0 + (threadIdx.x >> 6); y < y1; y += 4) {
        const uint8_t* p = src + (size_t)reflect101(ty * th + y, rows) * stride;
        for (int x = threadIdx.x & 63; x < tw; x += 64) atomicAdd(&h[p[reflect101(tx * tw + x, cols)]], 1);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[tile * 256 + threadIdx.x], h[threadIdx.x]);
}

// one block per tile: clip, redistribute, cumulative sum -> LUT
__global__ void __launch_bounds__(256)
clahe_lut_kernel(const int* __restrict__ hist, int clip, float lut_scale, uint8_t* __restrict__ lut, size_t zs) {
    hist = (const int*)((const char*)hist + (size_t)blockIdx.z * zs);
    lut += (size_t)blockIdx.z * zs;
    __shared__ int s[256];
    __shared__ int total;
    const int i = threadIdx.x;
    int v = hist[blockIdx.x * 256 + i];
    if (clip > 0) {
        const int over = v > clip ? v - clip : 0;
        if (v > clip) v = clip;
        if (i == 0) total = 0;
        __syncthreads();
        if (over) atomicAdd(&total, over);
        __syncthreads();
        const int clipped = total;
        const int batch = clipped / 256;
        const int residual = clipped - batch * 256;
        v += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            // for (k = 0; k < 256 && residual > 0; k += step, residual--) hist[k]++
            if (i % step == 0 && i / step < residual) v += 1;
        }
    }
    s[i] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
        const int t = i >= o ? s[i - o] : 0;
        __syncthreads();
        s[i] += t;
        __syncthreads();
    }
    lut[blockIdx.x * 256 + i] = sat_u8((float)s[i] * lut_scale);
}

// Histogram and LUT of one tile in one workgroup, the histogram in LDS (four copies, a wave uses copy wave % 4): a camera
// frame's CLAHE was a clear of the global histograms, the band kernel, the LUT kernel and the interpolation -- four
// dependent launches of 2-4 us each with their gaps at the head of a 0.55 ms frame; this replaces the first three. Integer
// counts and an integer prefix sum: the LUT is the two kernels' bit for bit. Used for tiles of at most CLAHE_FUSED_MAX_PX
// pixels (a 1024-thread workgroup walks 16 pixels per thread there); larger tiles keep the band kernel.
#define CLAHE_FUSED_MAX_PX 16384
__global__ void __launch_bounds__(1024)
clahe_tile_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, int tw, int th, int tiles_x, int clip,
                  float lut_scale, uint8_t* __restrict__ lut, size_t src_zs, size_t zs, int* __restrict__ clear64) {
    src += (size_t)blockIdx.z * src_zs;
    lut += (size_t)blockIdx.z * zs;
    if (clear64 && blockIdx.x == 0 && threadIdx.x < 64) ((int*)((char*)clear64 + (size_t)blockIdx.z * zs))[threadIdx.x] = 0;
    __shared__ int h[4][256];
    __shared__ int wsum[4];
    __shared__ int total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    ((int*)h)[tid] = 0;
    if (tid == 0) total = 0;
    __syncthreads();
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile % tiles_x;
    int* const mine = h[wave & 3];
    // a wave takes a row at a time (consecutive lanes, consecutive bytes)
    for (int y = wave; y < th; y += 16) {
        const uint8_t* p = src + (size_t)reflect101(ty * th + y, rows) * stride;
        for (int x = lane; x < tw; x += 64) atomicAdd(&mine[p[reflect101(tx * tw + x, cols)]], 1);
    }
    __syncthreads();
    int v = 0, over = 0;
    if (tid < 256) {
        v = h[0][tid] + h[1][tid] + h[2][tid] + h[3][tid];
        if (clip > 0) {
            over = v > clip ? v - clip : 0;
            if (v > clip) v = clip;
            if (over) atomicAdd(&total, over);
        }
    }
    __syncthreads();
    if (clip > 0 && tid < 256) {
        const int clipped = total;
        const int batch = clipped / 256;
        const int residual = clipped - batch * 256;
        v += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            if (tid % step == 0 && tid / step < residual) v += 1;
        }
    }
    // inclusive prefix over the 256 bins (the first four waves): within the wave by shuffles, then the waves before
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (tid < 256 && lane == 63) wsum[wave] = incl;
    __syncthreads();
    if (tid < 256) {
        for (int w = 0; w < wave; ++w) incl += wsum[w];
        lut[tile * 256 + tid] = sat_u8((float)incl * lut_scale);
    }
}

__global__ void __launch_bounds__(256)
clahe_apply_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, int tw, int th, int tiles_x,
                   int tiles_y, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst, size_t src_zs, size_t zs) {
    src += (size_t)blockIdx.z * src_zs;
    lut += (size_t)blockIdx.z * zs;
    dst += (size_t)blockIdx.z * zs;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0);
    ty2 = min(ty2, tiles_y - 1);
    const float txf = x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, tiles_x - 1);
    const int v = src[(size_t)y * stride + x];
    const float l11 = lut[(ty1 * tiles_x + tx1) * 256 + v], l12 = lut[(ty1 * tiles_x + tx2) * 256 + v];
    const float l21 = lut[(ty2 * tiles_x + tx1) * 256 + v], l22 = lut[(ty2 * tiles_x + tx2) * 256 + v];
    const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
    dst[(size_t)y * cols + x] = sat_u8(res);
}

int imatch_clahe_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, double clip_limit, int tiles_x,
                     int tiles_y, uint8_t* dst_dev, void* scratch /* >= tiles*(1024+256) bytes */, int batch, size_t src_zs,
                     size_t zs, int* clear64, bool* cleared) {
    if (cleared) *cleared = false;
    int eh = h, ew = w;
    if (w % tiles_x != 0 || h % tiles_y != 0) {
        eh = h + (tiles_y - h % tiles_y);
        ew = w + (tiles_x - w % tiles_x);
    }
    const int tw = ew / tiles_x, th = eh / tiles_y, tiles = tiles_x * tiles_y;
    const int total = tw * th;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * total / 256);
        if (clip < 1) clip = 1;
    }
    int* hist = (int*)scratch;
    uint8_t* lut = (uint8_t*)scratch + (size_t)tiles * 1024;
    // 0: the band kernel always; 1 (default): one workgroup per tile for single images with small tiles; 2: for batches too
    static const int fused_env = getenv("IMATCH_CLAHE_FUSED") ? atoi(getenv("IMATCH_CLAHE_FUSED")) : 1;
    if (total <= CLAHE_FUSED_MAX_PX && (fused_env == 2 || (fused_env == 1 && batch == 1))) {
        hipLaunchKernelGGL(clahe_tile_kernel, dim3(tiles, 1, batch), dim3(1024), 0, ctx->stream, src_dev, h, w, stride, tw, th,
                           tiles_x, clip, (float)255 / total, lut, src_zs, zs, clear64);
        if (cleared) *cleared = clear64 != nullptr;
        hipLaunchKernelGGL(clahe_apply_kernel, dim3((w + 63) / 64, (h + 3) / 4, batch), dim3(256), 0, ctx->stream, src_dev, h, w,
                           stride, tw, th, tiles_x, tiles_y, lut, dst_dev, src_zs, zs);
        HIP_TRY(ctx, hipGetLastError());
        return IMATCH_OK;
    }
    if (batch == 1) HIP_TRY(ctx, hipMemsetAsync(hist, 0, (size_t)tiles * 1024, ctx->stream));
    else HIP_TRY(ctx, hipMemset2DAsync(hist, zs, 0, (size_t)tiles * 1024, (size_t)batch, ctx->stream));
    int slices = (th + 63) / 64;
    if (slices > 64) slices = 64;
    // a camera frame has 64 tiles of 60 rows: one workgroup per tile walks them with one dependent byte load after the
    // other (12 us of a 0.7 ms frame); bands of four rows make it a thousand short workgroups instead
    if ((long long)tiles * slices * batch < 1024) slices = std::min((th + 3) / 4, (int)((1024 + (long long)tiles * batch - 1) / ((long long)tiles * batch)));
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(clahe_hist_kernel, dim3(tiles, slices, batch), dim3(256), 0, ctx->stream, src_dev, h, w, stride, tw,
                       th, tiles_x, hist, src_zs, zs);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(tiles, 1, batch), dim3(256), 0, ctx->stream, hist, clip, (float)255 / total, lut,
                       zs);
    hipLaunchKernelGGL(clahe_apply_kernel, dim3((w + 63) / 64, (h + 3) / 4, batch), dim3(256), 0, ctx->stream, src_dev, h, w,
                       stride, tw, th, tiles_x, tiles_y, lut, dst_dev, src_zs, zs);
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

// ---- INTER_AREA ----------------------------------------------------------------------------
// One thread per output pixel. The decimation table of resize.cpp:computeResizeAreaTab is
// recomputed per thread in double (3 kinds of entry: a fractional head, whole cells, a fractional
// tail); the row sums and the beta-weighted column sum follow ResizeArea_Invoker's order.
struct area_span {
    int s_head, s1, s2, s_tail;  // head source index or -1; whole cells [s1, s2); tail index or -1
    float a_head, a_mid, a_tail;
};

__device__ __forceinline__ area_span area_entry(int d, double scale, int ssize) {
    area_span e;
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, ssize - f1);
    int sx1 = (int)ceil(f1), sx2 = (int)floor(f2);
    sx2 = min(sx2, ssize - 1);
    sx1 = min(sx1, sx2);
    e.s_head = -1;
    e.s_tail = -1;
    e.a_head = e.a_tail = 0.f;
    if (sx1 - f1 > 1e-3) {
        e.s_head = sx1 - 1;
        e.a_head = (float)((sx1 - f1) / cell);
    }
    e.s1 = sx1;
    e.s2 = sx2;
    e.a_mid = (float)(1.0 / cell);
    if (f2 - sx2 > 1e-3) {
        e.s_tail = sx2;
        e.a_tail = (float)(fmin(fmin(f2 - sx2, 1.), cell) / cell);
    }
    return e;
}

__device__ __forceinline__ float area_row(const uint8_t* __restrict__ S, const area_span& ex) {
    float buf = 0.f;
    if (ex.s_head >= 0) buf += S[ex.s_head] * ex.a_head;
    for (int sx = ex.s1; sx < ex.s2; ++sx) buf += S[sx] * ex.a_mid;
    if (ex.s_tail >= 0) buf += S[ex.s_tail] * ex.a_tail;
    return buf;
}

__global__ void __launch_bounds__(256)
resize_area_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, uint8_t* __restrict__ dst, int drows,
                   int dcols, double scale_x, double scale_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dcols || dy >= drows) return;
    const area_span ex = area_entry(dx, scale_x, cols), ey = area_entry(dy, scale_y, rows);
    float sum = 0.f;
    if (ey.s_head >= 0) sum += ey.a_head * area_row(src + (size_t)ey.s_head * stride, ex);
    for (int sy = ey.s1; sy < ey.s2; ++sy) sum += ey.a_mid * area_row(src + (size_t)sy * stride, ex);
    if (ey.s_tail >= 0) sum += ey.a_tail * area_row(src + (size_t)ey.s_tail * stride, ex);
    dst[(size_t)dy * dcols + dx] = sat_u8(sum);
}

// integer factors: ResizeAreaFast (2x2: (a+b+c+d+2)>>2, otherwise sum * (1/area))
__global__ void __launch_bounds__(256)
resize_area_fast_kernel(const uint8_t* __restrict__ src, int stride, uint8_t* __restrict__ dst, int drows, int dcols,
                        int isx, int isy) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dcols || dy >= drows) return;
    int sum = 0;
    for (int j = 0; j < isy; ++j)
        for (int i = 0; i < isx; ++i) sum += src[(size_t)(dy * isy + j) * stride + dx * isx + i];
    const float scale = 1.f / (isx * isy);
    dst[(size_t)dy * dcols + dx] = (isx == 2 && isy == 2) ? (uint8_t)((sum + 2) >> 2) : sat_u8((float)sum * scale);
}

// INTER_AREA with an enlarged axis: resize.cpp's bilinear kernel with the area_mode coefficients, 8-bit fixed point
// (weights of 11 bits, int horizontal sums, the vertical pass's two-stage shifts). One thread per output pixel.
__device__ __forceinline__ void area_lin_coef(int d, double scale, double inv_scale, int ssize, bool is_x, int& s, int& a0, int& a1,
                                              bool& edge) {
    s = (int)floor(d * scale);
    float f = (float)((d + 1) - (s + 1) * inv_scale);
    f = f <= 0 ? 0.f : f - floorf(f);
    edge = false;
    if (is_x) {
        if (s < 0) { f = 0; s = 0; }
        if (s + 1 >= ssize) {
            edge = true;
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
    }
    a0 = __float2int_rn((1.f - f) * 2048.f);
    a1 = __float2int_rn(f * 2048.f);
}

__global__ void __launch_bounds__(256)
resize_area_enlarge_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, uint8_t* __restrict__ dst, int drows,
                           int dcols, double scale_x, double scale_y, double inv_x, double inv_y) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dcols || dy >= drows) return;
    int sx, a0, a1, sy, b0, b1;
    bool ex, ey;
    area_lin_coef(dx, scale_x, inv_x, cols, true, sx, a0, a1, ex);
    area_lin_coef(dy, scale_y, inv_y, rows, false, sy, b0, b1, ey);
    const int y0 = min(max(sy, 0), rows - 1), y1 = min(max(sy + 1, 0), rows - 1);
    const uint8_t* S0 = src + (size_t)y0 * stride;
    const uint8_t* S1 = src + (size_t)y1 * stride;
    int r0, r1;
    if (!ex) {
        r0 = S0[sx] * a0 + S0[sx + 1] * a1;
        r1 = S1[sx] * a0 + S1[sx + 1] * a1;
    } else {
        r0 = S0[sx] * 2048;
        r1 = S1[sx] * 2048;
    }
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    dst[(size_t)dy * dcols + dx] = (uint8_t)min(max(v, 0), 255);
}

int imatch_resize_area_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, uint8_t* dst_dev, int dh,
                           int dw) {
    const double inv_x = (double)dw / w, inv_y = (double)dh / h;
    const double sx = 1. / inv_x, sy = 1. / inv_y;
    if (dh > h || dw > w) {
        hipLaunchKernelGGL(resize_area_enlarge_kernel, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, ctx->stream, src_dev, h, w,
                           stride, dst_dev, dh, dw, sx, sy, inv_x, inv_y);
        HIP_TRY(ctx, hipGetLastError());
        return IMATCH_OK;
    }
    const int isx = (int)lrint(sx), isy = (int)lrint(sy);
    const dim3 grid((dw + 63) / 64, (dh + 3) / 4);
    if (fabs(sx - isx) < 2.220446049250313e-16 && fabs(sy - isy) < 2.220446049250313e-16)
        hipLaunchKernelGGL(resize_area_fast_kernel, grid, dim3(256), 0, ctx->stream, src_dev, stride, dst_dev, dh, dw, isx,
                           isy);
    else
        hipLaunchKernelGGL(resize_area_kernel, grid, dim3(256), 0, ctx->stream, src_dev, h, w, stride, dst_dev, dh, dw, sx,
                           sy);
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

// ---- host-pointer entry points ---------------------------------------------------------------
extern "C" {

int imatch_clahe(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, double clip_limit, int tiles_x,
                 int tiles_y, uint8_t* out) {
    if (!ctx || !img || !out || h <= 0 || w <= 0 || stride < w || tiles_x <= 0 || tiles_y <= 0 || tiles_x * tiles_y > 4096)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = ((size_t)h * w + 255) & ~(size_t)255;
    const size_t scratch = (size_t)tiles_x * tiles_y * 1280;
    int rc = imatch_ws_reserve(ctx, 2 * px + scratch);
    if (rc) return rc;
    uint8_t* src = (uint8_t*)ctx->ws;
    uint8_t* dst = src + px;
    rc = imatch_stage_upload(ctx, img, h, w, stride, src);
    if (rc) return rc;
    rc = imatch_clahe_dev(ctx, src, h, w, w, clip_limit, tiles_x, tiles_y, dst, dst + px);
    if (rc) return rc;
    rc = imatch_stage_download_begin(ctx, dst, (size_t)h * w);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    imatch_stage_download_end(ctx, out, (size_t)h * w);
    return IMATCH_OK;
}

int imatch_resize_area(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, uint8_t* out, int dh, int dw) {
    if (!ctx || !img || !out || h <= 0 || w <= 0 || stride < w || dh <= 0 || dw <= 0)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t px = ((size_t)h * w + 255) & ~(size_t)255;
    int rc = imatch_ws_reserve(ctx, px + (size_t)dh * dw);
    if (rc) return rc;
    uint8_t* src = (uint8_t*)ctx->ws;
    uint8_t* dst = src + px;
    rc = imatch_stage_upload(ctx, img, h, w, stride, src);
    if (rc) return rc;
    rc = imatch_resize_area_dev(ctx, src, h, w, w, dst, dh, dw);
    if (rc) return rc;
    rc = imatch_stage_download_begin(ctx, dst, (size_t)dh * dw);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    imatch_stage_download_end(ctx, out, (size_t)dh * dw);
    return IMATCH_OK;
}

}  // extern "C"
