// Match canvas: the debug image find_matches returns, reference match_finder.py:196
// (cv2.drawMatchesKnn(cadr.img, kp1, roi.img, kp2, good, None, flags=NOT_DRAW_SINGLE_POINTS)). The reference composes
// and draws it on the CPU for every frame (a 2048^2 crop beside the frame is a 13 MB canvas); here two kernels do it:
//   draw_owner_kernel  one wave per match rasterises its line (integer DDA) and both circles into an int32 "owner"
//                      plane with atomicMax on the match index -- the highest index wins a pixel, which is what drawing
//                      the matches in order gives, and makes the parallel result deterministic;
//   canvas_kernel      one pass over the canvas: owner >= 0 -> that match's colour, else the gray of img1 / img2 / 0.
// Pixels follow "Spec D" (oracle/draw_oracle.c), not OpenCV's anti-aliased primitives and RNG colours; the canvas has the
// layout and dtype cv2 returns. HBM-bound: 4 B owner write + read, 1 B gray read, 3 B write per canvas pixel.
#include "common.h"

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int px(float v) { return __float2int_rn(fminf(fmaxf(v, -1048576.f), 1048576.f)); }

__device__ __forceinline__ int dda(int k, int d, int m) {   // k d / m rounded half up, m > 0
    const long long num = 2LL * k * d + m, den = 2LL * m;
    long long q = num / den;
    if (num % den != 0 && num < 0) --q;
    return (int)q;
}

__device__ __forceinline__ void put(int* owner, int H, int W, int x, int y, int i) {
    if (x >= 0 && x < W && y >= 0 && y < H) atomicMax(owner + (size_t)y * W + x, i);
}

__global__ void __launch_bounds__(256)
draw_owner_kernel(const float2* __restrict__ xy1, const float2* __restrict__ xy2, int n, int w1, int H, int W,
                  int* __restrict__ owner) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float2 a = xy1[i], b = xy2[i];
    const int x0 = px(a.x), y0 = px(a.y), x1 = px(b.x) + w1, y1 = px(b.y);
    const int dx = x1 - x0, dy = y1 - y0;
    const int m = max(abs(dx), abs(dy));
    for (int k = lane; k <= m; k += 64) put(owner, H, W, x0 + (m ? dda(k, dx, m) : 0), y0 + (m ? dda(k, dy, m) : 0), i);
    const int r = 3;
    for (int t = lane; t < 2 * 81; t += 64) {   // two 9 x 9 boxes
        const int e = t / 81, c = t % 81, u = c % 9 - 4, v = c / 9 - 4;
        const int d4 = 4 * (u * u + v * v);
        if (d4 >= (2 * r - 1) * (2 * r - 1) && d4 < (2 * r + 1) * (2 * r + 1))
            put(owner, H, W, (e ? x1 : x0) + u, (e ? y1 : y0) + v, i);
    }
}

__global__ void __launch_bounds__(256)
canvas_kernel(const uint8_t* __restrict__ img1, int h1, int w1, const uint8_t* __restrict__ img2, int h2, int w2,
              const int* __restrict__ owner, int H, int W, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const int i = owner[p];
    uint8_t c0, c1, c2;
    if (i >= 0) {
        c0 = (uint8_t)(64 + mix32((uint32_t)i * 3u) % 192u);
        c1 = (uint8_t)(64 + mix32((uint32_t)i * 3u + 1u) % 192u);
        c2 = (uint8_t)(64 + mix32((uint32_t)i * 3u + 2u) % 192u);
    } else {
        uint8_t g = 0;
        if (x < w1) { if (y < h1) g = img1[(size_t)y * w1 + x]; }
        else if (y < h2) g = img2[(size_t)y * w2 + (x - w1)];
        c0 = c1 = c2 = g;
    }
    out[3 * p] = c0; out[3 * p + 1] = c1; out[3 * p + 2] = c2;
}

// ---- overlay primitives: the pose overlay of find_keypoints_transform (reference match_finder.py:249-250:
// cv2.circle(img, c, 10, 255, 5) + cv2.polylines(img, [quad], True, 255, 3)) and the map marker of
// utils.draw_circle_on_map_by_coord_and_angles (reference utils.py:75-76: cv2.line(.., color, 2) + cv2.circle(.., 4, color, 2)).
// Same scheme as the match canvas: one wave per primitive rasterises into the owner plane (atomicMax on the primitive's
// index: drawing in order), one pass composes. Pixels follow "Spec D" (oracle/draw_oracle.c): a thick line is the DDA line
// stamped with a disc brush 4 (u^2 + v^2) <= t^2; a ring is (2r - t)^2 <= 4 d^2 < (2r + t)^2; thickness < 0 fills the disc.
__global__ void __launch_bounds__(256)
overlay_owner_kernel(const imatch_draw_prim* __restrict__ prims, int n, int H, int W, int* __restrict__ owner) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const imatch_draw_prim p = prims[i];
    const int t = p.thickness;
    if (p.kind == IMATCH_DRAW_LINE) {
        const int dx = p.x1 - p.x0, dy = p.y1 - p.y0;
        const int m = max(abs(dx), abs(dy));
        const int hb = t / 2, S = 2 * hb + 1, SS = S * S;
        // Only the DDA steps whose brush can touch the image are walked: step k sits at x0 + round(k dx / m), so
        // k dx / m within [-hb - x0 - 1, W - 1 + hb - x0 + 1] is a superset of the steps that matter on the x axis (the
        // rounding moves a step by at most half a pixel), likewise on y; the axis with |d| = m makes the intersection at
        // most max(W, H) + 2 hb + 3 steps long, whatever the end points (a wild homography puts them millions of pixels
        // away: walking all m + 1 steps on one wave took minutes). Steps outside the range only produce pixels that
        // put() would drop, so the drawn pixels are the same.
        long long k_lo = 0, k_hi = m;
        auto clip_axis = [&](int p0, int d, int extent) {
            const long long lo = -(long long)hb - p0 - 1, hi = (long long)extent - 1 + hb - p0 + 1;   // bounds on k d / m
            if (d == 0 || m == 0) {
                if (0 < lo || 0 > hi) k_hi = -1;
                return;
            }
            // k d in [lo m, hi m]
            const long long a = lo * m, b = hi * m;
            auto fdiv = [](long long n, long long q) { long long r = n / q; if ((n % q != 0) && ((n < 0) != (q < 0))) --r; return r; };
            auto cdiv = [&](long long n, long long q) { return -fdiv(-n, q); };
            const long long ka = d > 0 ? cdiv(a, d) : cdiv(b, d), kb = d > 0 ? fdiv(b, d) : fdiv(a, d);
            k_lo = max(k_lo, ka);
            k_hi = min(k_hi, kb);
        };
        clip_axis(p.x0, dx, W);
        clip_axis(p.y0, dy, H);
        const long long total = k_hi >= k_lo ? (k_hi - k_lo + 1) * SS : 0;
        for (long long e = lane; e < total; e += 64) {
            const int k = (int)(k_lo + e / SS), c = (int)(e % SS), u = c % S - hb, v = c / S - hb;
            if (4 * (u * u + v * v) <= t * t)
                put(owner, H, W, p.x0 + (m ? dda(k, dx, m) : 0) + u, p.y0 + (m ? dda(k, dy, m) : 0) + v, i);
        }
    } else {
        const int r = p.x1;
        const int R = r + (t > 0 ? (t + 1) / 2 : 0) + 1, S = 2 * R + 1;
        const long long lo = (t > 0 && 2 * r > t) ? (long long)(2 * r - t) * (2 * r - t) : 0;
        const long long hi = t > 0 ? (long long)(2 * r + t) * (2 * r + t) : 4LL * r * r + 1;
        for (int e = lane; e < S * S; e += 64) {
            const int u = e % S - R, v = e / S - R;
            const long long d4 = 4LL * (u * u + v * v);
            if (d4 >= lo && d4 < hi) put(owner, H, W, p.x0 + u, p.y0 + v, i);
        }
    }
}

template <int CH>
__global__ void __launch_bounds__(256)
overlay_compose_kernel(const uint8_t* __restrict__ img, const imatch_draw_prim* __restrict__ prims, const int* __restrict__ owner,
                       int H, int W, uint8_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const int i = owner[p];
#pragma unroll
    for (int c = 0; c < CH; ++c) out[CH * p + c] = i >= 0 ? prims[i].color[c] : img[CH * p + c];
}

extern "C" int imatch_draw_overlay(imatch_ctx* ctx, const uint8_t* img, int h, int w, int channels, int stride,
                                   const imatch_draw_prim* prims, int n, uint8_t* out) {
    if (!ctx || !img || !out || h <= 0 || w <= 0 || (channels != 1 && channels != 3) || stride < w * channels || n < 0 ||
        (n > 0 && !prims))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    for (int i = 0; i < n; ++i) {
        const imatch_draw_prim& p = prims[i];
        const int lim = 1 << 28;   // (the kernel's 64-bit products of a step number and a difference stay below 2^60)
        bool ok = (p.kind == IMATCH_DRAW_LINE || p.kind == IMATCH_DRAW_CIRCLE) && abs(p.x0) <= lim && abs(p.y0) <= lim &&
                  p.thickness <= 255;
        if (ok && p.kind == IMATCH_DRAW_LINE) ok = abs(p.x1) <= lim && abs(p.y1) <= lim && p.thickness >= 1;
        if (ok && p.kind == IMATCH_DRAW_CIRCLE) ok = p.x1 >= 0 && p.x1 <= 4096 && p.thickness != 0;
        if (!ok) return imatch_set_err(ctx, IMATCH_EINVAL, "primitive %d: kind, coordinates (|v| <= 2^28), radius (<= 4096) or "
                                       "thickness (1..255; circles also < 0 = filled) out of range", i);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t row = (size_t)w * channels;
    const size_t bi = up((size_t)h * row), bo = up((size_t)h * w * 4), bp = up((size_t)(n > 0 ? n : 1) * sizeof(imatch_draw_prim));
    int rc = imatch_ws_reserve(ctx, 2 * bi + bo + bp);
    if (rc) return rc;
    uint8_t* d_img = (uint8_t*)ctx->ws;
    int* owner = (int*)(d_img + bi);
    uint8_t* d_out = (uint8_t*)owner + bo;
    imatch_draw_prim* d_pr = (imatch_draw_prim*)(d_out + bi);
    rc = imatch_hs_reserve(ctx, bi + bp);
    if (rc) return rc;
    uint8_t* hs = (uint8_t*)ctx->hs;
    for (int y = 0; y < h; ++y) memcpy(hs + (size_t)y * row, img + (size_t)y * stride, row);
    if (n > 0) memcpy(hs + bi, prims, (size_t)n * sizeof(imatch_draw_prim));
    HIP_TRY(ctx, hipMemcpyAsync(d_img, hs, bi, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(d_pr, hs + bi, bp, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(owner, 0xff, (size_t)h * w * 4, ctx->stream));
    if (n > 0) hipLaunchKernelGGL(overlay_owner_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, d_pr, n, h, w, owner);
    if (channels == 1)
        hipLaunchKernelGGL(overlay_compose_kernel<1>, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, d_img, d_pr, owner, h, w, d_out);
    else
        hipLaunchKernelGGL(overlay_compose_kernel<3>, dim3((w + 255) / 256, h), dim3(256), 0, ctx->stream, d_img, d_pr, owner, h, w, d_out);
    HIP_TRY(ctx, hipGetLastError());
    rc = imatch_stage_download_begin(ctx, d_out, (size_t)h * row);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    imatch_stage_download_end(ctx, out, (size_t)h * row);
    return IMATCH_OK;
}

extern "C" int imatch_draw_matches(imatch_ctx* ctx, const uint8_t* img1, int h1, int w1, int stride1, const uint8_t* img2,
                                   int h2, int w2, int stride2, const float* xy1, const float* xy2, int n, uint8_t* out) {
    if (!ctx || !img1 || !img2 || !out || h1 <= 0 || w1 <= 0 || h2 <= 0 || w2 <= 0 || stride1 < w1 || stride2 < w2 || n < 0 ||
        (n > 0 && (!xy1 || !xy2)))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int H = h1 > h2 ? h1 : h2, W = w1 + w2;
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b1 = up((size_t)h1 * w1), b2 = up((size_t)h2 * w2), bo = up((size_t)H * W * 4), bc = up((size_t)H * W * 3);
    const size_t bp = up((size_t)(n > 0 ? n : 1) * 8);
    int rc = imatch_ws_reserve(ctx, b1 + b2 + bo + bc + 2 * bp);
    if (rc) return rc;
    uint8_t* d1 = (uint8_t*)ctx->ws;
    uint8_t* d2 = d1 + b1;
    int* owner = (int*)(d2 + b2);
    uint8_t* canvas = (uint8_t*)owner + bo;
    float2* p1 = (float2*)(canvas + bc);
    float2* p2 = (float2*)((uint8_t*)p1 + bp);
    // both images and both point lists go up through one pinned block
    rc = imatch_hs_reserve(ctx, b1 + b2 + 2 * bp);
    if (rc) return rc;
    uint8_t* hs = (uint8_t*)ctx->hs;
    for (int y = 0; y < h1; ++y) memcpy(hs + (size_t)y * w1, img1 + (size_t)y * stride1, (size_t)w1);
    for (int y = 0; y < h2; ++y) memcpy(hs + b1 + (size_t)y * w2, img2 + (size_t)y * stride2, (size_t)w2);
    if (n > 0) {
        memcpy(hs + b1 + b2, xy1, (size_t)n * 8);
        memcpy(hs + b1 + b2 + bp, xy2, (size_t)n * 8);
    }
    HIP_TRY(ctx, hipMemcpyAsync(d1, hs, b1 + b2, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(p1, hs + b1 + b2, 2 * bp, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(owner, 0xff, (size_t)H * W * 4, ctx->stream));
    if (n > 0)
        hipLaunchKernelGGL(draw_owner_kernel, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, p1, p2, n, w1, H, W, owner);
    hipLaunchKernelGGL(canvas_kernel, dim3((W + 255) / 256, H), dim3(256), 0, ctx->stream, d1, h1, w1, d2, h2, w2, owner, H, W,
                       canvas);
    HIP_TRY(ctx, hipGetLastError());
    rc = imatch_stage_download_begin(ctx, canvas, (size_t)H * W * 3);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    imatch_stage_download_end(ctx, out, (size_t)H * W * 3);
    return IMATCH_OK;
}
