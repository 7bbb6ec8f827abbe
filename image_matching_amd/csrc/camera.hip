// Camera pre-step: what the reference's capture node does to every frame before it is published,
// csi_camera.py:138-167,184 (fisheye undistort + centre crop; same method in photo_publisher.py:143-172), and the gray
// conversion every consumer applies first (position_finder.py:171, mezhcadr.py:85, image_processing.py:54).
//
// The reference rebuilds the 1920x1080 rectification maps on every frame (initUndistortRectifyMap inside undistort()).
// They depend only on the calibration and the image size, so here they are built once per (K, D, newK, size, crop) on
// the host -- in f64 with OpenCV's running sums along a row, the order a per-pixel GPU evaluation could not
// reproduce to the last bit -- kept in HBM in OpenCV's CV_16SC2 fixed-point form, and only for the rows and columns
// the crop keeps. Per frame the GPU runs ONE kernel: fixed-point bilinear remap of the kept region (integer
// arithmetic, bit-identical to the oracle), optionally straight to gray. HBM-bound byte streaming:
// 6 B of map + 12 source bytes (L2-resident neighbours) + 3 B out per kept pixel.
#include <math.h>
#include <stdlib.h>
#include "common.h"

struct imatch_camera_state {
    double key[22];            // K, D, newK
    int w, h, x0, y0, cw, ch;  // source size and kept window of the rectified image
    short2* map1;              // device [ch][cw]: integer source position
    uint16_t* map2;            // device [ch][cw]: (fy << 5) | fx, 5-bit fractions
};

void imatch_camera_state_free(imatch_ctx* ctx) {
    if (!ctx->camera) return;
    if (ctx->camera->map1) (void)hipFree(ctx->camera->map1);
    if (ctx->camera->map2) (void)hipFree(ctx->camera->map2);
    free(ctx->camera);
    ctx->camera = nullptr;
}

// ---- gray ------------------------------------------------------------------------------------------------------
// cv2.cvtColor(COLOR_BGR2GRAY / COLOR_RGB2GRAY), 8-bit: (3735 B + 19235 G + 9798 R + 2^14) >> 15.
// A thread converts 4 pixels: three aligned 4-byte loads, one 4-byte store.
__global__ void __launch_bounds__(256)
cvt_gray_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, int c0w, int c2w, uint8_t* __restrict__ dst) {
    const int y = blockIdx.y;
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (y >= rows || x4 >= cols) return;
    const uint8_t* row = src + (size_t)y * stride;
    uint8_t g[4];
    const bool vec = x4 + 4 <= cols && ((((size_t)row) + 3 * x4) & 3) == 0;
    if (vec) {
        const uint32_t* p = (const uint32_t*)(row + 3 * x4);
        const uint32_t a = p[0], b = p[1], c = p[2];
        const uint8_t px[12] = {(uint8_t)a, (uint8_t)(a >> 8), (uint8_t)(a >> 16), (uint8_t)(a >> 24),
                                (uint8_t)b, (uint8_t)(b >> 8), (uint8_t)(b >> 16), (uint8_t)(b >> 24),
                                (uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24)};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            g[i] = (uint8_t)((px[3 * i] * c0w + px[3 * i + 1] * 19235 + px[3 * i + 2] * c2w + (1 << 14)) >> 15);
    } else {
        for (int i = 0; i < 4 && x4 + i < cols; ++i) {
            const uint8_t* q = row + 3 * (x4 + i);
            g[i] = (uint8_t)((q[0] * c0w + q[1] * 19235 + q[2] * c2w + (1 << 14)) >> 15);
        }
    }
    uint8_t* o = dst + (size_t)y * cols + x4;
    if (x4 + 4 <= cols && (((size_t)o) & 3) == 0) {
        *(uint32_t*)o = g[0] | (g[1] << 8) | (g[2] << 16) | ((uint32_t)g[3] << 24);
    } else {
        for (int i = 0; i < 4 && x4 + i < cols; ++i) o[i] = g[i];
    }
}

int imatch_cvt_gray_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, int swap_rb, uint8_t* dst_dev) {
    const dim3 grid((w + 1023) / 1024, h);
    hipLaunchKernelGGL(cvt_gray_kernel, grid, dim3(256), 0, ctx->stream, src_dev, h, w, stride, swap_rb ? 9798 : 3735,
                       swap_rb ? 3735 : 9798, dst_dev);
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

// ---- fisheye model on the host -------------------------------------------------------------------------------------
static void undistort_point(const double* K, const double* D, double px, double py, double* ox, double* oy) {
    const double wx = (px - K[2]) / K[0], wy = (py - K[5]) / K[4];
    double theta_d = sqrt(wx * wx + wy * wy);
    theta_d = fmin(fmax(theta_d, -M_PI / 2), M_PI / 2);
    double scale = 1.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int it = 0; it < 10; ++it) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double a = D[0] * t2, b = D[1] * t4, c = D[2] * t6, d = D[3] * t8;
            const double fix = (theta * (1 + a + b + c + d) - theta_d) / (1 + 3 * a + 5 * b + 7 * c + 9 * d);
            theta = theta - fix;
            if (fabs(fix) < 1e-8) break;
        }
        scale = tan(theta) / theta_d;
    }
    *ox = wx * scale;
    *oy = wy * scale;
}

// one row of the CV_16SC2 maps, columns [0, x1): the homogeneous point is advanced by running sums from column 0
static void map_row(const double* K, const double* D, const double* iR, int i, int x0, int x1, short2* m1, uint16_t* m2) {
    double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
    for (int j = 0; j < x1; ++j) {
        if (j >= x0) {
            double u, v;
            if (_w <= 0) {
                u = v = -INFINITY;
            } else {
                const double x = _x / _w, y = _y / _w;
                const double r = sqrt(x * x + y * y);
                const double theta = atan(r);
                const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
                const double theta_d = theta * (1 + D[0] * t2 + D[1] * t4 + D[2] * t6 + D[3] * t8);
                const double scale = (r == 0) ? 1.0 : theta_d / r;
                u = K[0] * x * scale + K[2];
                v = K[4] * y * scale + K[5];
            }
            const int iu = (int)lrint(u * 32), iv = (int)lrint(v * 32);
            m1[j - x0] = make_short2((short)(iu >> 5), (short)(iv >> 5));
            m2[j - x0] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
        _x += iR[0];
        _y += iR[3];
        _w += iR[6];
    }
}

static int camera_maps(imatch_ctx* ctx, const double* K, const double* D, const double* newK, int h, int w, int x0, int y0,
                       int cw, int ch) {
    double key[22];
    memcpy(key, K, 72);
    memcpy(key + 9, D, 32);
    memcpy(key + 13, newK, 72);
    imatch_camera_state* st = ctx->camera;
    if (st && !memcmp(st->key, key, sizeof key) && st->w == w && st->h == h && st->x0 == x0 && st->y0 == y0 &&
        st->cw == cw && st->ch == ch)
        return IMATCH_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // a previous frame may still read the old maps
    imatch_camera_state_free(ctx);
    st = (imatch_camera_state*)calloc(1, sizeof *st);
    if (!st) return imatch_set_err(ctx, IMATCH_ENOMEM, "host allocation failed");
    ctx->camera = st;
    const size_t n = (size_t)cw * ch;
    short2* h1 = (short2*)malloc(n * sizeof(short2));
    uint16_t* h2 = (uint16_t*)malloc(n * sizeof(uint16_t));
    if (!h1 || !h2) {
        free(h1); free(h2);
        return imatch_set_err(ctx, IMATCH_ENOMEM, "host allocation failed");
    }
    const double a = newK[0], s = newK[1], c = newK[2], b = newK[4], d = newK[5];
    const double iR[9] = {1 / a, -s / (a * b), (s * d - c * b) / (a * b), 0, 1 / b, -d / b, 0, 0, 1};
    imatch_parallel_for(ctx, ch, [&](int r) { map_row(K, D, iR, y0 + r, x0, x0 + cw, h1 + (size_t)r * cw, h2 + (size_t)r * cw); });
    hipError_t e = hipMalloc((void**)&st->map1, n * sizeof(short2));
    if (e == hipSuccess) e = hipMalloc((void**)&st->map2, n * sizeof(uint16_t));
    if (e == hipSuccess) e = hipMemcpy(st->map1, h1, n * sizeof(short2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(st->map2, h2, n * sizeof(uint16_t), hipMemcpyHostToDevice);
    free(h1);
    free(h2);
    if (e != hipSuccess) {
        imatch_camera_state_free(ctx);
        return imatch_set_err(ctx, IMATCH_EHIP, "rectification maps: %s", hipGetErrorString(e));
    }
    memcpy(st->key, key, sizeof key);
    st->w = w; st->h = h; st->x0 = x0; st->y0 = y0; st->cw = cw; st->ch = ch;
    return IMATCH_OK;
}

// ---- remap ---------------------------------------------------------------------------------------------------------
// cv2.remap with CV_16SC2 maps, INTER_LINEAR (what INTER_AREA becomes inside remap), BORDER_CONSTANT 0.
// Thread per kept pixel; CH = 1 or 3 interleaved channels; GRAY: 3-channel BGR source, gray output (the conversion
// is applied to the remapped BGR value, i.e. remap then cvtColor as the reference's nodes do it).
template <int CH, bool GRAY>
__global__ void __launch_bounds__(256)
remap_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, const short2* __restrict__ map1,
             const uint16_t* __restrict__ map2, int drows, int dcols, uint8_t* __restrict__ dst) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dcols || y >= drows) return;
    const size_t p = (size_t)y * dcols + x;
    const short2 s = map1[p];
    const int m = map2[p];
    const int fx = m & 31, fy = (m >> 5) & 31;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if (w00 == 32768) { w00 = 32767; w11 = 1; }   // saturate_cast<short> of the unit weight, and OpenCV's sum correction
    const bool x0in = s.x >= 0 && s.x < cols, x1in = s.x + 1 >= 0 && s.x + 1 < cols;
    const bool y0in = s.y >= 0 && s.y < rows, y1in = s.y + 1 >= 0 && s.y + 1 < rows;
    const uint8_t* r0 = src + (ptrdiff_t)s.y * stride + (ptrdiff_t)s.x * CH;
    const uint8_t* r1 = r0 + stride;
    int out[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        const int v00 = (x0in && y0in) ? r0[k] : 0, v01 = (x1in && y0in) ? r0[CH + k] : 0;
        const int v10 = (x0in && y1in) ? r1[k] : 0, v11 = (x1in && y1in) ? r1[CH + k] : 0;
        out[k] = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    }
    if (GRAY) {
        dst[p] = (uint8_t)((out[0] * 3735 + out[CH > 1 ? 1 : 0] * 19235 + out[CH > 2 ? 2 : 0] * 9798 + (1 << 14)) >> 15);
    } else {
#pragma unroll
        for (int k = 0; k < CH; ++k) dst[p * CH + k] = (uint8_t)out[k];
    }
}

int imatch_fisheye_undistort_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int channels, int stride,
                                 const double* K, const double* D, const double* newK, int x0, int y0, int cw, int ch,
                                 int to_gray, uint8_t* dst_dev) {
    int rc = camera_maps(ctx, K, D, newK, h, w, x0, y0, cw, ch);
    if (rc) return rc;
    const imatch_camera_state* st = ctx->camera;
    const dim3 grid((cw + 63) / 64, (ch + 3) / 4);
#define REMAP(CH_, G_) hipLaunchKernelGGL((remap_kernel<CH_, G_>), grid, dim3(256), 0, ctx->stream, src_dev, h, w, stride, \
                                          st->map1, st->map2, ch, cw, dst_dev)
    if (channels == 3 && to_gray) REMAP(3, true);
    else if (channels == 3) REMAP(3, false);
    else REMAP(1, false);
#undef REMAP
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

extern "C" {

int imatch_cvt_gray(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, int swap_rb, uint8_t* out) {
    if (!ctx || !img || !out || h <= 0 || w <= 0 || stride < 3 * w)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sb = ((size_t)h * w * 3 + 255) & ~(size_t)255;
    int rc = imatch_ws_reserve(ctx, sb + (size_t)h * w);
    if (rc) return rc;
    uint8_t* src = (uint8_t*)ctx->ws;
    uint8_t* dst = src + sb;
    rc = imatch_stage_upload(ctx, img, h, 3 * w, stride, src);
    if (rc) return rc;
    rc = imatch_cvt_gray_dev(ctx, src, h, w, 3 * w, swap_rb, dst);
    if (rc) return rc;
    rc = imatch_stage_download_begin(ctx, dst, (size_t)h * w);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    imatch_stage_download_end(ctx, out, (size_t)h * w);
    return IMATCH_OK;
}

int imatch_fisheye_new_camera_matrix(const double* K, const double* D, int w, int h, double balance, double* newK) {
    if (!K || !D || !newK || w <= 0 || h <= 0) return IMATCH_EINVAL;
    balance = fmin(fmax(balance, 0.0), 1.0);
    double p[4][2] = {{(double)(w / 2), 0}, {(double)w, (double)(h / 2)}, {(double)(w / 2), (double)h}, {0, (double)(h / 2)}};
    double cnx = 0, cny = 0;
    for (int i = 0; i < 4; ++i) {
        undistort_point(K, D, p[i][0], p[i][1], &p[i][0], &p[i][1]);
        cnx += p[i][0];
        cny += p[i][1];
    }
    cnx /= 4;
    cny /= 4;
    const double aspect = K[0] / K[4];
    cny *= aspect;
    double minx = 1e300, miny = 1e300, maxx = -1e300, maxy = -1e300;
    for (int i = 0; i < 4; ++i) {
        p[i][1] *= aspect;
        miny = fmin(miny, p[i][1]); maxy = fmax(maxy, p[i][1]);
        minx = fmin(minx, p[i][0]); maxx = fmax(maxx, p[i][0]);
    }
    const double f1 = w * 0.5 / (cnx - minx), f2 = w * 0.5 / (maxx - cnx);
    const double f3 = h * 0.5 * aspect / (cny - miny), f4 = h * 0.5 * aspect / (maxy - cny);
    const double fmn = fmin(f1, fmin(f2, fmin(f3, f4))), fmx = fmax(f1, fmax(f2, fmax(f3, f4)));
    const double f = balance * fmn + (1.0 - balance) * fmx;
    newK[0] = f; newK[1] = 0; newK[2] = -cnx * f + w * 0.5;
    newK[3] = 0; newK[4] = f / aspect; newK[5] = (-cny * f + h * aspect * 0.5) / aspect;
    newK[6] = 0; newK[7] = 0; newK[8] = 1;
    return IMATCH_OK;
}

int imatch_fisheye_undistort(imatch_ctx* ctx, const uint8_t* img, int h, int w, int channels, int stride, const double* K,
                             const double* D, const double* newK, int x0, int y0, int cw, int ch, int to_gray,
                             uint8_t* out) {
    if (!ctx || !img || !out || !K || !D || !newK || h <= 0 || w <= 0 || (channels != 1 && channels != 3) ||
        stride < w * channels || x0 < 0 || y0 < 0 || cw <= 0 || ch <= 0 || x0 + cw > w || y0 + ch > h ||
        (to_gray && channels != 3))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    if (w > 32767 || h > 32767) return imatch_set_err(ctx, IMATCH_ESIZE, "CV_16SC2 maps address at most 32767 pixels a side");
    if (!(newK[0] != 0) || !(newK[4] != 0)) return imatch_set_err(ctx, IMATCH_EINVAL, "singular new camera matrix");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t sb = ((size_t)h * w * channels + 255) & ~(size_t)255;
    const size_t ob = (size_t)cw * ch * (to_gray ? 1 : channels);
    int rc = imatch_ws_reserve(ctx, sb + ob);
    if (rc) return rc;
    uint8_t* src = (uint8_t*)ctx->ws;
    uint8_t* dst = src + sb;
    rc = imatch_stage_upload(ctx, img, h, w * channels, stride, src);
    if (rc) return rc;
    rc = imatch_fisheye_undistort_dev(ctx, src, h, w, channels, w * channels, K, D, newK, x0, y0, cw, ch, to_gray, dst);
    if (rc) return rc;
    rc = imatch_stage_download_begin(ctx, dst, ob);
    if (rc) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    imatch_stage_download_end(ctx, out, ob);
    return IMATCH_OK;
}

}  // extern "C"
