/*
 * camera_oracle.c -- CPU restatement of the camera pre-step: TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Follows the reference's capture node, image_processing/src/image_processing/csi_camera.py:138-167
 * (`undistort`: cv2.fisheye.estimateNewCameraMatrixForUndistortRectify -> cv2.fisheye.initUndistortRectifyMap(CV_16SC2)
 * -> cv2.remap(INTER_AREA, BORDER_CONSTANT) -> centre crop; the same method is in photo_publisher.py:143-172) and the
 * gray conversion every consumer applies to the published frame (position_finder.py:171, mezhcadr.py:85,
 * photo_publisher.py:112: cv2.cvtColor(COLOR_BGR2GRAY); image_processing.py:54: COLOR_RGB2GRAY for the map).
 *
 * PARITY UNPINNED: the arithmetic is OpenCV's (calib3d/src/fisheye.cpp, imgproc/src/imgwarp.cpp, color_rgb.simd.hpp);
 * OpenCV is absent here and the reference holds no fixtures. Restated from the published 4.x sources:
 *  - gray: fixed point, shift 15, coefficients R 9798, G 19235, B 3735 (sum 32768), round to nearest by adding 1 << 14.
 *    (OpenCV 3.x used the 14-bit set 4899/9617/1868; the reference needs >= 4.5 for USAC_MAGSAC.)
 *  - new camera matrix: the four edge mid-points (w/2, 0), (w, h/2), (w/2, h), (0, h/2) (integer halves) are
 *    undistorted (Newton on theta, 10 steps, stop below 1e-8), their mean becomes the centre, focal = balance * fmin +
 *    (1 - balance) * fmax over the four half-extents at identity aspect ratio.
 *  - maps: iR = inv(newK) (OpenCV inverts through an SVD; the closed form of the upper-triangular inverse is used
 *    here, equal to rounding); per row the homogeneous point starts at i * iR[:,1] + iR[:,2] and is advanced by
 *    iR[:,0] per column (running sums, as OpenCV's loop), theta = atan(r), theta_d = theta (1 + k1 t^2 + k2 t^4 + k3
 *    t^6 + k4 t^8), u = fx x theta_d / r + cx; CV_16SC2: iu = cvRound(u * 32), map1 = (iu >> 5, iv >> 5) as shorts,
 *    map2 = (iv & 31) * 32 + (iu & 31).
 *  - remap: INTER_AREA is replaced by INTER_LINEAR inside cv2.remap; fixed-point bilinear with the 32 x 32 weight
 *    table (weights a * b * 32 for 5-bit fractions, sum 32768; the (0, 0) entry saturates to 32767 and OpenCV's
 *    correction puts the missing 1 on the last weight), value = (sum + (1 << 14)) >> 15 per channel; taps outside the
 *    source read the constant border value 0.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* cv2.cvtColor(img, COLOR_BGR2GRAY) (swap_rb = 0) or COLOR_RGB2GRAY (swap_rb = 1), 8-bit, 3 interleaved channels */
void oracle_cvt_gray(const uint8_t* src, int rows, int cols, int swap_rb, uint8_t* dst) {
    for (int64_t i = 0; i < (int64_t)rows * cols; ++i) {
        const int c0 = src[3 * i], g = src[3 * i + 1], c2 = src[3 * i + 2];
        const int b = swap_rb ? c2 : c0, r = swap_rb ? c0 : c2;
        dst[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    }
}

/* cv2.fisheye.undistortPoints for one pixel, R = I, no P: normalised undistorted coordinates */
static void fisheye_undistort_point(const double* K, const double* D, double px, double py, double* ox, double* oy) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const double wx = (px - cx) / fx, wy = (py - cy) / fy;
    double theta_d = sqrt(wx * wx + wy * wy);
    if (theta_d > M_PI / 2) theta_d = M_PI / 2;
    if (theta_d < -M_PI / 2) theta_d = -M_PI / 2;
    double scale = 1.0;
    if (theta_d > 1e-8) {
        double theta = theta_d;
        for (int j = 0; j < 10; ++j) {
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t6 * t2;
            const double k0 = D[0] * t2, k1 = D[1] * t4, k2 = D[2] * t6, k3 = D[3] * t8;
            const double fix = (theta * (1 + k0 + k1 + k2 + k3) - theta_d) / (1 + 3 * k0 + 5 * k1 + 7 * k2 + 9 * k3);
            theta = theta - fix;
            if (fabs(fix) < 1e-8) break;
        }
        scale = tan(theta) / theta_d;
    }
    *ox = wx * scale;
    *oy = wy * scale;
}

/* cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(K, D, (w, h), eye(3), balance=balance) (csi_camera.py:159) */
void oracle_fisheye_new_K(const double* K, const double* D, int w, int h, double balance, double* newK) {
    balance = fmin(fmax(balance, 0.0), 1.0);
    double p[4][2] = {{(double)(w / 2), 0}, {(double)w, (double)(h / 2)}, {(double)(w / 2), (double)h}, {0, (double)(h / 2)}};
    double cnx = 0, cny = 0;
    for (int i = 0; i < 4; ++i) {
        fisheye_undistort_point(K, D, p[i][0], p[i][1], &p[i][0], &p[i][1]);
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
    const double ncx = -cnx * f + w * 0.5, ncy = (-cny * f + h * aspect * 0.5) / aspect;
    const double out[9] = {f, 0, ncx, 0, f / aspect, ncy, 0, 0, 1};
    memcpy(newK, out, sizeof out);
}

static int cv_round(double v) { return (int)lrint(v); }   /* round half to even under the default rounding mode */

/* cv2.fisheye.initUndistortRectifyMap(K, D, eye(3), newK, (w, h), CV_16SC2) (csi_camera.py:160) */
void oracle_fisheye_maps(const double* K, const double* D, const double* newK, int w, int h, int16_t* map1, uint16_t* map2) {
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    /* inverse of [[a, s, c], [0, b, d], [0, 0, 1]] */
    const double a = newK[0], s = newK[1], c = newK[2], b = newK[4], d = newK[5];
    const double iR[9] = {1 / a, -s / (a * b), (s * d - c * b) / (a * b), 0, 1 / b, -d / b, 0, 0, 1};
    for (int i = 0; i < h; ++i) {
        double _x = i * iR[1] + iR[2], _y = i * iR[4] + iR[5], _w = i * iR[7] + iR[8];
        for (int j = 0; j < w; ++j) {
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
                u = fx * x * scale + cx;
                v = fy * y * scale + cy;
            }
            const int iu = cv_round(u * 32), iv = cv_round(v * 32);
            map1[((int64_t)i * w + j) * 2] = (int16_t)(iu >> 5);
            map1[((int64_t)i * w + j) * 2 + 1] = (int16_t)(iv >> 5);
            map2[(int64_t)i * w + j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
            _x += iR[0];
            _y += iR[3];
            _w += iR[6];
        }
    }
}

/* cv2.remap(src, map1, map2, INTER_AREA -> INTER_LINEAR, BORDER_CONSTANT, 0) for 8-bit images of `ch` interleaved
 * channels; the maps are [drows][dcols] (csi_camera.py:161) */
void oracle_remap_linear(const uint8_t* src, int rows, int cols, int ch, const int16_t* map1, const uint16_t* map2,
                         int drows, int dcols, uint8_t* dst) {
    for (int64_t p = 0; p < (int64_t)drows * dcols; ++p) {
        const int sx = map1[2 * p], sy = map1[2 * p + 1];
        const int fxq = map2[p] & 31, fyq = (map2[p] >> 5) & 31;
        int wgt[4] = {(32 - fxq) * (32 - fyq) * 32, fxq * (32 - fyq) * 32, (32 - fxq) * fyq * 32, fxq * fyq * 32};
        if (wgt[0] == 32768) { wgt[0] = 32767; wgt[3] = 1; }
        for (int k = 0; k < ch; ++k) {
            int acc = 0;
            for (int t = 0; t < 4; ++t) {
                const int x = sx + (t & 1), y = sy + (t >> 1);
                const int v = (x >= 0 && x < cols && y >= 0 && y < rows) ? src[((int64_t)y * cols + x) * ch + k] : 0;
                acc += v * wgt[t];
            }
            dst[p * ch + k] = (uint8_t)((acc + (1 << 14)) >> 15);
        }
    }
}
