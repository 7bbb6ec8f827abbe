/*
 * draw_oracle.c -- CPU statement of the match canvas: TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Follows the reference's debug canvas, match_finder.py:196:
 *   cv2.drawMatchesKnn(cadr.img, keypoints_1, roi.img, roi.kp, good, None, flags=NOT_DRAW_SINGLE_POINTS)
 * i.e. a [max(h1,h2)][w1+w2][3] 8-bit canvas, img1 on the left and img2 on the right (gray replicated to three
 * channels, the rest zero), and per good match a circle around both keypoints joined by a line in one colour.
 *
 * PARITY UNPINNED, and not pixel parity by design: OpenCV draws these with anti-aliased sub-pixel primitives in
 * colours taken from its global RNG, which no caller can reproduce (the canvas is only published for a human to look
 * at, position_finder.py:319-320). "Spec D", what this file and the HIP kernel both implement exactly:
 *   - end points p1 = (cvRound(x1), cvRound(y1)), p2 = (cvRound(x2) + w1, cvRound(y2)), coordinates held to +-2^20;
 *   - line: n = max(|dx|, |dy|) + 1 pixels, pixel k = p1 + round_half_up(k * d / (n - 1)) per coordinate (integer DDA);
 *   - circles: the pixels with (2r - 1)^2 <= 4 (dx^2 + dy^2) < (2r + 1)^2 around both end points, r = 3;
 *   - colour of match i: channel c = 64 + (mix32(i * 3 + c) % 192); where several matches cover a pixel the one
 *     with the highest index wins (cv2 draws them in order too).
 *
 * oracle_draw_overlay follows the reference's other drawing calls: the pose overlay of find_keypoints_transform,
 * match_finder.py:249-250 (cv2.circle(img, c, 10, 255, 5); cv2.polylines(img, [quad], True, 255, 3, LINE_AA)) and the map
 * marker of utils.py:75-76 (cv2.line(img, p, p2, color, 2); cv2.circle(img, p, 4, color, 2)). Integer coordinates as the
 * reference passes them; primitives are drawn in order. "Spec D" for them:
 *   - line of thickness t: the DDA line above, every pixel stamped with the disc brush 4 (u^2 + v^2) <= t^2,
 *     |u|, |v| <= t / 2 (integer division);
 *   - circle of radius r, thickness t > 0: the ring (2r - t)^2 <= 4 (u^2 + v^2) < (2r + t)^2 (no lower bound when
 *     2r <= t); t < 0: the filled disc u^2 + v^2 <= r^2;
 *   - colour: the primitive's channel values as given (cv2's Scalar), channel 0 for a 1-channel image.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

static void put(int32_t* owner, int H, int W, int x, int y, int i) {
    if (x >= 0 && x < W && y >= 0 && y < H && owner[(int64_t)y * W + x] < i) owner[(int64_t)y * W + x] = i;
}

/* cvRound of a coordinate held to +-2^20 (NaN counts as the lower bound), so the DDA below stays in range */
static int px(float v) { return (int)lrintf(fminf(fmaxf(v, -1048576.f), 1048576.f)); }

/* floor((2 k d + m) / (2 m)) for m > 0: k d / m rounded half up */
static int dda(int k, int d, int m) {
    const long long num = 2LL * k * d + m, den = 2LL * m;
    long long q = num / den;
    if (num % den != 0 && num < 0) --q;
    return (int)q;
}

void oracle_draw_matches(const uint8_t* img1, int h1, int w1, const uint8_t* img2, int h2, int w2, const float* xy1,
                         const float* xy2, int n, uint8_t* out) {
    const int H = h1 > h2 ? h1 : h2, W = w1 + w2, r = 3;
    int32_t* owner = (int32_t*)malloc((size_t)H * W * sizeof(int32_t));
    memset(owner, 0xff, (size_t)H * W * sizeof(int32_t));
    for (int i = 0; i < n; ++i) {
        const int x0 = px(xy1[2 * i]), y0 = px(xy1[2 * i + 1]);
        const int x1 = px(xy2[2 * i]) + w1, y1 = px(xy2[2 * i + 1]);
        const int dx = x1 - x0, dy = y1 - y0;
        const int m = abs(dx) > abs(dy) ? abs(dx) : abs(dy);
        for (int k = 0; k <= m; ++k) put(owner, H, W, x0 + (m ? dda(k, dx, m) : 0), y0 + (m ? dda(k, dy, m) : 0), i);
        for (int e = 0; e < 2; ++e)
            for (int v = -r - 1; v <= r + 1; ++v)
                for (int u = -r - 1; u <= r + 1; ++u) {
                    const int d4 = 4 * (u * u + v * v);
                    if (d4 >= (2 * r - 1) * (2 * r - 1) && d4 < (2 * r + 1) * (2 * r + 1))
                        put(owner, H, W, (e ? x1 : x0) + u, (e ? y1 : y0) + v, i);
                }
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t* o = out + ((int64_t)y * W + x) * 3;
            const int i = owner[(int64_t)y * W + x];
            if (i >= 0) {
                for (int c = 0; c < 3; ++c) o[c] = (uint8_t)(64 + mix32((uint32_t)i * 3u + (uint32_t)c) % 192u);
            } else {
                uint8_t g = 0;
                if (x < w1) { if (y < h1) g = img1[(int64_t)y * w1 + x]; }
                else if (y < h2) g = img2[(int64_t)y * w2 + (x - w1)];
                o[0] = o[1] = o[2] = g;
            }
        }
    free(owner);
}

/* prims: int32 [n][8] = kind (0 line, 1 circle), x0, y0, x1 (circle: radius), y1, thickness, colour bytes (4), reserved */
void oracle_draw_overlay(const uint8_t* img, int h, int w, int ch, const int32_t* prims, int n, uint8_t* out) {
    int32_t* owner = (int32_t*)malloc((size_t)h * w * sizeof(int32_t));
    memset(owner, 0xff, (size_t)h * w * sizeof(int32_t));
    for (int i = 0; i < n; ++i) {
        const int32_t* p = prims + 8 * i;
        const int x0 = p[1], y0 = p[2], t = p[5];
        if (p[0] == 0) {
            const int dx = p[3] - x0, dy = p[4] - y0;
            const int m = abs(dx) > abs(dy) ? abs(dx) : abs(dy);
            const int hb = t / 2;
            for (int k = 0; k <= m; ++k) {
                const int cx = x0 + (m ? dda(k, dx, m) : 0), cy = y0 + (m ? dda(k, dy, m) : 0);
                for (int v = -hb; v <= hb; ++v)
                    for (int u = -hb; u <= hb; ++u)
                        if (4 * (u * u + v * v) <= t * t) put(owner, h, w, cx + u, cy + v, i);
            }
        } else {
            const int r = p[3];
            const int R = r + (t > 0 ? (t + 1) / 2 : 0) + 1;
            const long long lo = (t > 0 && 2 * r > t) ? (long long)(2 * r - t) * (2 * r - t) : 0;
            const long long hi = t > 0 ? (long long)(2 * r + t) * (2 * r + t) : 4LL * r * r + 1;
            for (int v = -R; v <= R; ++v)
                for (int u = -R; u <= R; ++u) {
                    const long long d4 = 4LL * (u * u + v * v);
                    if (d4 >= lo && d4 < hi) put(owner, h, w, x0 + u, y0 + v, i);
                }
        }
    }
    for (int64_t q = 0; q < (int64_t)h * w; ++q) {
        const int i = owner[q];
        for (int c = 0; c < ch; ++c) out[q * ch + c] = i >= 0 ? ((const uint8_t*)(prims + 8 * i + 6))[c] : img[q * ch + c];
    }
    free(owner);
}
