/*
 * TEST INFRASTRUCTURE -- CPU oracle for the robust homography stage. Never imported by the product
 * path; used only by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED. The reference calls cv2.findHomography(src, dst, cv2.USAC_MAGSAC, 0.1)
 * (image_processing/src/image_processing/match_finder.py:233). OpenCV's USAC/MAGSAC++ (modules/
 * calib3d/src/usac/) is randomised, unpinned in version and absent from this image, and the reference
 * holds no fixture for it; SURVEY.md "Spec H" therefore fixes the contract as "the same geometric
 * answer within 1 px corner reprojection", not the same samples or mask. This file states the
 * estimator the build uses ("Spec H'"), in double precision, independently of the HIP/C++ code:
 *
 *   1. Hartley-normalise src and dst (centroid to origin, mean distance sqrt 2).
 *   2. M = 4096 hypotheses; hypothesis h draws 4 distinct indices with a counter-based splitmix64
 *      stream (seed, h); a 4-point homography is built in closed form from the two projective
 *      bases (H = B * adj(A)); samples with a near-collinear triple, or whose triples change
 *      orientation between src and dst, are skipped.
 *   3. Each hypothesis is scored in f32 by the MSAC sum  sum_i max(0, 1 - e_i^2 / tau^2),
 *      e = forward transfer error, tau = max(thr, 3 px); the best score wins, ties to the lower h.
 *   4. Refit on all points: 6 rounds of Tukey-weighted least squares (h33 = 1, 8x8 normal
 *      equations), then 4 Gauss-Newton steps on the weighted forward transfer error.
 *   5. Denormalise, scale to H[2][2] = 1; mask_i = (e_i <= thr) in pixels, as OpenCV's mask.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define HOMO_HYPOTHESES 4096
#define HOMO_TAU_FLOOR 3.0
#define HOMO_IRLS 6
#define HOMO_GN 4

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* 4 distinct indices in [0,n) for hypothesis h; returns 0 if 32 draws did not give 4 distinct */
static int sample4(uint64_t seed, uint32_t h, int n, int idx[4]) {
    int k = 0;
    for (uint32_t cnt = 0; cnt < 32 && k < 4; ++cnt) {
        const uint64_t u = splitmix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)h * 32 + cnt + 1));
        const int j = (int)(u % (uint64_t)n);
        int dup = 0;
        for (int t = 0; t < k; ++t) dup |= idx[t] == j;
        if (!dup) idx[k++] = j;
    }
    return k == 4;
}

static void hartley(const float* p, int n, double T[3]) { /* T = (scale, tx, ty): x' = s*(x - tx) */
    double cx = 0, cy = 0;
    for (int i = 0; i < n; ++i) { cx += p[2 * i]; cy += p[2 * i + 1]; }
    cx /= n; cy /= n;
    double md = 0;
    for (int i = 0; i < n; ++i) md += sqrt((p[2 * i] - cx) * (p[2 * i] - cx) + (p[2 * i + 1] - cy) * (p[2 * i + 1] - cy));
    md /= n;
    T[0] = md > 1e-12 ? sqrt(2.0) / md : 1.0;
    T[1] = cx; T[2] = cy;
}

/* projective basis: A maps e1,e2,e3,(1,1,1) to the 4 points; returns 0 when a triple is near-collinear.
 * sgn receives the signs of the four triple orientations. */
static int proj_basis(const double x[4], const double y[4], double A[9], int sgn[4]) {
    /* M = [p1 p2 p3], lambda = M^-1 p4 */
    const double c0 = y[1] - y[2], c1 = y[2] - y[0], c2 = y[0] - y[1];
    const double det = x[0] * c0 + x[1] * c1 + x[2] * c2;
    /* adj(M) * p4 */
    const double l0 = (y[1] - y[2]) * x[3] + (x[2] - x[1]) * y[3] + (x[1] * y[2] - x[2] * y[1]);
    const double l1 = (y[2] - y[0]) * x[3] + (x[0] - x[2]) * y[3] + (x[2] * y[0] - x[0] * y[2]);
    const double l2 = (y[0] - y[1]) * x[3] + (x[1] - x[0]) * y[3] + (x[0] * y[1] - x[1] * y[0]);
    const double eps = 1e-6;
    if (fabs(det) < eps || fabs(l0) < eps || fabs(l1) < eps || fabs(l2) < eps) return 0;
    sgn[0] = det > 0; sgn[1] = l0 > 0; sgn[2] = l1 > 0; sgn[3] = l2 > 0;
    const double a = l0 / det, b = l1 / det, c = l2 / det;
    A[0] = a * x[0]; A[1] = b * x[1]; A[2] = c * x[2];
    A[3] = a * y[0]; A[4] = b * y[1]; A[5] = c * y[2];
    A[6] = a;        A[7] = b;        A[8] = c;
    return 1;
}

static void adj3(const double A[9], double R[9]) {
    R[0] = A[4] * A[8] - A[5] * A[7]; R[1] = A[2] * A[7] - A[1] * A[8]; R[2] = A[1] * A[5] - A[2] * A[4];
    R[3] = A[5] * A[6] - A[3] * A[8]; R[4] = A[0] * A[8] - A[2] * A[6]; R[5] = A[2] * A[3] - A[0] * A[5];
    R[6] = A[3] * A[7] - A[4] * A[6]; R[7] = A[1] * A[6] - A[0] * A[7]; R[8] = A[0] * A[4] - A[1] * A[3];
}

static void mul3(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

/* returns 0 if the sample is degenerate; H (normalised coordinates), scaled to unit max-abs entry */
static int hyp_from_sample(const double* sx, const double* sy, const double* dx, const double* dy, const int idx[4], double H[9]) {
    double xs[4], ys[4], xd[4], yd[4], A[9], B[9], Ai[9];
    int s1[4], s2[4];
    for (int k = 0; k < 4; ++k) { xs[k] = sx[idx[k]]; ys[k] = sy[idx[k]]; xd[k] = dx[idx[k]]; yd[k] = dy[idx[k]]; }
    if (!proj_basis(xs, ys, A, s1) || !proj_basis(xd, yd, B, s2)) return 0;
    /* orientation of the four triples must be preserved (all the same or all flipped is a reflection: rejected) */
    for (int k = 0; k < 4; ++k) if (s1[k] != s2[k]) return 0;
    adj3(A, Ai);
    mul3(B, Ai, H);
    double m = 0;
    for (int k = 0; k < 9; ++k) m = fmax(m, fabs(H[k]));
    if (!(m > 0) || !isfinite(m)) return 0;
    for (int k = 0; k < 9; ++k) H[k] /= m;
    return 1;
}

/* MSAC score in f32, points in ascending order, unfused arithmetic */
static float msac_score(const float Hf[9], const float* sx, const float* sy, const float* dx, const float* dy, int n, float inv_tau2) {
    float score = 0.f;
    for (int i = 0; i < n; ++i) {
        const float w = Hf[6] * sx[i] + Hf[7] * sy[i] + Hf[8];
        const float u = (Hf[0] * sx[i] + Hf[1] * sy[i] + Hf[2]) / w;
        const float v = (Hf[3] * sx[i] + Hf[4] * sy[i] + Hf[5]) / w;
        const float ex = u - dx[i], ey = v - dy[i];
        const float t = 1.f - (ex * ex + ey * ey) * inv_tau2;
        if (t > 0.f) score += t;
    }
    return score;
}

static int solve8(double A[8][9]) { /* Gauss-Jordan with partial pivoting, solution in column 8 */
    for (int c = 0; c < 8; ++c) {
        int p = c;
        for (int r = c + 1; r < 8; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (fabs(A[p][c]) < 1e-14) return 0;
        if (p != c) for (int k = 0; k < 9; ++k) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
        const double inv = 1.0 / A[c][c];
        for (int k = c; k < 9; ++k) A[c][k] *= inv;
        for (int r = 0; r < 8; ++r) if (r != c) {
            const double f = A[r][c];
            if (f != 0) for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
        }
    }
    return 1;
}

/* src, dst: float [n][2]. H_out: 9 doubles (row-major, H[8] = 1). mask: n bytes or NULL.
 * Returns 1 with a model, 0 without (fewer than 4 points, or no non-degenerate sample). */
int oracle_homography(const float* src, const float* dst, int n, double thr, uint64_t seed, double* H_out, uint8_t* mask) {
    if (mask) memset(mask, 0, (size_t)(n > 0 ? n : 0));
    if (n < 4) return 0;
    double Ts[3], Td[3];
    hartley(src, n, Ts);
    hartley(dst, n, Td);
    double *sx = malloc(sizeof(double) * 4 * n), *sy = sx + n, *dx = sy + n, *dy = dx + n;
    float *fsx = malloc(sizeof(float) * 4 * n), *fsy = fsx + n, *fdx = fsy + n, *fdy = fdx + n;
    for (int i = 0; i < n; ++i) {
        sx[i] = Ts[0] * (src[2 * i] - Ts[1]); sy[i] = Ts[0] * (src[2 * i + 1] - Ts[2]);
        dx[i] = Td[0] * (dst[2 * i] - Td[1]); dy[i] = Td[0] * (dst[2 * i + 1] - Td[2]);
        fsx[i] = (float)sx[i]; fsy[i] = (float)sy[i]; fdx[i] = (float)dx[i]; fdy[i] = (float)dy[i];
    }
    const double tau = fmax(thr, HOMO_TAU_FLOOR) * Td[0]; /* in normalised dst units */
    const float inv_tau2 = (float)(1.0 / (tau * tau));
    double best_H[9];
    float best = -1.f;
    for (uint32_t h = 0; h < HOMO_HYPOTHESES; ++h) {
        int idx[4];
        double H[9];
        if (!sample4(seed, h, n, idx)) continue;
        if (!hyp_from_sample(sx, sy, dx, dy, idx, H)) continue;
        float Hf[9];
        for (int k = 0; k < 9; ++k) Hf[k] = (float)H[k];
        const float s = msac_score(Hf, fsx, fsy, fdx, fdy, n, inv_tau2);
        if (s > best) { best = s; memcpy(best_H, H, sizeof(H)); }
    }
    int ok = best >= 0.f && fabs(best_H[8]) > 1e-9;
    double h8[8];
    if (ok) for (int k = 0; k < 8; ++k) h8[k] = best_H[k] / best_H[8];
    double* w = malloc(sizeof(double) * n);
    const double tau2 = tau * tau;
    for (int it = 0; ok && it < HOMO_IRLS + HOMO_GN; ++it) {
        double N[8][9];
        memset(N, 0, sizeof(N));
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const double x = sx[i], y = sy[i], u = dx[i], v = dy[i];
            const double ww = h8[6] * x + h8[7] * y + 1.0;
            const double pu = (h8[0] * x + h8[1] * y + h8[2]) / ww, pv = (h8[3] * x + h8[4] * y + h8[5]) / ww;
            const double ex = u - pu, ey = v - pv, e2 = ex * ex + ey * ey;
            if (it < HOMO_IRLS || it == HOMO_IRLS) { /* weights are frozen during the Gauss-Newton steps */
                const double t = 1.0 - e2 / tau2;
                w[i] = t > 0 ? t * t : 0.0;
            }
            if (w[i] <= 0) continue;
            ++cnt;
            double r1[9], r2[9];
            if (it < HOMO_IRLS) { /* algebraic rows, h33 = 1 */
                r1[0] = x; r1[1] = y; r1[2] = 1; r1[3] = 0; r1[4] = 0; r1[5] = 0; r1[6] = -x * u; r1[7] = -y * u; r1[8] = u;
                r2[0] = 0; r2[1] = 0; r2[2] = 0; r2[3] = x; r2[4] = y; r2[5] = 1; r2[6] = -x * v; r2[7] = -y * v; r2[8] = v;
            } else { /* Jacobian of the projection, right-hand side = residual */
                const double iw = 1.0 / ww;
                r1[0] = x * iw; r1[1] = y * iw; r1[2] = iw; r1[3] = 0; r1[4] = 0; r1[5] = 0; r1[6] = -x * pu * iw; r1[7] = -y * pu * iw; r1[8] = ex;
                r2[0] = 0; r2[1] = 0; r2[2] = 0; r2[3] = x * iw; r2[4] = y * iw; r2[5] = iw; r2[6] = -x * pv * iw; r2[7] = -y * pv * iw; r2[8] = ey;
            }
            for (int a = 0; a < 8; ++a) {
                if (r1[a] != 0) for (int b = 0; b < 9; ++b) N[a][b] += w[i] * r1[a] * r1[b];
                if (r2[a] != 0) for (int b = 0; b < 9; ++b) N[a][b] += w[i] * r2[a] * r2[b];
            }
        }
        if (cnt < 4 || !solve8(N)) { ok = it > 0; break; }
        if (it < HOMO_IRLS) for (int k = 0; k < 8; ++k) h8[k] = N[k][8];
        else for (int k = 0; k < 8; ++k) h8[k] += N[k][8];
    }
    if (ok) {
        /* H = Td^-1 * Hn * Ts */
        const double Hn[9] = {h8[0], h8[1], h8[2], h8[3], h8[4], h8[5], h8[6], h8[7], 1.0};
        const double TS[9] = {Ts[0], 0, -Ts[0] * Ts[1], 0, Ts[0], -Ts[0] * Ts[2], 0, 0, 1};
        const double TDi[9] = {1.0 / Td[0], 0, Td[1], 0, 1.0 / Td[0], Td[2], 0, 0, 1};
        double t[9], H[9];
        mul3(Hn, TS, t);
        mul3(TDi, t, H);
        if (fabs(H[8]) < 1e-12) ok = 0;
        else {
            for (int k = 0; k < 9; ++k) H_out[k] = H[k] / H[8];
            if (mask)
                for (int i = 0; i < n; ++i) {
                    const double x = src[2 * i], y = src[2 * i + 1];
                    const double ww = H_out[6] * x + H_out[7] * y + 1.0;
                    const double ex = (H_out[0] * x + H_out[1] * y + H_out[2]) / ww - dst[2 * i];
                    const double ey = (H_out[3] * x + H_out[4] * y + H_out[5]) / ww - dst[2 * i + 1];
                    mask[i] = (ex * ex + ey * ey <= thr * thr) ? 1 : 0;
                }
        }
    }
    free(sx); free(fsx); free(w);
    return ok;
}
