// Robust homography: replaces cv2.findHomography(src, dst, cv2.USAC_MAGSAC, 0.1)
// (reference match_finder.py:233). OpenCV's USAC sampler is sequential; here all hypotheses are
// generated and scored at once: one thread per hypothesis builds a 4-point homography in closed
// form (two projective bases, f64) and scores it against every correspondence (MSAC sum, f32);
// a 64-bit atomicMax on (score bits, ~h) picks the winner deterministically. The winner is refit
// on the host (Tukey-weighted least squares + Gauss-Newton, f64, a few hundred points).
#include "common.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <thread>
#include <vector>

#define HOMO_HYPOTHESES 4096
#define HOMO_TAU_FLOOR 3.0
#define HOMO_IRLS 6
#define HOMO_GN 4

__host__ __device__ static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ static inline bool sample4(uint64_t seed, uint32_t h, int n, int idx[4]) {
    int k = 0;
    for (uint32_t cnt = 0; cnt < 32 && k < 4; ++cnt) {
        const uint64_t u = splitmix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)h * 32 + cnt + 1));
        const int j = (int)(u % (uint64_t)n);
        bool dup = false;
        for (int t = 0; t < k; ++t) dup |= idx[t] == j;
        if (!dup) idx[k++] = j;
    }
    return k == 4;
}

// A maps e1, e2, e3, (1,1,1) to the four points; false when a triple is near-collinear
__host__ __device__ static inline bool proj_basis(const double x[4], const double y[4], double A[9], int& sgn) {
    const double c0 = y[1] - y[2], c1 = y[2] - y[0], c2 = y[0] - y[1];
    const double det = x[0] * c0 + x[1] * c1 + x[2] * c2;
    const double l0 = (y[1] - y[2]) * x[3] + (x[2] - x[1]) * y[3] + (x[1] * y[2] - x[2] * y[1]);
    const double l1 = (y[2] - y[0]) * x[3] + (x[0] - x[2]) * y[3] + (x[2] * y[0] - x[0] * y[2]);
    const double l2 = (y[0] - y[1]) * x[3] + (x[1] - x[0]) * y[3] + (x[0] * y[1] - x[1] * y[0]);
    const double eps = 1e-6;
    if (fabs(det) < eps || fabs(l0) < eps || fabs(l1) < eps || fabs(l2) < eps) return false;
    sgn = (det > 0 ? 1 : 0) | (l0 > 0 ? 2 : 0) | (l1 > 0 ? 4 : 0) | (l2 > 0 ? 8 : 0);
    const double a = l0 / det, b = l1 / det, c = l2 / det;
    A[0] = a * x[0]; A[1] = b * x[1]; A[2] = c * x[2];
    A[3] = a * y[0]; A[4] = b * y[1]; A[5] = c * y[2];
    A[6] = a;        A[7] = b;        A[8] = c;
    return true;
}

__host__ __device__ static inline void mul3(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// pts: double [4][n] = sx, sy, dx, dy in normalised coordinates
__host__ __device__ static inline bool hypothesis(const double* pts, int n, uint64_t seed, uint32_t h, double H[9]) {
    int idx[4];
    if (!sample4(seed, h, n, idx)) return false;
    double xs[4], ys[4], xd[4], yd[4], A[9], B[9], Ai[9];
    for (int k = 0; k < 4; ++k) {
        xs[k] = pts[idx[k]]; ys[k] = pts[n + idx[k]]; xd[k] = pts[2 * n + idx[k]]; yd[k] = pts[3 * n + idx[k]];
    }
    int s1, s2;
    if (!proj_basis(xs, ys, A, s1) || !proj_basis(xd, yd, B, s2)) return false;
    if (s1 != s2) return false;  // a triple changes orientation: not a physical view
    Ai[0] = A[4] * A[8] - A[5] * A[7]; Ai[1] = A[2] * A[7] - A[1] * A[8]; Ai[2] = A[1] * A[5] - A[2] * A[4];
    Ai[3] = A[5] * A[6] - A[3] * A[8]; Ai[4] = A[0] * A[8] - A[2] * A[6]; Ai[5] = A[2] * A[3] - A[0] * A[5];
    Ai[6] = A[3] * A[7] - A[4] * A[6]; Ai[7] = A[1] * A[6] - A[0] * A[7]; Ai[8] = A[0] * A[4] - A[1] * A[3];
    mul3(B, Ai, H);
    double m = 0;
    for (int k = 0; k < 9; ++k) m = fmax(m, fabs(H[k]));
    if (!(m > 0) || !isfinite(m)) return false;
    for (int k = 0; k < 9; ++k) H[k] /= m;
    return true;
}

// grid (HOMO_HYPOTHESES / 256, segments): segment b (= image b of a batch; a single call is one
// segment) holds n_b = off[b+1] - off[b] correspondences, its normalised points at pts + 4 * off[b]
// doubles ([4][n_b]: sx, sy, dx, dy) and fpts + 4 * off[b] floats, its MSAC scale in inv_tau2[b] and
// its sampler seed in seeds[b].
// (seeds and scales travel in the uploaded block behind the points: with 256 segments they would not fit the 4 KB of
// kernel arguments beside the offsets)
struct HomoSeg {
    imatch_segtab seg;
    const float* inv_tau2;                 // [segments], device
    const unsigned long long* seeds;       // [segments], device
};

__global__ void __launch_bounds__(256)
homography_score_kernel(const double* __restrict__ pts_all, const float* __restrict__ fpts_all, HomoSeg hs,
                        unsigned long long* __restrict__ best, const int* __restrict__ n_dev) {
    const int b = blockIdx.y;
    // n_dev: a single problem whose size the host has not seen yet (imatch_match_localize); 0 = nothing to score
    const int n = n_dev ? *n_dev : hs.seg.off[b + 1] - hs.seg.off[b];
    if (n < 4) return;
    const double* pts = pts_all + 4 * (size_t)hs.seg.off[b];
    const float* fpts = fpts_all + 4 * (size_t)hs.seg.off[b];
    const uint64_t seed = hs.seeds[b];
    const float inv_tau2 = hs.inv_tau2[b];
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= HOMO_HYPOTHESES) return;
    double H[9];
    if (!hypothesis(pts, n, seed, h, H)) return;
    float Hf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Hf[k] = (float)H[k];
    const float *sx = fpts, *sy = fpts + n, *dx = fpts + 2 * n, *dy = fpts + 3 * n;
    float score = 0.f;
    for (int i = 0; i < n; ++i) {
        const float w = Hf[6] * sx[i] + Hf[7] * sy[i] + Hf[8];
        const float u = (Hf[0] * sx[i] + Hf[1] * sy[i] + Hf[2]) / w;
        const float v = (Hf[3] * sx[i] + Hf[4] * sy[i] + Hf[5]) / w;
        const float ex = u - dx[i], ey = v - dy[i];
        const float t = 1.f - (ex * ex + ey * ey) * inv_tau2;
        if (t > 0.f) score += t;
    }
    // scores are >= 0: their bit patterns order like the floats; ties go to the lower h
    const unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (0xFFFFFFFFu - h);
    atomicMax(best + b, key);
}

// The same scores for a single problem (or a few), where a thread per hypothesis leaves the chip idle: a frame's 4096
// hypotheses are 64 waves of one long serial loop each (25-35 us for 100 correspondences, 130 us for 1500). Here a workgroup
// takes 16 hypotheses: sixteen lanes build them (the closed form is a long chain of f64 divisions -- a wave per hypothesis
// building its own was slower than the thread-per-hypothesis kernel), then each wave scores four of them with its 64 lanes
// on 64 correspondences at a time, and the terms are added in index order, one v_readlane + one v_add_f32 each: the sum
// every thread of homography_score_kernel forms, bit for bit (a term that is not > 0 is added as + 0, which leaves a
// non-negative sum as it is). A batch keeps the thread-per-hypothesis kernel: its million hypotheses fill the chip as they are.
#define HOMO_WG_HYP 16
__global__ void __launch_bounds__(256)
homography_score_wg_kernel(const double* __restrict__ pts_all, const float* __restrict__ fpts_all, HomoSeg hs,
                           unsigned long long* __restrict__ best) {
    __shared__ float Hs[HOMO_WG_HYP][9];
    __shared__ int ok[HOMO_WG_HYP];
    const int b = blockIdx.y;
    const int n = hs.seg.off[b + 1] - hs.seg.off[b];
    if (n < 4) return;
    const double* pts = pts_all + 4 * (size_t)hs.seg.off[b];
    const float* fpts = fpts_all + 4 * (size_t)hs.seg.off[b];
    const float inv_tau2 = hs.inv_tau2[b];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t h0 = blockIdx.x * HOMO_WG_HYP;
    if (threadIdx.x < HOMO_WG_HYP) {
        const uint32_t h = h0 + threadIdx.x;
        double H[9];
        const bool good = h < HOMO_HYPOTHESES && hypothesis(pts, n, hs.seeds[b], h, H);
        ok[threadIdx.x] = good;
        if (good)
            for (int k = 0; k < 9; ++k) Hs[threadIdx.x][k] = (float)H[k];
    }
    __syncthreads();
    const float *sx = fpts, *sy = fpts + n, *dx = fpts + 2 * n, *dy = fpts + 3 * n;
    for (int j = 0; j < HOMO_WG_HYP / 4; ++j) {
        const int hl = wave * (HOMO_WG_HYP / 4) + j;     // wave-uniform
        if (!ok[hl]) continue;
        float Hf[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) Hf[k] = Hs[hl][k];
        float score = 0.f;
        for (int base = 0; base < n; base += 64) {
            const int i = base + lane;
            float t = 0.f;
            if (i < n) {
                const float w = Hf[6] * sx[i] + Hf[7] * sy[i] + Hf[8];
                const float u = (Hf[0] * sx[i] + Hf[1] * sy[i] + Hf[2]) / w;
                const float v = (Hf[3] * sx[i] + Hf[4] * sy[i] + Hf[5]) / w;
                const float ex = u - dx[i], ey = v - dy[i];
                t = 1.f - (ex * ex + ey * ey) * inv_tau2;
                if (!(t > 0.f)) t = 0.f;
            }
#pragma unroll
            for (int k = 0; k < 64; ++k) score += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), k));
        }
        if (lane == 0) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(score) << 32) | (0xFFFFFFFFu - (h0 + (uint32_t)hl));
            atomicMax(best + b, key);
        }
    }
}

static void hartley(const float* p, int n, double T[3]) {
    double cx = 0, cy = 0;
    for (int i = 0; i < n; ++i) { cx += p[2 * i]; cy += p[2 * i + 1]; }
    cx /= n; cy /= n;
    double md = 0;
    for (int i = 0; i < n; ++i) md += sqrt((p[2 * i] - cx) * (p[2 * i] - cx) + (p[2 * i + 1] - cy) * (p[2 * i + 1] - cy));
    md /= n;
    T[0] = md > 1e-12 ? sqrt(2.0) / md : 1.0;
    T[1] = cx; T[2] = cy;
}

static bool solve8(double A[8][9]) {
    for (int c = 0; c < 8; ++c) {
        int p = c;
        for (int r = c + 1; r < 8; ++r) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
        if (fabs(A[p][c]) < 1e-14) return false;
        if (p != c) for (int k = 0; k < 9; ++k) std::swap(A[c][k], A[p][k]);
        const double inv = 1.0 / A[c][c];
        for (int k = c; k < 9; ++k) A[c][k] *= inv;
        for (int r = 0; r < 8; ++r) if (r != c) {
            const double f = A[r][c];
            if (f != 0) for (int k = c; k < 9; ++k) A[r][k] -= f * A[c][k];
        }
    }
    return true;
}

// Tukey-weighted least squares rounds, then Gauss-Newton on the forward transfer error. The normal equations
//     N[a][b] = sum_i w_i r1_i[a] r1_i[b] + w_i r2_i[a] r2_i[b],   r1 = (p0 p1 p2 0 0 0 p3 p4 p5),  r2 = (0 0 0 q0 q1 q2 q3 q4 q5)
// are summed point after point in index order, every term as (w r[a]) r[b] and r1's before r2's, exactly as
// oracle/homography_oracle.c does it -- but only where the rows are not structurally zero (a term w r[a] 0 adds +-0 and
// changes nothing), four columns per vector operation, and with the per-point part (projection, error, weight: four
// divisions) in a loop of its own that the compiler vectorises over points. Same IEEE operations on the same operands in the
// same order: the result is the scalar loop's bit for bit (a 1 500-point refit 0.49 -> 0.18 ms, a 115-point one 37 -> 14 us
// on the development host; an AVX2 clone is picked at load time, no fused multiply-add in either).
#if defined(__HIP_DEVICE_COMPILE__)
#define IMATCH_HOST_CLONES
#else
#define IMATCH_HOST_CLONES __attribute__((target_clones("avx2", "default")))
#endif
typedef double v4d __attribute__((vector_size(32)));
#define V4D_ALL(s) ((v4d){(s), (s), (s), (s)})

// ws: [6][n] doubles of scratch
IMATCH_HOST_CLONES
static bool refit(const double* __restrict__ P, int n, double tau2, double h8[8], double* __restrict__ ws) {
    const double *sx = P, *sy = P + n, *dx = P + 2 * n, *dy = P + 3 * n;
    double *w = ws, *a_ww = ws + n, *a_pu = ws + 2 * (size_t)n, *a_pv = ws + 3 * (size_t)n, *a_ex = ws + 4 * (size_t)n,
           *a_ey = ws + 5 * (size_t)n;
    for (int i = 0; i < n; ++i) w[i] = 0.0;
    for (int it = 0; it < HOMO_IRLS + HOMO_GN; ++it) {
        {
            const double h0 = h8[0], h1 = h8[1], h2 = h8[2], h3 = h8[3], h4 = h8[4], h5 = h8[5], h6 = h8[6], h7 = h8[7];
            const bool reweigh = it <= HOMO_IRLS;  // frozen during the remaining Gauss-Newton steps
            for (int i = 0; i < n; ++i) {
                const double x = sx[i], y = sy[i];
                const double ww = h6 * x + h7 * y + 1.0;
                const double pu = (h0 * x + h1 * y + h2) / ww, pv = (h3 * x + h4 * y + h5) / ww;
                const double ex = dx[i] - pu, ey = dy[i] - pv, e2 = ex * ex + ey * ey;
                a_ww[i] = ww; a_pu[i] = pu; a_pv[i] = pv; a_ex[i] = ex; a_ey[i] = ey;
                if (reweigh) {
                    const double t = 1.0 - e2 / tau2;
                    w[i] = t > 0 ? t * t : 0.0;
                }
            }
        }
        // rows 0-2 on columns {0,1,2} (Al) and {6,7,8} (Ah), rows 3-5 on {3,4,5} (Bl) and {6,7,8} (Bh), rows 6-7 on all nine
        const v4d z = V4D_ALL(0.0);
        v4d Al[3] = {z, z, z}, Ah[3] = {z, z, z}, Bl[3] = {z, z, z}, Bh[3] = {z, z, z}, D0[2] = {z, z}, D3[2] = {z, z}, D6[2] = {z, z};
        int cnt = 0;
        for (int i = 0; i < n; ++i) {
            const double wi = w[i];
            if (!(wi > 0)) continue;
            ++cnt;
            const double x = sx[i], y = sy[i], u = dx[i], v = dy[i];
            double p[6], q[6];
            if (it < HOMO_IRLS) {
                p[0] = x; p[1] = y; p[2] = 1; p[3] = -x * u; p[4] = -y * u; p[5] = u;
                q[0] = x; q[1] = y; q[2] = 1; q[3] = -x * v; q[4] = -y * v; q[5] = v;
            } else {
                const double iw = 1.0 / a_ww[i], pu = a_pu[i], pv = a_pv[i];
                p[0] = x * iw; p[1] = y * iw; p[2] = iw; p[3] = -x * pu * iw; p[4] = -y * pu * iw; p[5] = a_ex[i];
                q[0] = x * iw; q[1] = y * iw; q[2] = iw; q[3] = -x * pv * iw; q[4] = -y * pv * iw; q[5] = a_ey[i];
            }
            const v4d pl = {p[0], p[1], p[2], 0.0}, ph = {p[3], p[4], p[5], 0.0};
            const v4d ql = {q[0], q[1], q[2], 0.0}, qh = {q[3], q[4], q[5], 0.0};
            for (int a = 0; a < 3; ++a) {
                const v4d wa = V4D_ALL(wi * p[a]), wb = V4D_ALL(wi * q[a]);
                Al[a] += wa * pl; Ah[a] += wa * ph;
                Bl[a] += wb * ql; Bh[a] += wb * qh;
            }
            for (int a = 0; a < 2; ++a) {
                const v4d wa = V4D_ALL(wi * p[3 + a]), wb = V4D_ALL(wi * q[3 + a]);
                D0[a] += wa * pl; D3[a] += wb * ql;
                D6[a] += wa * ph; D6[a] += wb * qh;   // r1's term first
            }
        }
        double N[8][9];
        memset(N, 0, sizeof(N));
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) {
                N[a][b] = Al[a][b]; N[a][6 + b] = Ah[a][b];
                N[3 + a][3 + b] = Bl[a][b]; N[3 + a][6 + b] = Bh[a][b];
            }
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 3; ++b) { N[6 + a][b] = D0[a][b]; N[6 + a][3 + b] = D3[a][b]; N[6 + a][6 + b] = D6[a][b]; }
        if (cnt < 4 || !solve8(N)) return it > 0;
        if (it < HOMO_IRLS) for (int k = 0; k < 8; ++k) h8[k] = N[k][8];
        else for (int k = 0; k < 8; ++k) h8[k] += N[k][8];
    }
    return true;
}

// Everything after the scoring of one problem of n >= 4 correspondences: the winner is built again from its index (the
// code the device ran), refit, taken back to pixel coordinates; mask = forward error <= thr as OpenCV's. P: the normalised
// points [4][n], sp / dp: the pixel coordinates [n][2], ts / td: the two Hartley transforms (scale, centre x, centre y).
static void finish_problem(const double* P, const float* sp, const float* dp, int n, const double* ts, const double* td,
                           double tau, unsigned long long key, uint64_t seed, double thr, double* Ho, uint8_t* mask, int* found) {
    if (key == 0) return;  // no non-degenerate sample
    const uint32_t h = 0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu);
    double Hb[9];
    if (!hypothesis(P, n, seed, h, Hb) || fabs(Hb[8]) <= 1e-9) return;
    double h8[8];
    for (int k = 0; k < 8; ++k) h8[k] = Hb[k] / Hb[8];
    std::vector<double> scratch(6 * (size_t)n);
    if (!refit(P, n, tau * tau, h8, scratch.data())) return;
    const double Hn[9] = {h8[0], h8[1], h8[2], h8[3], h8[4], h8[5], h8[6], h8[7], 1.0};
    const double TS[9] = {ts[0], 0, -ts[0] * ts[1], 0, ts[0], -ts[0] * ts[2], 0, 0, 1};
    const double TDi[9] = {1.0 / td[0], 0, td[1], 0, 1.0 / td[0], td[2], 0, 0, 1};
    double t[9], H[9];
    mul3(Hn, TS, t);
    mul3(TDi, t, H);
    if (fabs(H[8]) < 1e-12) return;
    for (int k = 0; k < 9; ++k) Ho[k] = H[k] / H[8];
    if (mask)
        for (int i = 0; i < n; ++i) {
            const double x = sp[2 * i], y = sp[2 * i + 1];
            const double ww = Ho[6] * x + Ho[7] * y + 1.0;
            const double ex = (Ho[0] * x + Ho[1] * y + Ho[2]) / ww - dp[2 * i];
            const double ey = (Ho[3] * x + Ho[4] * y + Ho[5]) / ww - dp[2 * i + 1];
            mask[i] = (ex * ex + ey * ey <= thr * thr) ? 1 : 0;
        }
    *found = 1;
}

// normalised points of one problem: P [4][n] doubles (sx, sy, dx, dy), F the same rounded to f32
static void normalise_points(const float* sp, const float* dp, int n, const double* ts, const double* td, double* P, float* F) {
    for (int i = 0; i < n; ++i) {
        P[i] = ts[0] * (sp[2 * i] - ts[1]); P[n + i] = ts[0] * (sp[2 * i + 1] - ts[2]);
        P[2 * n + i] = td[0] * (dp[2 * i] - td[1]); P[3 * n + i] = td[0] * (dp[2 * i + 1] - td[2]);
    }
    if (F)
        for (int i = 0; i < 4 * n; ++i) F[i] = (float)P[i];
}

// Shared by the single and the batched entry point: segment b's correspondences are src/dst rows
// [off[b], off[b+1]).
static int homography_run(imatch_ctx* ctx, const float* src, const float* dst, const imatch_segtab& seg, double thr,
                          const uint64_t* seeds, double* H_out /* [n_seg][9] */, uint8_t* mask, int* found) {
    const int nseg = seg.n;
    const int64_t ntot = seg.off[nseg];
    for (int b = 0; b < nseg; ++b) found[b] = 0;
    if (mask && ntot > 0) memset(mask, 0, (size_t)ntot);
    bool any = false;
    for (int b = 0; b < nseg; ++b) any = any || seg.off[b + 1] - seg.off[b] >= 4;
    if (!any) return IMATCH_OK;  // cv2.findHomography returns (None, None)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t dbytes = sizeof(double) * 4 * (size_t)ntot, fbytes = sizeof(float) * 4 * (size_t)ntot;
    const size_t kbytes = 8 * (size_t)nseg;
    // pinned / device block: points f64, points f32, seeds, scales (uploaded in one copy), then the result keys
    const size_t sbytes = 8 * (size_t)nseg, tbytes = (4 * (size_t)nseg + 7) & ~(size_t)7;
    const size_t up_bytes = dbytes + fbytes + sbytes + tbytes;
    int rc = imatch_hs_reserve(ctx, up_bytes + kbytes + 64);
    if (rc) return rc;
    rc = imatch_ws_reserve(ctx, up_bytes + kbytes + 256);
    if (rc) return rc;
    double* Pall = (double*)ctx->hs;
    float* Fall = (float*)((char*)ctx->hs + dbytes);
    unsigned long long* h_seeds = (unsigned long long*)((char*)ctx->hs + dbytes + fbytes);
    float* h_itau2 = (float*)((char*)ctx->hs + dbytes + fbytes + sbytes);
    memset(h_itau2, 0, tbytes);
    HomoSeg hs;
    memset(&hs, 0, sizeof(hs));
    hs.seg = seg;
    std::vector<double> Ts(3 * (size_t)nseg), Td(3 * (size_t)nseg), tau((size_t)nseg, 0.0);
    for (int b = 0; b < nseg; ++b) {
        const int n = seg.off[b + 1] - seg.off[b];
        h_seeds[b] = seeds[b];
        if (n < 4) continue;
        const float* sp = src + 2 * (size_t)seg.off[b];
        const float* dp = dst + 2 * (size_t)seg.off[b];
        double* ts = &Ts[3 * b];
        double* td = &Td[3 * b];
        hartley(sp, n, ts);
        hartley(dp, n, td);
        normalise_points(sp, dp, n, ts, td, Pall + 4 * (size_t)seg.off[b], Fall + 4 * (size_t)seg.off[b]);
        tau[b] = fmax(thr, HOMO_TAU_FLOOR) * td[0];
        h_itau2[b] = (float)(1.0 / (tau[b] * tau[b]));
    }
    char* dev = (char*)ctx->ws;
    unsigned long long* d_best = (unsigned long long*)(dev + up_bytes);
    hs.seeds = (const unsigned long long*)(dev + dbytes + fbytes);
    hs.inv_tau2 = (const float*)(dev + dbytes + fbytes + sbytes);
    // the result keys start at zero: they travel up as zeros behind the points (one operation instead of copy + clear) and
    // come back into the same pinned words
    unsigned long long* h_key = (unsigned long long*)((char*)ctx->hs + up_bytes);
    memset(h_key, 0, kbytes);
    HIP_TRY(ctx, hipMemcpyAsync(dev, ctx->hs, up_bytes + kbytes, hipMemcpyHostToDevice, ctx->stream));
    // one problem (or a few): sixteen hypotheses per workgroup, scored by its waves; a batch: a thread per hypothesis (same
    // scores bit for bit)
    static const int wave_env = getenv("IMATCH_HOMO_WAVE") ? atoi(getenv("IMATCH_HOMO_WAVE")) : 1;   // 0: A/B runs
    (void)imatch_stage_begin(ctx, IMATCH_STAGE_HOMOGRAPHY, ctx->stream);
    // (with a hundred correspondences the hypotheses' construction, a serial f64 chain, is most of either kernel and the
    // thread-per-hypothesis form is the shorter one: 93 against 121 us per call; with 1500 it is 354 against 288)
    int n_max = 0;
    for (int b = 0; b < nseg; ++b) n_max = std::max(n_max, seg.off[b + 1] - seg.off[b]);
    if (nseg < 8 && wave_env && n_max >= 512)
        hipLaunchKernelGGL(homography_score_wg_kernel, dim3(HOMO_HYPOTHESES / HOMO_WG_HYP, nseg), dim3(256), 0, ctx->stream,
                           (const double*)dev, (const float*)(dev + dbytes), hs, d_best);
    else
        hipLaunchKernelGGL(homography_score_kernel, dim3(HOMO_HYPOTHESES / (nseg >= 8 ? 256 : 64), nseg), dim3(nseg >= 8 ? 256 : 64), 0,
                           ctx->stream, (const double*)dev, (const float*)(dev + dbytes), hs, d_best, (const int*)nullptr);
    HIP_TRY(ctx, hipGetLastError());
    (void)imatch_stage_end(ctx, IMATCH_STAGE_HOMOGRAPHY, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(h_key, d_best, kbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    auto finish = [&](int b) {
        const int n = seg.off[b + 1] - seg.off[b];
        if (n < 4) return;
        finish_problem(Pall + 4 * (size_t)seg.off[b], src + 2 * (size_t)seg.off[b], dst + 2 * (size_t)seg.off[b], n, &Ts[3 * b],
                       &Td[3 * b], tau[b], h_key[b], seeds[b], thr, H_out + 9 * (size_t)b, mask ? mask + seg.off[b] : nullptr,
                       found + b);
    };
    if (nseg >= 4) {  // the problems are independent: refit them on the context's host workers
        imatch_parallel_for(ctx, nseg, [&](int b) { finish(b); });
    } else {
        for (int b = 0; b < nseg; ++b) finish(b);
    }
    return IMATCH_OK;
}

// ---- one frame's match and localisation without a host round trip in the middle (imatch_match_localize) ---------------
// The per-frame calls hand the ratio test's survivors to the host, which looks up their coordinates, normalises them and
// sends them back for the scoring: a copy, a wake-up, host work and another copy, 40 us of a 0.58 ms frame in which the GPU
// waits. Here one workgroup does all of it where the kNN's result lies: the ratio test and the compaction in query order
// (ratio_kernel's, knn.hip), the gather of both sides' coordinates from the sets' tables, the two Hartley transforms --
// the centre and the mean distance are sums in index order, so one lane adds them up, the other lanes supply the square
// roots -- and the normalised points; the scoring kernel reads the count from the device. The arithmetic is the host's
// (hartley, normalise_points: IEEE double operations in the same order, no contraction in either build), so the points,
// the scores and the winner are those of the separate calls; the host still checks the transforms that come back
// against its own, bit for bit, and takes the separate path when they differ or when a list does not fit.
#define TAIL_CAP 1024
struct TailHdr {
    int n_good;      // survivors of the ratio test; beyond cap nothing else is valid
    int scored;      // 1: the points were normalised and the hypotheses scored
    double ts[3], td[3], tau;
    unsigned long long key;   // the scoring kernel's atomicMax word
    unsigned long long seed;
    float inv_tau2;
    int n_score;     // what the scoring kernel reads: n_good, or 0
};

__global__ void __launch_bounds__(1024)
match_tail_kernel(const int4* __restrict__ in, int nq, double ratio, const float2* __restrict__ q_xy,
                  const float2* __restrict__ t_xy, int cap, int min_good, double thr, unsigned long long seed,
                  TailHdr* __restrict__ hdr, int* __restrict__ gq, int* __restrict__ gt, float* __restrict__ gd,
                  double* __restrict__ P, float* __restrict__ F) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    __shared__ float2 sxy[TAIL_CAP], dxy[TAIL_CAP];
    __shared__ double term[2][TAIL_CAP];
    __shared__ double T[2][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {   // ratio_kernel's loop, one segment
        const int q = q0 + threadIdx.x;
        bool good = false;
        int4 e = make_int4(0, -1, 0, -1);
        float d1 = 0.f;
        if (q < nq) {
            e = in[q];
            if (e.y >= 0 && e.w >= 0) {
                d1 = sqrtf((float)e.x);
                const float d2 = sqrtf((float)e.z);
                good = (double)d1 < ratio * (double)d2;
            }
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(good);
        const int within = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (good && off + within < cap) {
            gq[off + within] = q;
            gt[off + within] = e.y;
            gd[off + within] = d1;
            sxy[off + within] = q_xy[q];
            dxy[off + within] = t_xy[e.y];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base_s += t;
        }
        __syncthreads();
    }
    const int n = base_s;
    const bool go = n > min_good && n >= 4 && n <= cap;
    if (threadIdx.x == 0) {
        hdr->n_good = n;
        hdr->scored = go ? 1 : 0;
        hdr->key = 0ull;
        hdr->seed = seed;
        if (!go) hdr->n_score = 0;
    }
    if (!go) return;
    // hartley(): centre = the coordinates' sum in index order / n
    if (lane == 0 && wave < 2) {
        const float2* p = wave == 0 ? sxy : dxy;
        double cx = 0, cy = 0;
        for (int i = 0; i < n; ++i) { cx += p[i].x; cy += p[i].y; }
        T[wave][1] = cx / n;
        T[wave][2] = cy / n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * n; i += 1024) {
        const int side = i >= n, k = side ? i - n : i;
        const float2 p = side ? dxy[k] : sxy[k];
        const double cx = T[side][1], cy = T[side][2];
        term[side][k] = sqrt((p.x - cx) * (p.x - cx) + (p.y - cy) * (p.y - cy));
    }
    __syncthreads();
    if (lane == 0 && wave < 2) {
        double md = 0;
        for (int i = 0; i < n; ++i) md += term[wave][i];
        md /= n;
        T[wave][0] = md > 1e-12 ? sqrt(2.0) / md : 1.0;
    }
    __syncthreads();
    const double ts0 = T[0][0], ts1 = T[0][1], ts2 = T[0][2], td0 = T[1][0], td1 = T[1][1], td2 = T[1][2];
    for (int i = threadIdx.x; i < n; i += 1024) {   // normalise_points()
        const double a = ts0 * (sxy[i].x - ts1), b = ts0 * (sxy[i].y - ts2);
        const double c = td0 * (dxy[i].x - td1), d = td0 * (dxy[i].y - td2);
        P[i] = a; P[n + i] = b; P[2 * n + i] = c; P[3 * n + i] = d;
        F[i] = (float)a; F[n + i] = (float)b; F[2 * n + i] = (float)c; F[3 * n + i] = (float)d;
    }
    if (threadIdx.x == 0) {
        const double tau = fmax(thr, HOMO_TAU_FLOOR) * td0;
        hdr->ts[0] = ts0; hdr->ts[1] = ts1; hdr->ts[2] = ts2;
        hdr->td[0] = td0; hdr->td[1] = td1; hdr->td[2] = td2;
        hdr->tau = tau;
        hdr->inv_tau2 = (float)(1.0 / (tau * tau));
        hdr->n_score = n;
    }
}

static int match_localize_separately(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, const int32_t* out4,
                                     double ratio, int min_good, double thr, uint64_t seed, int32_t* good_q, int32_t* good_t,
                                     float* good_d, int64_t* n_good, double* H_out, uint8_t* mask, int* found) {
    int rc = imatch_ratio_dev(ctx, out4, q->n, ratio, good_q, good_t, good_d, n_good);
    if (rc) return rc;
    const int n = (int)*n_good;
    if (n <= min_good || n < 4) return IMATCH_OK;
    std::vector<float> src(2 * (size_t)n), dst(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        src[2 * i] = q->kp[7 * (size_t)good_q[i]]; src[2 * i + 1] = q->kp[7 * (size_t)good_q[i] + 1];
        dst[2 * i] = t->kp[7 * (size_t)good_t[i]]; dst[2 * i + 1] = t->kp[7 * (size_t)good_t[i] + 1];
    }
    return imatch_homography(ctx, src.data(), dst.data(), n, thr, seed, H_out, mask, found);
}

extern "C" int imatch_match_localize(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, double ratio,
                                     int min_good, double thr, uint64_t seed, int32_t* good_q, int32_t* good_t, float* good_d,
                                     int64_t* n_good, double* H_out, uint8_t* mask, int* found) {
    if (!ctx || !q || !t || !n_good || !H_out || !found || (q->n > 0 && (!good_q || !good_t || !good_d)))
        return imatch_set_err(ctx, IMATCH_EINVAL, "null argument");
    if (!q->has_xy || !t->has_xy || (q->n > 0 && !q->kp) || (t->n > 0 && !t->kp))
        return imatch_set_err(ctx, IMATCH_EINVAL, "both descriptor sets must come from imatch_sift_extract* (they carry their "
                                                  "keypoints' coordinates); use imatch_knn2_dev / imatch_ratio_dev / imatch_homography");
    if (q->n > INT_MAX / 2) return imatch_set_err(ctx, IMATCH_ESIZE, "too many queries");
    *n_good = 0;
    *found = 0;
    if (q->n == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    static const int cap_env = getenv("IMATCH_TAIL_CAP") ? atoi(getenv("IMATCH_TAIL_CAP")) : TAIL_CAP;   // tests: force the overflow path
    const int cap = std::max(4, std::min(cap_env, TAIL_CAP));
    const int nq = (int)q->n;
    // device block: kNN result, header, lists, points
    const size_t o_out4 = 0, o_hdr = o_out4 + (((size_t)nq * 16 + 255) & ~(size_t)255), o_gq = o_hdr + 256,
                 o_gt = o_gq + 4 * (size_t)cap, o_gd = o_gt + 4 * (size_t)cap, o_P = (o_gd + 4 * (size_t)cap + 255) & ~(size_t)255,
                 o_F = o_P + 32 * (size_t)cap, total = o_F + 16 * (size_t)cap;
    struct Block {   // back to the pool on every way out
        imatch_ctx* c;
        char* p;
        ~Block() { if (p) imatch_pool_free(c, p); }
    } blk{ctx, (char*)imatch_pool_alloc(ctx, total)};
    if (!blk.p) return imatch_set_err(ctx, IMATCH_ENOMEM, "match block for %d queries", nq);
    int32_t* out4 = (int32_t*)(blk.p + o_out4);
    TailHdr* d_hdr = (TailHdr*)(blk.p + o_hdr);
    const size_t back = 256 + 12 * (size_t)cap;   // header + the three lists, contiguous
    int rc = imatch_hs_reserve(ctx, back);
    if (rc) return rc;
    rc = imatch_knn2_dev(ctx, q, t, 0, out4);
    if (rc) return rc;
    (void)imatch_stage_begin(ctx, IMATCH_STAGE_RATIO, ctx->stream);
    hipLaunchKernelGGL(match_tail_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const int4*)out4, nq, ratio, (const float2*)q->xy,
                       (const float2*)t->xy, cap, min_good, thr, (unsigned long long)seed, d_hdr, (int*)(blk.p + o_gq),
                       (int*)(blk.p + o_gt), (float*)(blk.p + o_gd), (double*)(blk.p + o_P), (float*)(blk.p + o_F));
    HIP_TRY(ctx, hipGetLastError());
    (void)imatch_stage_end(ctx, IMATCH_STAGE_RATIO, ctx->stream);
    HomoSeg hs;
    memset(&hs, 0, sizeof(hs));
    hs.seg.n = 1;
    hs.seeds = &d_hdr->seed;
    hs.inv_tau2 = &d_hdr->inv_tau2;
    (void)imatch_stage_begin(ctx, IMATCH_STAGE_HOMOGRAPHY, ctx->stream);
    hipLaunchKernelGGL(homography_score_kernel, dim3(HOMO_HYPOTHESES / 64, 1), dim3(64), 0, ctx->stream,
                       (const double*)(blk.p + o_P), (const float*)(blk.p + o_F), hs, &d_hdr->key, (const int*)&d_hdr->n_score);
    HIP_TRY(ctx, hipGetLastError());
    (void)imatch_stage_end(ctx, IMATCH_STAGE_HOMOGRAPHY, ctx->stream);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hs, d_hdr, back, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const TailHdr hdr = *(const TailHdr*)ctx->hs;
    const int n = hdr.n_good;
    if (n > cap)   // the lists did not fit the block: the per-frame calls, on the kNN result that is still here
        return match_localize_separately(ctx, q, t, out4, ratio, min_good, thr, seed, good_q, good_t, good_d, n_good, H_out, mask, found);
    memcpy(good_q, (char*)ctx->hs + 256, 4 * (size_t)n);
    memcpy(good_t, (char*)ctx->hs + 256 + 4 * (size_t)cap, 4 * (size_t)n);
    memcpy(good_d, (char*)ctx->hs + 256 + 8 * (size_t)cap, 4 * (size_t)n);
    *n_good = n;
    if (mask && n > 0) memset(mask, 0, (size_t)n);
    if (!hdr.scored) return IMATCH_OK;   // too few matches for a homography (position_finder.py:316)
    // the host's own transforms: the device's must be these, bit for bit
    std::vector<float> src(2 * (size_t)n), dst(2 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        if (good_q[i] < 0 || good_q[i] >= q->n || good_t[i] < 0 || good_t[i] >= t->n)
            return imatch_set_err(ctx, IMATCH_EHIP, "match %d of %d names rows %d / %d", i, n, good_q[i], good_t[i]);
        src[2 * i] = q->kp[7 * (size_t)good_q[i]]; src[2 * i + 1] = q->kp[7 * (size_t)good_q[i] + 1];
        dst[2 * i] = t->kp[7 * (size_t)good_t[i]]; dst[2 * i + 1] = t->kp[7 * (size_t)good_t[i] + 1];
    }
    double ts[3], td[3];
    hartley(src.data(), n, ts);
    hartley(dst.data(), n, td);
    const double tau = fmax(thr, HOMO_TAU_FLOOR) * td[0];
    static const int force_env = getenv("IMATCH_TAIL_MISMATCH") ? atoi(getenv("IMATCH_TAIL_MISMATCH")) : 0;   // tests: the mismatch path
    if (force_env || memcmp(ts, hdr.ts, sizeof(ts)) || memcmp(td, hdr.td, sizeof(td)) || memcmp(&tau, &hdr.tau, sizeof(tau))) {
        ctx->tail_mismatches++;
        return imatch_homography(ctx, src.data(), dst.data(), n, thr, seed, H_out, mask, found);
    }
    std::vector<double> P(4 * (size_t)n);
    normalise_points(src.data(), dst.data(), n, ts, td, P.data(), nullptr);
    finish_problem(P.data(), src.data(), dst.data(), n, ts, td, tau, hdr.key, seed, thr, H_out, mask, found);
    return IMATCH_OK;
}

extern "C" int imatch_match_localize_debug_mismatches(imatch_ctx* ctx) { return ctx ? ctx->tail_mismatches : 0; }

extern "C" int imatch_homography(imatch_ctx* ctx, const float* src, const float* dst, int n, double thr, uint64_t seed,
                                 double* H_out, uint8_t* mask, int* found) {
    if (!ctx || !found || !H_out || n < 0 || (n > 0 && (!src || !dst))) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    imatch_segtab seg;
    memset(&seg, 0, sizeof(seg));
    seg.n = 1;
    seg.off[1] = n;
    return homography_run(ctx, src, dst, seg, thr, &seed, H_out, mask, found);
}

extern "C" int imatch_homography_batch(imatch_ctx* ctx, const float* src, const float* dst, const int64_t* offsets,
                                       int n_seg, double thr, const uint64_t* seeds, double* H_out, uint8_t* mask,
                                       int* found) {
    if (!ctx || !found || !H_out || !offsets || !seeds || n_seg < 1 || n_seg > IMATCH_MAX_BATCH)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    imatch_segtab seg;
    memset(&seg, 0, sizeof(seg));
    seg.n = n_seg;
    for (int b = 0; b <= n_seg; ++b) {
        if (offsets[b] < 0 || offsets[b] > INT_MAX / 8 || (b > 0 && offsets[b] < offsets[b - 1]) || (b == 0 && offsets[0] != 0))
            return imatch_set_err(ctx, IMATCH_EINVAL, "segment offsets must start at 0 and ascend");
        seg.off[b] = (int)offsets[b];
    }
    if (seg.off[n_seg] > 0 && (!src || !dst)) return imatch_set_err(ctx, IMATCH_EINVAL, "null points");
    return homography_run(ctx, src, dst, seg, thr, seeds, H_out, mask, found);
}
