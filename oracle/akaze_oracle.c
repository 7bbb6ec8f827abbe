/*
 * akaze_oracle.c -- CPU restatement of the AKAZE path: TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Follows the detector the checked-in reference actually runs, match_finder.py:210-211,220:
 *   cv2.AKAZE_create(descriptor_type=cv2.AKAZE_DESCRIPTOR_KAZE, threshold=0.0005, nOctaves=4, nOctaveLayers=5,
 *                    diffusivity=cv2.KAZE_DIFF_PM_G2).detectAndCompute(img, None)
 * (SURVEY 8f rank 3). Output: keypoints (pt, size, angle, response, octave, class_id = evolution level) and 64 float
 * M-SURF descriptors of unit length.
 *
 * PARITY UNPINNED: the arithmetic is OpenCV's (features2d/src/kaze/AKAZEFeatures.cpp, nldiffusion_functions.cpp,
 * fed.cpp), absent here, and the reference holds no fixtures. Restated from the published 4.x sources ("Spec A"):
 *  A1 evolution: octave o < 4 (stops when the level is narrower than 80 or lower than 40), sublevel j < 5: size =
 *     (int)(W / 2^o) x (int)(H / 2^o), esigma = 1.6 * 2^(j/5 + o), sigma_size = cvRound(esigma * 1.5 / 2^o),
 *     etime = esigma^2 / 2, border = cvRound(12 sqrt(2) sigma_size) + 1; FED cycle per level from
 *     fed_tau_by_process_time(etime_i - etime_{i-1}, 1, 0.25, reordering).
 *  A2 scale space: img / 255 (f32); Lt_0 = GaussianBlur(9 x 9, sigma 1.6, BORDER_REPLICATE); contrast factor k = the
 *     70 % point of the 300-bin histogram of |grad| of GaussianBlur(img, 5 x 5, sigma 1) under the unnormalised Scharr
 *     pair (interior pixels, bin 0 excluded), x 0.75 per octave; per level: Lt copied (or halved by INTER_AREA at a
 *     new octave), Lsmooth = GaussianBlur(Lt, 5 x 5, 1), Scharr Lx, Ly, conductivity g2 = 1 / (1 + |grad|^2 / k^2),
 *     then the cycle's explicit steps Lt += tau/2 * sum over the 4 neighbours (g_c + g_n)(L_n - L_c), edges dropping
 *     the missing neighbour, corners 0.
 *  A3 detector response: Lx, Ly, Lxx, Lxy, Lyy by the 3-tap Scharr pair stretched to +- sigma_size (sepFilter2D,
 *     BORDER_REFLECT_101) from Lsmooth; Ldet = (Lxx Lyy - Lxy^2) sigma_size^4.
 *  A4 extrema: inside the border, Ldet > threshold and > its 8 neighbours; row-major scan, the FIRST kept point within
 *     sigma_size (window [-r, r) in both axes, L2 <= r) is replaced if weaker, else the candidate is dropped; then the
 *     level below (coordinates x ratio), then the level above (coordinates / ratio), each clearing the weaker point of
 *     the first neighbour found.
 *  A5 sub-pixel: 2-D quadratic fit of Ldet, |dx|, |dy| <= 1; pt = (x + dx) * ratio + (ratio - 1) / 2, size = 2 * 1.5 *
 *     esigma, response = Ldet, octave, class_id = level.
 *  A6 orientation: 109 samples of (Lx, Ly) within radius 6 at step scale = cvRound(size / 2 / ratio), Gaussian 2.5
 *     table weights, angles sorted into 42 slices, the 7-slice window with the largest summed vector.
 *  A7 descriptor (M-SURF 64): 4 x 4 subregions of 9 x 9 samples at step `scale` on the rotated 24-scale grid, bilinear
 *     Lx, Ly, Gaussian 2.5 scale weight per sample and 1.5 per subregion, (sum dx, sum dy, sum |dx|, sum |dy|), unit norm.
 * Points where this restatement chose without being able to check (no OpenCV here): the cross-level passes search with the
 * scanning level's sigma_size (not multiplied by the octave ratio); KeyPoint.angle is kept in degrees and converted in the
 * descriptor; the 2 x 2 box of the octave step adds in the vector path's order ((a + b) + (c + d)); Lx / Ly reach the
 * orientation and the descriptor unscaled (a per-level factor would cancel in both).
 * Where OpenCV's float order depends on its build (SIMD width, FMA) one order is fixed here: filters are ascending-tap
 * fmaf chains (as sift_oracle.c), everything else is the source's expression order without contraction. fastAtan2 and
 * OpenCV's exp are libm's here. A new octave is resize(INTER_AREA) of the last level (oracle_resize_area_f32).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int oracle_resize_area_f32(const float* src, int rows, int cols, float* dst, int drows, int dcols);

#define AK_MAX_LEVELS 32
#define AK_MAX_STEPS 256

typedef struct {
    int rows, cols, octave, sublevel, sigma_size, border, nsteps;
    float esigma, etime, ratio;
    float tau[AK_MAX_STEPS];
    float *Lt, *Lsmooth, *Lx, *Ly, *Ldet;
} ak_level;

typedef struct {
    int n_levels, n_kp;
    float kcontrast;
    ak_level lv[AK_MAX_LEVELS];
    float* kp;    /* [n_kp][7]: x, y, size, angle, response, octave, class_id */
    float* desc;  /* [n_kp][64] */
} oracle_akaze;

static int cv_round_f(double v) { return (int)lrint(v); }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

/* ---- fed.cpp ---------------------------------------------------------------------------------------------------- */
static int is_prime(int n) {
    if (n <= 1) return 0;
    if (n == 2 || n == 3 || n == 5 || n == 7) return 1;
    if (n % 2 == 0 || n % 3 == 0 || n % 5 == 0 || n % 7 == 0) return 0;
    const int upper = (int)(sqrt((double)n + 1.0));
    for (int d = 11; d <= upper; d += 2)
        if (n % d == 0) return 0;
    return 1;
}

/* fed_tau_by_process_time(T, 1, tau_max, reordering = true, tau) */
int oracle_fed_tau(float T, float tau_max, float* tau /* >= AK_MAX_STEPS */) {
    const int n = (int)ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f);
    if (n <= 0 || n > AK_MAX_STEPS) return n <= 0 ? 0 : -1;
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

/* ---- filters ---------------------------------------------------------------------------------------------------- */
/* getGaussianKernel(n, sigma, CV_32F), bit-exact form as in sift_oracle.c */
static void gauss_kernel_n(int n, double sigma, float* k) {
    const int n2 = (n - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double vals[32], sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        vals[i] = exp((double)(x * x) * scale2x);
        sum += vals[i];
    }
    sum = sum * 2 + 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) k[i] = k[n - 1 - i] = (float)(vals[i] * mul1);
    k[n2] = (float)mul1;
}

/* separable filter with taps at offsets off[t] (ascending), row pass with (kx, offx) then column pass with (ky, offy);
 * border: 0 = replicate, 1 = reflect101. s = kx[0] * v0, then s = fmaf(k[t], v_t, s). */
static void sep_filter(const float* src, int rows, int cols, const float* kx, const int* offx, int nx, const float* ky,
                       const int* offy, int ny, int border, float* dst) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const float* s = src + (size_t)y * cols;
        float* d = tmp + (size_t)y * cols;
        for (int x = 0; x < cols; ++x) {
            float acc = 0.f;
            for (int t = 0; t < nx; ++t) {
                const int xx = border ? reflect101(x + offx[t], cols) : clampi(x + offx[t], 0, cols - 1);
                acc = t ? fmaf(kx[t], s[xx], acc) : kx[t] * s[xx];
            }
            d[x] = acc;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        float* d = dst + (size_t)y * cols;
        for (int t = 0; t < ny; ++t) {
            const int yy = border ? reflect101(y + offy[t], rows) : clampi(y + offy[t], 0, rows - 1);
            const float* s = tmp + (size_t)yy * cols;
            const float kt = ky[t];
            if (t == 0) for (int x = 0; x < cols; ++x) d[x] = kt * s[x];
            else for (int x = 0; x < cols; ++x) d[x] = fmaf(kt, s[x], d[x]);
        }
    }
    free(tmp);
}

/* GaussianBlur(src, dst, Size(n, n), sigma, sigma, BORDER_REPLICATE) */
void oracle_akaze_gauss(const float* src, int rows, int cols, int n, double sigma, float* dst) {
    float k[32];
    int off[32];
    gauss_kernel_n(n, sigma, k);
    for (int i = 0; i < n; ++i) off[i] = i - n / 2;
    sep_filter(src, rows, cols, k, off, n, k, off, n, 0, dst);
}

/* Scharr(src, dst, CV_32F, dx, dy, 1, 0, BORDER_DEFAULT): unnormalised kernels [-1 0 1] and [3 10 3] */
static void scharr(const float* src, int rows, int cols, int xorder, float* dst) {
    static const float kd[2] = {-1.f, 1.f}, ks[3] = {3.f, 10.f, 3.f};
    static const int od[2] = {-1, 1}, os[3] = {-1, 0, 1};
    if (xorder) sep_filter(src, rows, cols, kd, od, 2, ks, os, 3, 1, dst);
    else sep_filter(src, rows, cols, ks, os, 3, kd, od, 2, 1, dst);
}

/* compute_derivative_kernels + sepFilter2D: the Scharr pair stretched to +- scale; order 1 along x if xorder */
static void scaled_deriv(const float* src, int rows, int cols, int xorder, int scale, float* dst) {
    const float w = 10.0f / 3.0f, norm = 1.0f / (2.0f * scale * (w + 2.0f));
    const float kd[2] = {-1.f, 1.f}, ks[3] = {norm, w * norm, norm};
    const int od[2] = {-scale, scale}, os[3] = {-scale, 0, scale};
    if (xorder) sep_filter(src, rows, cols, kd, od, 2, ks, os, 3, 1, dst);
    else sep_filter(src, rows, cols, ks, os, 3, kd, od, 2, 1, dst);
}

/* compute_kcontrast(Lx, Ly, 0.7, 300) */
static float kcontrast(const float* Lx, const float* Ly, int rows, int cols, float perc, int nbins) {
    const int total = (rows - 2) * (cols - 2);
    if (total <= 0) return 0.03f;
    float* modg = (float*)malloc(sizeof(float) * (size_t)total);
    float hmax = 0.f;
    int n = 0;
    for (int i = 1; i < rows - 1; ++i)
        for (int j = 1; j < cols - 1; ++j) {
            const float lx = Lx[(size_t)i * cols + j], ly = Ly[(size_t)i * cols + j];
            const float d = sqrtf(lx * lx + ly * ly);
            modg[n++] = d;
            if (d > hmax) hmax = d;
        }
    if (hmax == 0.f) { free(modg); return 0.03f; }
    int* hist = (int*)calloc((size_t)nbins, sizeof(int));
    const float mul = (float)(nbins - 1) / hmax;    /* Mat *= double scalar on CV_32F: the product is formed in f32 */
    for (int i = 0; i < total; ++i) hist[(int)(modg[i] * mul)]++;
    const int nthreshold = (int)((total - hist[0]) * perc);
    int nelements = 0;
    float k = 0.03f;
    for (int b = 1; b < nbins; ++b) {
        if (nelements >= nthreshold) { k = (float)hmax * b / nbins; break; }
        nelements += hist[b];
    }
    free(hist);
    free(modg);
    return k;
}

/* nld_step_scalar: Lstep = step_size * divergence, then Lt += Lstep */
static void nld_step(float* Lt, const float* Lf, int rows, int cols, float step_size, float* Lstep) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const float* tc = Lt + (size_t)y * cols; const float* fc = Lf + (size_t)y * cols;
        const float* ta = y > 0 ? tc - cols : NULL; const float* fa = y > 0 ? fc - cols : NULL;
        const float* tb = y + 1 < rows ? tc + cols : NULL; const float* fb = y + 1 < rows ? fc + cols : NULL;
        float* d = Lstep + (size_t)y * cols;
        for (int x = 0; x < cols; ++x) {
            const int edge_x = (x == 0 || x == cols - 1), edge_y = (y == 0 || y == rows - 1);
            if (edge_x && edge_y) { d[x] = 0.f; continue; }
            float s;
            if (!edge_x && !edge_y) {
                s = (fc[x] + fc[x + 1]) * (tc[x + 1] - tc[x]) + (fc[x] + fc[x - 1]) * (tc[x - 1] - tc[x]) +
                    (fc[x] + fb[x]) * (tb[x] - tc[x]) + (fc[x] + fa[x]) * (ta[x] - tc[x]);
            } else if (edge_y) {            /* top or bottom row: right, left, then the one vertical neighbour */
                const float* tv = y == 0 ? tb : ta; const float* fv = y == 0 ? fb : fa;
                s = (fc[x] + fc[x + 1]) * (tc[x + 1] - tc[x]) + (fc[x] + fc[x - 1]) * (tc[x - 1] - tc[x]) +
                    (fc[x] + fv[x]) * (tv[x] - tc[x]);
            } else {                        /* left or right column: the one horizontal neighbour, below, above */
                const int xn = x == 0 ? 1 : x - 1;
                s = (fc[x] + fc[xn]) * (tc[xn] - tc[x]) + (fc[x] + fb[x]) * (tb[x] - tc[x]) + (fc[x] + fa[x]) * (ta[x] - tc[x]);
            }
            d[x] = s * step_size;
        }
    }
    for (size_t i = 0; i < (size_t)rows * cols; ++i) Lt[i] = Lt[i] + Lstep[i];
}

/* ---- A6 / A7 ------------------------------------------------------------------------------------------------------ */
static float gauss25(int i, int j) {   /* OpenSURF's 7 x 7 table: 1 / (2 pi 2.5^2) exp(-(i^2 + j^2) / (2 2.5^2)) to 8 decimals */
    const double v = exp(-(double)(i * i + j * j) / 12.5) / (2.0 * M_PI * 6.25);
    return (float)(floor(v * 1e8 + 0.5) / 1e8);
}

static float angle_deg(float y, float x) {   /* fastAtan2: degrees in [0, 360) */
    float a = atan2f(y, x) * (float)(180.0 / M_PI);
    if (a < 0) a += 360.f;
    if (a >= 360.f) a -= 360.f;
    return a;
}

static float main_orientation(const ak_level* e, float ptx, float pty, float size) {
    const int scale = cv_round_f(0.5f * size / e->ratio);
    const int x0 = cv_round_f(ptx / e->ratio), y0 = cv_round_f(pty / e->ratio);
    float resX[109], resY[109], Ang[109];
    int k = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j < 36) {
                const int y = clampi(y0 + j * scale, 0, e->rows - 1), x = clampi(x0 + i * scale, 0, e->cols - 1);
                const float w = gauss25(abs(i), abs(j));
                resX[k] = w * e->Lx[(size_t)y * e->cols + x];
                resY[k] = w * e->Ly[(size_t)y * e->cols + x];
                Ang[k] = angle_deg(resY[k], resX[k]) * (float)(M_PI / 180.0);
                ++k;
            }
    enum { SL = 42, WIN = 7 };
    const float ang_step = (float)(2.0 * M_PI / SL);
    int slice_n[SL] = {0}, slice[SL][109];
    for (int i = 0; i < 109; ++i) {
        int b = (int)(Ang[i] / ang_step);
        if (b < 0 || b >= SL) b = 0;
        slice[b][slice_n[b]++] = i;
    }
    float maxX = 0.f, maxY = 0.f;
    for (int i = 0; i < WIN; ++i)
        for (int j = 0; j < slice_n[i]; ++j) { maxX += resX[slice[i][j]]; maxY += resY[slice[i][j]]; }
    float maxNorm = maxX * maxX + maxY * maxY;
    for (int sn = 1; sn <= SL - WIN; ++sn) {
        if (slice_n[sn - 1] == 0 && slice_n[sn + WIN - 1] == 0) continue;
        float sx = 0.f, sy = 0.f;
        for (int i = sn; i < sn + WIN; ++i)
            for (int j = 0; j < slice_n[i]; ++j) { sx += resX[slice[i][j]]; sy += resY[slice[i][j]]; }
        const float norm = sx * sx + sy * sy;
        if (norm > maxNorm) { maxNorm = norm; maxX = sx; maxY = sy; }
    }
    for (int sn = SL - WIN + 1; sn < SL; ++sn) {
        const int remain = sn + WIN - SL;
        if (slice_n[sn - 1] == 0 && slice_n[remain - 1] == 0) continue;
        float sx = 0.f, sy = 0.f;
        for (int i = sn; i < SL; ++i)
            for (int j = 0; j < slice_n[i]; ++j) { sx += resX[slice[i][j]]; sy += resY[slice[i][j]]; }
        for (int i = 0; i < remain; ++i)
            for (int j = 0; j < slice_n[i]; ++j) { sx += resX[slice[i][j]]; sy += resY[slice[i][j]]; }
        const float norm = sx * sx + sy * sy;
        if (norm > maxNorm) { maxNorm = norm; maxX = sx; maxY = sy; }
    }
    return angle_deg(maxY, maxX);
}

static float gaussian(float x, float y, float sigma) { return expf(-(x * x + y * y) / (2.0f * sigma * sigma)); }

static void msurf64(const ak_level* e, float ptx, float pty, float size, float angle_d, int octave, float* desc) {
    const int sample_step = 5, pattern_size = 12;
    const float ratio = (float)(1 << octave);
    const int scale = cv_round_f(0.5f * size / ratio);
    const float angle = angle_d * (float)(M_PI / 180.f);
    const float yf = pty / ratio, xf = ptx / ratio, co = cosf(angle), si = sinf(angle);
    float len = 0.f, cx = -0.5f, cy = 0.5f;
    int dcount = 0, i = -8;
    while (i < pattern_size) {
        int j = -8;
        i = i - 4;
        cx += 1.0f;
        cy = -0.5f;
        while (j < pattern_size) {
            float dx = 0.f, dy = 0.f, mdx = 0.f, mdy = 0.f;
            cy += 1.0f;
            j = j - 4;
            const int ky = i + sample_step, kx = j + sample_step;
            const float xs = xf + (-kx * scale * si + ky * scale * co);
            const float ys = yf + (kx * scale * co + ky * scale * si);
            for (int k = i; k < i + 9; ++k)
                for (int l = j; l < j + 9; ++l) {
                    const float sample_y = yf + (l * scale * co + k * scale * si);
                    const float sample_x = xf + (-l * scale * si + k * scale * co);
                    const float gauss_s1 = gaussian(xs - sample_x, ys - sample_y, 2.5f * scale);
                    const int y1 = (int)floorf(sample_y), x1 = (int)floorf(sample_x);
                    const int y2 = y1 + 1, x2 = x1 + 1;
                    if (x1 < 0 || y1 < 0 || x2 >= e->cols || y2 >= e->rows) continue;
                    const float fx = sample_x - x1, fy = sample_y - y1;
                    const float* Lx = e->Lx; const float* Ly = e->Ly;
                    const size_t a = (size_t)y1 * e->cols + x1, b = (size_t)y1 * e->cols + x2, c = (size_t)y2 * e->cols + x1,
                                 d = (size_t)y2 * e->cols + x2;
                    const float rx = (1.0f - fx) * (1.0f - fy) * Lx[a] + fx * (1.0f - fy) * Lx[b] + (1.0f - fx) * fy * Lx[c] +
                                     fx * fy * Lx[d];
                    const float ry = (1.0f - fx) * (1.0f - fy) * Ly[a] + fx * (1.0f - fy) * Ly[b] + (1.0f - fx) * fy * Ly[c] +
                                     fx * fy * Ly[d];
                    const float rry = gauss_s1 * (rx * co + ry * si);
                    const float rrx = gauss_s1 * (-rx * si + ry * co);
                    dx += rrx; dy += rry; mdx += fabsf(rrx); mdy += fabsf(rry);
                }
            const float gauss_s2 = gaussian(cx - 2.0f, cy - 2.0f, 1.5f);
            desc[dcount++] = dx * gauss_s2;
            desc[dcount++] = dy * gauss_s2;
            desc[dcount++] = mdx * gauss_s2;
            desc[dcount++] = mdy * gauss_s2;
            len += (dx * dx + dy * dy + mdx * mdx + mdy * mdy) * gauss_s2 * gauss_s2;
            j += 9;
        }
        i += 9;
    }
    len = sqrtf(len);
    const float len_inv = 1.0f / len;
    for (int t = 0; t < 64; ++t) desc[t] *= len_inv;
}

/* ---- A4 ----------------------------------------------------------------------------------------------------------- */
static int find_neighbor_point(int x, int y, const uint8_t* mask, int rows, int cols, int r, int* idx) {
    for (int i = y - r; i < y + r; ++i) {
        if (i < 0 || i >= rows) continue;
        for (int j = x - r; j < x + r; ++j) {
            if (j < 0 || j >= cols || mask[(size_t)i * cols + j] == 0) continue;
            const int dx = j - x, dy = i - y;
            if (dx * dx + dy * dy <= r * r) { *idx = i * cols + j; return 1; }
        }
    }
    return 0;
}

/* ---- driver ------------------------------------------------------------------------------------------------------- */
void oracle_akaze_free(oracle_akaze* A) {
    if (!A) return;
    for (int i = 0; i < A->n_levels; ++i) {
        free(A->lv[i].Lt); free(A->lv[i].Lsmooth); free(A->lv[i].Lx); free(A->lv[i].Ly); free(A->lv[i].Ldet);
    }
    free(A->kp); free(A->desc); free(A);
}

/* AKAZE_create(DESCRIPTOR_KAZE, 0, 3, threshold, omax, nsublevels, DIFF_PM_G2).detectAndCompute(image, None);
 * NULL for bad arguments. */
oracle_akaze* oracle_akaze_run(const uint8_t* image, int rows, int cols, double threshold, int omax, int nsublevels) {
    const float soffset = 1.6f, derivative_factor = 1.5f, smax = 12.0f * sqrtf(2.0f);
    if (rows < 3 || cols < 3 || omax < 1 || nsublevels < 1 || omax * nsublevels > AK_MAX_LEVELS) return NULL;
    oracle_akaze* A = (oracle_akaze*)calloc(1, sizeof *A);
    /* A1 */
    for (int i = 0, power = 1; i <= omax - 1; ++i, power *= 2) {
        const float rfactor = 1.0f / power;
        const int lh = (int)(rows * rfactor), lw = (int)(cols * rfactor);
        if ((lw < 80 || lh < 40) && i != 0) break;
        for (int j = 0; j < nsublevels; ++j) {
            ak_level* e = &A->lv[A->n_levels++];
            e->rows = lh; e->cols = lw;
            e->esigma = soffset * powf(2.f, (float)j / (float)nsublevels + i);
            e->sigma_size = cv_round_f(e->esigma * derivative_factor / power);
            e->etime = 0.5f * (e->esigma * e->esigma);
            e->octave = i; e->sublevel = j; e->ratio = (float)power;
            e->border = cv_round_f(smax * e->sigma_size) + 1;
        }
    }
    for (int i = 1; i < A->n_levels; ++i) {
        A->lv[i].nsteps = oracle_fed_tau(A->lv[i].etime - A->lv[i - 1].etime, 0.25f, A->lv[i].tau);
        if (A->lv[i].nsteps < 0) { oracle_akaze_free(A); return NULL; }
    }
    /* A2 */
    const size_t px0 = (size_t)rows * cols;
    float* img = (float*)malloc(sizeof(float) * px0);
    for (size_t i = 0; i < px0; ++i) img[i] = (float)image[i] * (float)(1.0 / 255.0);   /* convertTo(CV_32F, 1/255): f32 product */
    for (int i = 0; i < A->n_levels; ++i) {
        ak_level* e = &A->lv[i];
        const size_t px = (size_t)e->rows * e->cols;
        e->Lt = (float*)malloc(sizeof(float) * px); e->Lsmooth = (float*)malloc(sizeof(float) * px);
        e->Lx = (float*)malloc(sizeof(float) * px); e->Ly = (float*)malloc(sizeof(float) * px);
        e->Ldet = (float*)malloc(sizeof(float) * px);
    }
    oracle_akaze_gauss(img, rows, cols, 9, soffset, A->lv[0].Lsmooth);   /* getGaussianKernelSize(1.6) = 9 */
    memcpy(A->lv[0].Lt, A->lv[0].Lsmooth, sizeof(float) * px0);
    if (A->n_levels > 1) {
        float* sm = (float*)malloc(sizeof(float) * px0);
        float* gx = (float*)malloc(sizeof(float) * px0);
        float* gy = (float*)malloc(sizeof(float) * px0);
        oracle_akaze_gauss(img, rows, cols, 5, 1.0, sm);
        scharr(sm, rows, cols, 1, gx);
        scharr(sm, rows, cols, 0, gy);
        float kc = kcontrast(gx, gy, rows, cols, 0.7f, 300);
        A->kcontrast = kc;
        for (int i = 1; i < A->n_levels; ++i) {
            ak_level* e = &A->lv[i];
            const ak_level* p = &A->lv[i - 1];
            const size_t px = (size_t)e->rows * e->cols;
            if (e->octave > p->octave) {
                oracle_resize_area_f32(p->Lt, p->rows, p->cols, e->Lt, e->rows, e->cols);   /* sift_oracle.c */
                kc *= 0.75f;
            } else {
                memcpy(e->Lt, p->Lt, sizeof(float) * px);
            }
            oracle_akaze_gauss(e->Lt, e->rows, e->cols, 5, 1.0, e->Lsmooth);
            scharr(e->Lsmooth, e->rows, e->cols, 1, gx);
            scharr(e->Lsmooth, e->rows, e->cols, 0, gy);
            const float k2inv = 1.0f / (kc * kc);
            for (size_t t = 0; t < px; ++t) sm[t] = 1.0f / (1.0f + ((gx[t] * gx[t] + gy[t] * gy[t]) * k2inv));   /* pm_g2 */
            for (int s = 0; s < e->nsteps; ++s) nld_step(e->Lt, sm, e->rows, e->cols, e->tau[s] * 0.5f, gx);
        }
        free(sm); free(gx); free(gy);
    }
    free(img);
    /* A3 */
    for (int i = 0; i < A->n_levels; ++i) {
        ak_level* e = &A->lv[i];
        const size_t px = (size_t)e->rows * e->cols;
        float* Lxx = (float*)malloc(sizeof(float) * px);
        float* Lxy = (float*)malloc(sizeof(float) * px);
        float* Lyy = (float*)malloc(sizeof(float) * px);
        scaled_deriv(e->Lsmooth, e->rows, e->cols, 1, e->sigma_size, e->Lx);
        scaled_deriv(e->Lx, e->rows, e->cols, 1, e->sigma_size, Lxx);
        scaled_deriv(e->Lx, e->rows, e->cols, 0, e->sigma_size, Lxy);
        scaled_deriv(e->Lsmooth, e->rows, e->cols, 0, e->sigma_size, e->Ly);
        scaled_deriv(e->Ly, e->rows, e->cols, 0, e->sigma_size, Lyy);
        const float s4 = (float)(e->sigma_size * e->sigma_size * e->sigma_size * e->sigma_size);
        for (size_t t = 0; t < px; ++t) e->Ldet[t] = (Lxx[t] * Lyy[t] - Lxy[t] * Lxy[t]) * s4;
        free(Lxx); free(Lxy); free(Lyy);
    }
    /* A4 */
    uint8_t* mask[AK_MAX_LEVELS];
    const float dthr = (float)threshold;
    for (int i = 0; i < A->n_levels; ++i) {
        const ak_level* e = &A->lv[i];
        mask[i] = (uint8_t*)calloc((size_t)e->rows * e->cols, 1);
        if (e->border + 1 >= e->rows) continue;
        for (int y = e->border; y < e->rows - e->border; ++y)
            for (int x = e->border; x < e->cols - e->border; ++x) {
                const float* c = e->Ldet + (size_t)y * e->cols + x;
                const float v = *c;
                if (v <= dthr) continue;
                if (v <= c[-1] || v <= c[1]) continue;
                if (v <= c[-e->cols - 1] || v <= c[-e->cols] || v <= c[-e->cols + 1]) continue;
                if (v <= c[e->cols - 1] || v <= c[e->cols] || v <= c[e->cols + 1]) continue;
                int idx = 0;
                if (find_neighbor_point(x, y, mask[i], e->rows, e->cols, e->sigma_size, &idx)) {
                    if (v > e->Ldet[idx]) mask[i][idx] = 0;
                    else continue;
                }
                mask[i][(size_t)y * e->cols + x] = 1;
            }
    }
    for (int i = 1; i < A->n_levels; ++i) {   /* against the level below */
        const ak_level* e = &A->lv[i]; const ak_level* p = &A->lv[i - 1];
        const int diff_ratio = (int)e->ratio / (int)p->ratio;
        for (int y = 0; y < e->rows; ++y)
            for (int x = 0; x < e->cols; ++x)
                if (mask[i][(size_t)y * e->cols + x]) {
                    int idx = 0;
                    if (find_neighbor_point(x * diff_ratio, y * diff_ratio, mask[i - 1], p->rows, p->cols, e->sigma_size, &idx))
                        if (e->Ldet[(size_t)y * e->cols + x] > p->Ldet[idx]) mask[i - 1][idx] = 0;
                }
    }
    for (int i = A->n_levels - 2; i >= 0; --i) {   /* against the level above */
        const ak_level* e = &A->lv[i]; const ak_level* n = &A->lv[i + 1];
        const int diff_ratio = (int)n->ratio / (int)e->ratio;
        for (int y = 0; y < e->rows; ++y)
            for (int x = 0; x < e->cols; ++x)
                if (mask[i][(size_t)y * e->cols + x]) {
                    int idx = 0;
                    if (find_neighbor_point(x / diff_ratio, y / diff_ratio, mask[i + 1], n->rows, n->cols, e->sigma_size, &idx))
                        if (e->Ldet[(size_t)y * e->cols + x] > n->Ldet[idx]) mask[i + 1][idx] = 0;
                }
    }
    /* A5 */
    int cap = 1024;
    A->kp = (float*)malloc(sizeof(float) * 7 * cap);
    for (int i = 0; i < A->n_levels; ++i) {
        const ak_level* e = &A->lv[i];
        const int cols_ = e->cols;
        const float* ldet = e->Ldet;
        for (int y = 0; y < e->rows; ++y)
            for (int x = 0; x < cols_; ++x) {
                if (!mask[i][(size_t)y * cols_ + x]) continue;
                const float Dx = 0.5f * (ldet[y * cols_ + x + 1] - ldet[y * cols_ + x - 1]);
                const float Dy = 0.5f * (ldet[(y + 1) * cols_ + x] - ldet[(y - 1) * cols_ + x]);
                const float Dxx = ldet[y * cols_ + x + 1] + ldet[y * cols_ + x - 1] - 2.0f * ldet[y * cols_ + x];
                const float Dyy = ldet[(y + 1) * cols_ + x] + ldet[(y - 1) * cols_ + x] - 2.0f * ldet[y * cols_ + x];
                const float Dxy = 0.25f * (ldet[(y + 1) * cols_ + x + 1] + ldet[(y - 1) * cols_ + x - 1] -
                                           ldet[(y - 1) * cols_ + x + 1] - ldet[(y + 1) * cols_ + x - 1]);
                /* cv::solve, 2 x 2 CV_32F, DECOMP_LU: closed form in double */
                float dx = 0.f, dy = 0.f;
                double det = (double)Dxx * Dyy - (double)Dxy * Dxy;
                if (det != 0) {
                    det = 1. / det;
                    dx = (float)(((double)(-Dx) * Dyy - (double)(-Dy) * Dxy) * det);
                    dy = (float)(((double)(-Dy) * Dxx - (double)(-Dx) * Dxy) * det);
                }
                if (fabsf(dx) > 1.0f || fabsf(dy) > 1.0f) continue;
                if (A->n_kp == cap) { cap *= 2; A->kp = (float*)realloc(A->kp, sizeof(float) * 7 * cap); }
                float* k = A->kp + 7 * (size_t)A->n_kp++;
                k[0] = (x + dx) * e->ratio + .5f * (e->ratio - 1.0f);
                k[1] = (y + dy) * e->ratio + .5f * (e->ratio - 1.0f);
                k[2] = 2.0f * (e->esigma * derivative_factor);
                k[3] = 0.f;
                k[4] = ldet[y * cols_ + x];
                k[5] = (float)e->octave;
                k[6] = (float)i;
            }
    }
    for (int i = 0; i < A->n_levels; ++i) free(mask[i]);
    /* A6, A7 */
    A->desc = (float*)malloc(sizeof(float) * 64 * (size_t)(A->n_kp ? A->n_kp : 1));
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < A->n_kp; ++t) {
        float* k = A->kp + 7 * (size_t)t;
        const ak_level* e = &A->lv[(int)k[6]];
        k[3] = main_orientation(e, k[0], k[1], k[2]);
        msurf64(e, k[0], k[1], k[2], k[3], (int)k[5], A->desc + 64 * (size_t)t);
    }
    return A;
}

int oracle_akaze_count(const oracle_akaze* A) { return A->n_kp; }
int oracle_akaze_levels(const oracle_akaze* A) { return A->n_levels; }
float oracle_akaze_kcontrast(const oracle_akaze* A) { return A->kcontrast; }
void oracle_akaze_level_info(const oracle_akaze* A, int i, int* rows, int* cols, int* sigma_size, int* border, int* nsteps,
                             float* esigma) {
    const ak_level* e = &A->lv[i];
    *rows = e->rows; *cols = e->cols; *sigma_size = e->sigma_size; *border = e->border; *nsteps = e->nsteps; *esigma = e->esigma;
}
void oracle_akaze_get(const oracle_akaze* A, float* kp, float* desc) {
    if (kp) memcpy(kp, A->kp, sizeof(float) * 7 * (size_t)A->n_kp);
    if (desc) memcpy(desc, A->desc, sizeof(float) * 64 * (size_t)A->n_kp);
}
/* kind: 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet */
int oracle_akaze_layer(const oracle_akaze* A, int kind, int i, float* out) {
    if (i < 0 || i >= A->n_levels || kind < 0 || kind > 4) return -1;
    const ak_level* e = &A->lv[i];
    const float* src = kind == 0 ? e->Lt : kind == 1 ? e->Lsmooth : kind == 2 ? e->Lx : kind == 3 ? e->Ly : e->Ldet;
    memcpy(out, src, sizeof(float) * (size_t)e->rows * e->cols);
    return 0;
}
