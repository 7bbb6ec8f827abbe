/*
 * TEST INFRASTRUCTURE -- CPU oracle for CLAHE, INTER_AREA resize and SIFT detectAndCompute.
 * Never imported by the product path; used only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference has no tests or golden vectors and its arithmetic
 * is OpenCV's (cv2, unpinned version >= 4.5, absent from this image). This file restates the
 * published OpenCV 4.x algorithms (modules/features2d/src/sift.dispatch.cpp + sift.simd.hpp,
 * modules/imgproc/src/clahe.cpp, resize.cpp, smooth.dispatch.cpp) for the calls the reference
 * makes:
 *   cv2.createCLAHE(clipLimit=30.0, tileGridSize=(8,8)).apply   match_finder.py:200-203
 *   cv2.resize(img, dim, interpolation=cv2.INTER_AREA)          utils.py:17-22
 *   SIFT_create(nfeatures=0, nOctaveLayers, contrastThreshold, edgeThreshold, sigma)
 *       .detectAndCompute(img, None)                            match_finder.py:205-209,220
 * Where OpenCV's float operation order depends on its build (SIMD width, FMA contraction) this
 * file fixes one order and says so; the HIP kernels follow the same order, so oracle-vs-HIP
 * differences come only from exp/atan2/sin/cos and from parallel summation order.
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline int cv_round(float v) { return (int)lrintf(v); } /* cvRound: round half to even */
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor(float v) { return (int)floorf(v); }
static inline uint8_t sat_u8(float v) {
    int i = cv_round(v);
    return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* ------------------------------------------------------------------------------------------
 * Spec C: CLAHE (clahe.cpp CLAHE_Impl::apply, CLAHE_CalcLut_Body, CLAHE_Interpolation_Body);
 * reference call match_finder.py:200,203.
 * ---------------------------------------------------------------------------------------- */
void oracle_clahe(const uint8_t* src, int rows, int cols, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst) {
    int erows = rows, ecols = cols;
    const uint8_t* ext = src;
    uint8_t* ext_buf = NULL;
    if (cols % tiles_x != 0 || rows % tiles_y != 0) {
        /* copyMakeBorder(src, 0, tilesY - rows % tilesY, 0, tilesX - cols % tilesX, REFLECT_101) */
        erows = rows + (tiles_y - rows % tiles_y);
        ecols = cols + (tiles_x - cols % tiles_x);
        ext_buf = (uint8_t*)malloc((size_t)erows * ecols);
        for (int y = 0; y < erows; ++y)
            for (int x = 0; x < ecols; ++x)
                ext_buf[(size_t)y * ecols + x] = src[(size_t)reflect101(y, rows) * cols + reflect101(x, cols)];
        ext = ext_buf;
    }
    const int tw = ecols / tiles_x, th = erows / tiles_y;
    const int tile_total = tw * th;
    const float lut_scale = (float)(255) / tile_total;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / 256);
        if (clip < 1) clip = 1;
    }
    uint8_t* lut = (uint8_t*)malloc((size_t)tiles_x * tiles_y * 256);
    for (int k = 0; k < tiles_x * tiles_y; ++k) {
        const int ty = k / tiles_x, tx = k % tiles_x;
        int hist[256];
        memset(hist, 0, sizeof(hist));
        for (int y = 0; y < th; ++y) {
            const uint8_t* p = ext + (size_t)(ty * th + y) * ecols + tx * tw;
            for (int x = 0; x < tw; ++x) hist[p[x]]++;
        }
        if (clip > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; ++i)
                if (hist[i] > clip) {
                    clipped += hist[i] - clip;
                    hist[i] = clip;
                }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) hist[i] += batch;
            if (residual != 0) {
                int step = 256 / residual;
                if (step < 1) step = 1;
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        for (int i = 0; i < 256; ++i) {
            sum += hist[i];
            lut[(size_t)k * 256 + i] = sat_u8((float)sum * lut_scale);
        }
    }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < rows; ++y) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tiles_y - 1) ty2 = tiles_y - 1;
        for (int x = 0; x < cols; ++x) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = cv_floor(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tiles_x - 1) tx2 = tiles_x - 1;
            const int v = src[(size_t)y * cols + x];
            const float l11 = lut[(size_t)(ty1 * tiles_x + tx1) * 256 + v];
            const float l12 = lut[(size_t)(ty1 * tiles_x + tx2) * 256 + v];
            const float l21 = lut[(size_t)(ty2 * tiles_x + tx1) * 256 + v];
            const float l22 = lut[(size_t)(ty2 * tiles_x + tx2) * 256 + v];
            /* order fixed here: products and sums unfused, as written in clahe.cpp */
            const float res = (l11 * xa1 + l12 * xa) * ya1 + (l21 * xa1 + l22 * xa) * ya;
            dst[(size_t)y * cols + x] = sat_u8(res);
        }
    }
    free(lut);
    free(ext_buf);
}

/* ------------------------------------------------------------------------------------------
 * Spec R: cv2.resize(..., INTER_AREA) for 8-bit single channel when shrinking (resize.cpp:
 * computeResizeAreaTab + ResizeArea_Invoker; ResizeAreaFast for integer factors). Reference call
 * utils.py:17-22 (dsize computed by the caller).
 * ---------------------------------------------------------------------------------------- */
typedef struct { int si, di; float alpha; } dec_alpha;

static int area_tab(int ssize, int dsize, double scale, dec_alpha* tab) {
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = fmin(scale, ssize - fsx1);
        int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
        if (sx2 > ssize - 1) sx2 = ssize - 1;
        if (sx1 > sx2) sx1 = sx2;
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx1 - 1; tab[k++].alpha = (float)((sx1 - fsx1) / cell);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx; tab[k].si = sx; tab[k++].alpha = (float)(1.0 / cell);
        }
        if (fsx2 - sx2 > 1e-3) {
            tab[k].di = dx; tab[k].si = sx2;
            tab[k++].alpha = (float)(fmin(fmin(fsx2 - sx2, 1.), cell) / cell);
        }
    }
    return k;
}

/* resize(INTER_AREA) when an axis is enlarged: resize.cpp switches to the bilinear kernel with INTER_AREA's own
 * coefficients (area_mode): s = cvFloor(d * scale), f = (d + 1) - (s + 1) / scale clipped to [0, 1); 8-bit images take the
 * fixed-point path: weights saturate_cast<short>(w * 2048), the horizontal pass keeps int sums (a source column beyond the
 * last one contributes S[last] * 2048), the vertical pass is ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2. */
static void area_lin_coef(int d, double scale, double inv_scale, int ssize, int is_x, int* s_out, short* a0, short* a1, int* edge) {
    int s = (int)floor(d * scale);
    float f = (float)((d + 1) - (s + 1) * inv_scale);
    f = f <= 0 ? 0.f : f - (float)cv_floor(f);
    *edge = 0;
    if (is_x) {
        if (s < 0) { f = 0; s = 0; }
        if (s + 1 >= ssize) {
            *edge = 1;                       /* xmax: the pass reads S[s] * ONE */
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
    }
    *s_out = s;
    *a0 = (short)lrintf((1.f - f) * 2048.f);
    *a1 = (short)lrintf(f * 2048.f);
}

static void resize_area_enlarge(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols) {
    const double inv_x = (double)dcols / cols, inv_y = (double)drows / rows;
    const double sx = 1. / inv_x, sy = 1. / inv_y;
    int* xo = (int*)malloc(sizeof(int) * dcols);
    short* xa = (short*)malloc(sizeof(short) * 2 * dcols);
    int* xe = (int*)malloc(sizeof(int) * dcols);
    /* xmax is the FIRST column whose source pair leaves the image; every column from there on takes the edge form */
    int xmax = dcols;
    for (int dx = 0; dx < dcols; ++dx) {
        area_lin_coef(dx, sx, inv_x, cols, 1, &xo[dx], &xa[2 * dx], &xa[2 * dx + 1], &xe[dx]);
        if (xe[dx] && dx < xmax) xmax = dx;
    }
    int* r0 = (int*)malloc(sizeof(int) * dcols);
    int* r1 = (int*)malloc(sizeof(int) * dcols);
    for (int dy = 0; dy < drows; ++dy) {
        int s, e;
        short b0, b1;
        area_lin_coef(dy, sy, inv_y, rows, 0, &s, &b0, &b1, &e);
        const int y0 = s < 0 ? 0 : (s < rows ? s : rows - 1), y1 = s + 1 < 0 ? 0 : (s + 1 < rows ? s + 1 : rows - 1);
        const uint8_t* S0 = src + (size_t)y0 * cols;
        const uint8_t* S1 = src + (size_t)y1 * cols;
        for (int dx = 0; dx < dcols; ++dx) {
            if (dx < xmax) {
                r0[dx] = S0[xo[dx]] * xa[2 * dx] + S0[xo[dx] + 1] * xa[2 * dx + 1];
                r1[dx] = S1[xo[dx]] * xa[2 * dx] + S1[xo[dx] + 1] * xa[2 * dx + 1];
            } else {
                r0[dx] = S0[xo[dx]] * 2048;
                r1[dx] = S1[xo[dx]] * 2048;
            }
        }
        uint8_t* D = dst + (size_t)dy * dcols;
        for (int dx = 0; dx < dcols; ++dx) {
            const int v = (((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2;
            D[dx] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
    free(xo); free(xa); free(xe); free(r0); free(r1);
}

/* returns 0, or -1 for an empty destination */
int oracle_resize_area(const uint8_t* src, int rows, int cols, uint8_t* dst, int drows, int dcols) {
    if (drows <= 0 || dcols <= 0) return -1;
    if (drows > rows || dcols > cols) {   /* scale_x < 1 or scale_y < 1: the area_mode bilinear path */
        resize_area_enlarge(src, rows, cols, dst, drows, dcols);
        return 0;
    }
    /* resize(): inv_scale = dsize/ssize; scale = 1./inv_scale; iscale = saturate_cast<int>(scale) */
    const double sx = 1. / ((double)dcols / cols), sy = 1. / ((double)drows / rows);
    const int isx = cv_round_d(sx), isy = cv_round_d(sy);
    if (fabs(sx - isx) < DBL_EPSILON && fabs(sy - isy) < DBL_EPSILON) {
        /* ResizeAreaFast: integer box; 2x2 has the (a+b+c+d+2)>>2 form, others sum*(1/area) */
        const int area = isx * isy;
        const float scale = 1.f / area;
        for (int y = 0; y < drows; ++y)
            for (int x = 0; x < dcols; ++x) {
                int sum = 0;
                for (int j = 0; j < isy; ++j)
                    for (int i = 0; i < isx; ++i) sum += src[(size_t)(y * isy + j) * cols + x * isx + i];
                dst[(size_t)y * dcols + x] = (isx == 2 && isy == 2) ? (uint8_t)((sum + 2) >> 2) : sat_u8((float)sum * scale);
            }
        return 0;
    }
    dec_alpha* xtab = (dec_alpha*)malloc(sizeof(dec_alpha) * (size_t)cols * 2);
    dec_alpha* ytab = (dec_alpha*)malloc(sizeof(dec_alpha) * (size_t)rows * 2);
    const int nx = area_tab(cols, dcols, sx, xtab), ny = area_tab(rows, drows, sy, ytab);
    float* buf = (float*)malloc(sizeof(float) * dcols);
    float* sum = (float*)calloc(dcols, sizeof(float));
    int prev_dy = ytab[0].di;
    for (int j = 0; j < ny; ++j) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di;
        const uint8_t* S = src + (size_t)ytab[j].si * cols;
        for (int dx = 0; dx < dcols; ++dx) buf[dx] = 0.f;
        for (int k = 0; k < nx; ++k) buf[xtab[k].di] += S[xtab[k].si] * xtab[k].alpha;
        if (dy != prev_dy) {
            uint8_t* D = dst + (size_t)prev_dy * dcols;
            for (int dx = 0; dx < dcols; ++dx) {
                D[dx] = sat_u8(sum[dx]);
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dcols; ++dx) sum[dx] += beta * buf[dx];
        }
    }
    uint8_t* D = dst + (size_t)prev_dy * dcols;
    for (int dx = 0; dx < dcols; ++dx) D[dx] = sat_u8(sum[dx]);
    free(xtab); free(ytab); free(buf); free(sum);
    return 0;
}

/* resize(src, dst, dsize, 0, 0, INTER_AREA) for f32 planes, shrinking (the octave step of the AKAZE scale space,
 * AKAZEFeatures.cpp create_nonlinear_scale_space): exact 2 x 2 boxes take ResizeAreaFast's vector order
 * ((a + b) + (c + d)) * 0.25, everything else the decimation table above with f32 row sums and beta-weighted rows. */
int oracle_resize_area_f32(const float* src, int rows, int cols, float* dst, int drows, int dcols) {
    if (drows <= 0 || dcols <= 0 || drows > rows || dcols > cols) return -1;
    const double sx = 1. / ((double)dcols / cols), sy = 1. / ((double)drows / rows);
    const int isx = cv_round_d(sx), isy = cv_round_d(sy);
    if (fabs(sx - isx) < DBL_EPSILON && fabs(sy - isy) < DBL_EPSILON && isx == 2 && isy == 2) {
        for (int y = 0; y < drows; ++y)
            for (int x = 0; x < dcols; ++x) {
                const float* s = src + (size_t)(2 * y) * cols + 2 * x;
                dst[(size_t)y * dcols + x] = ((s[0] + s[1]) + (s[cols] + s[cols + 1])) * 0.25f;
            }
        return 0;
    }
    dec_alpha* xtab = (dec_alpha*)malloc(sizeof(dec_alpha) * (size_t)cols * 2);
    dec_alpha* ytab = (dec_alpha*)malloc(sizeof(dec_alpha) * (size_t)rows * 2);
    const int nx = area_tab(cols, dcols, sx, xtab), ny = area_tab(rows, drows, sy, ytab);
    float* buf = (float*)malloc(sizeof(float) * dcols);
    float* sum = (float*)calloc(dcols, sizeof(float));
    int prev_dy = ytab[0].di;
    for (int j = 0; j < ny; ++j) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di;
        const float* S = src + (size_t)ytab[j].si * cols;
        for (int dx = 0; dx < dcols; ++dx) buf[dx] = 0.f;
        for (int k = 0; k < nx; ++k) buf[xtab[k].di] += S[xtab[k].si] * xtab[k].alpha;
        if (dy != prev_dy) {
            float* D = dst + (size_t)prev_dy * dcols;
            for (int dx = 0; dx < dcols; ++dx) {
                D[dx] = sum[dx];
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dcols; ++dx) sum[dx] += beta * buf[dx];
        }
    }
    float* D = dst + (size_t)prev_dy * dcols;
    for (int dx = 0; dx < dcols; ++dx) D[dx] = sum[dx];
    free(xtab); free(ytab); free(buf); free(sum);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Spec S: SIFT (sift.dispatch.cpp / sift.simd.hpp), reference call match_finder.py:205-209,220.
 * ---------------------------------------------------------------------------------------- */
#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_INIT_SIGMA 0.5f
#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS (3 * SIFT_ORI_SIG_FCTR)
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_SCL_FCTR 3.f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.f
#define SIFT_FIRST_OCTAVE (-1)

typedef struct { float x, y, size, angle, response; int octave; } okp;

typedef struct {
    int n_octaves, n_layers;      /* nOctaveLayers */
    int* rows; int* cols;         /* per octave */
    float** gauss;                /* [n_octaves*(n_layers+3)] */
    float** dog;                  /* [n_octaves*(n_layers+2)] */
    okp* kp; int n_kp;            /* final, sorted, deduplicated, rescaled to input coordinates */
    float* desc;                  /* [n_kp][128] */
} oracle_sift;

/* S4: getGaussianKernel(ksize, sigma, CV_32F) in its bit-exact form (smooth.dispatch.cpp
 * getGaussianKernelBitExact): doubled coordinates x = 2i-(n-1), t = exp(x*x * (-0.125/sigma^2)),
 * centre tap 1, normalised by 1/sum in double, cast to float. exp here is libm's, not softdouble's. */
int oracle_gauss_kernel(double sigma, float* k /* >= 64 */) {
    int n = cv_round_d(sigma * 4 * 2 + 1) | 1;
    const int n2 = (n - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    double vals[64];
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n2; i++, x += 2) {
        vals[i] = exp((double)(x * x) * scale2x);
        sum += vals[i];
    }
    sum *= 2;
    sum += 1.0;
    const double mul1 = 1.0 / sum;
    for (int i = 0; i < n2; i++) {
        const double t = vals[i] * mul1;
        k[i] = (float)t;
        k[n - 1 - i] = (float)t;
    }
    k[n2] = (float)(1.0 * mul1);
    return n;
}

/* GaussianBlur(src, dst, Size(), sigma, sigma), f32, BORDER_REFLECT_101, separable: row pass then
 * column pass, each s = 0; s = fmaf(k[t], src[..], s) for t ascending (order fixed here). */
void oracle_gauss_blur(const float* src, int rows, int cols, double sigma, float* dst) {
    float k[64];
    const int n = oracle_gauss_kernel(sigma, k), r = n / 2;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)rows * cols);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        const float* s = src + (size_t)y * cols;
        float* d = tmp + (size_t)y * cols;
        for (int x = 0; x < cols; ++x) {
            float acc = 0.f;
            if (x >= r && x + r < cols)
                for (int t = 0; t < n; ++t) acc = fmaf(k[t], s[x + t - r], acc);
            else
                for (int t = 0; t < n; ++t) acc = fmaf(k[t], s[reflect101(x + t - r, cols)], acc);
            d[x] = acc;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < rows; ++y) {
        float* d = dst + (size_t)y * cols;
        for (int x = 0; x < cols; ++x) d[x] = 0.f;
        for (int t = 0; t < n; ++t) {
            const float* s = tmp + (size_t)reflect101(y + t - r, rows) * cols;
            const float kt = k[t];
            for (int x = 0; x < cols; ++x) d[x] = fmaf(kt, s[x], d[x]);
        }
    }
    free(tmp);
}

/* S1: u8 -> f32, resize x2 INTER_LINEAR (resize.cpp: src = (dst+0.5)/2 - 0.5, clamp at the
 * borders; horizontal pass then vertical pass), exact in f32 for 8-bit input. */
static void upsample2_linear(const uint8_t* src, int rows, int cols, float* dst) {
    const int drows = rows * 2, dcols = cols * 2;
    int* xo = (int*)malloc(sizeof(int) * dcols);
    float* xa = (float*)malloc(sizeof(float) * dcols);
    for (int dx = 0; dx < dcols; ++dx) {
        float fx = (float)((dx + 0.5) * 0.5 - 0.5);
        int sx = cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
        xo[dx] = sx; xa[dx] = fx;
    }
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < drows; ++dy) {
        float fy = (float)((dy + 0.5) * 0.5 - 0.5);
        int sy = cv_floor(fy);
        fy -= sy;
        if (sy < 0) { fy = 0; sy = 0; }
        if (sy >= rows - 1) { fy = 0; sy = rows - 1; }
        const int sy1 = sy + 1 < rows ? sy + 1 : sy;
        const uint8_t* s0 = src + (size_t)sy * cols;
        const uint8_t* s1 = src + (size_t)sy1 * cols;
        float* d = dst + (size_t)dy * dcols;
        for (int dx = 0; dx < dcols; ++dx) {
            const int sx = xo[dx], sx1 = sx + 1 < cols ? sx + 1 : sx;
            const float a1 = xa[dx], a0 = 1.f - a1;
            const float h0 = s0[sx] * a0 + s0[sx1] * a1;
            const float h1 = s1[sx] * a0 + s1[sx1] * a1;
            d[dx] = h0 * (1.f - fy) + h1 * fy;
        }
    }
    free(xo); free(xa);
}

/* S7: adjustLocalExtrema. Matx33f::solve(DECOMP_LU) for 3x3 is Cramer's rule (matx.hpp
 * Matx_FastSolveOp<_Tp,3,3,1>), restated with the same expression order in f32. */
static int solve3(const float a[3][3], const float b[3], float x[3]) {
    float d = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
              a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (d == 0) { x[0] = x[1] = x[2] = 0; return 0; }
    d = 1 / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
    return 1;
}

static int adjust_local_extrema(const oracle_sift* S, okp* kpt, int octv, int* layer_io, int* r_io, int* c_io,
                                float contrast_thr, float edge_thr, float sigma) {
    const int nl = S->n_layers, rows = S->rows[octv], cols = S->cols[octv];
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale,
                cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0, contr = 0;
    int i = 0, layer = *layer_io, r = *r_io, c = *c_io;
#define AT(p, rr, cc) (p)[(size_t)(rr) * cols + (cc)]
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const int idx = octv * (nl + 2) + layer;
        const float *img = S->dog[idx], *prev = S->dog[idx - 1], *next = S->dog[idx + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        const float v2 = AT(img, r, c) * 2;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dss = (AT(next, r, c) + AT(prev, r, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float dxs = (AT(next, r, c + 1) - AT(next, r, c - 1) - AT(prev, r, c + 1) + AT(prev, r, c - 1)) * cross_deriv_scale;
        const float dys = (AT(next, r + 1, c) - AT(next, r - 1, c) - AT(prev, r + 1, c) + AT(prev, r - 1, c)) * cross_deriv_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        solve3(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return 0;
        c += cv_round(xc); r += cv_round(xr); layer += cv_round(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= cols - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER || r >= rows - SIFT_IMG_BORDER) return 0;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return 0;
    {
        const int idx = octv * (nl + 2) + layer;
        const float *img = S->dog[idx], *prev = S->dog[idx - 1], *next = S->dog[idx + 1];
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
        contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nl < contrast_thr) return 0;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return 0;
    }
#undef AT
    kpt->x = (c + xc) * (1 << octv);
    kpt->y = (r + xr) * (1 << octv);
    kpt->octave = octv + (layer << 8) + (cv_round_d((xi + 0.5) * 255) << 16);
    kpt->size = sigma * powf(2.f, (layer + xi) / nl) * (1 << octv) * 2;
    kpt->response = fabsf(contr);
    *layer_io = layer; *r_io = r; *c_io = c;
    return 1;
}

/* fastAtan2 stand-in: libm atan2f in degrees [0,360) (S11: OpenCV's own approximation is accurate
 * to ~0.3 deg; not restated). */
static inline float atan2_deg(float y, float x) {
    float a = atan2f(y, x) * (float)(180.0 / M_PI);
    if (a < 0) a += 360.f;
    return a;
}

/* S8: calcOrientationHist */
static float calc_orientation_hist(const float* img, int rows, int cols, int px, int py, int radius, float sigma, float* hist) {
    const int n = SIFT_ORI_HIST_BINS;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    float temphist[SIFT_ORI_HIST_BINS + 4];
    for (int i = 0; i < n + 4; ++i) temphist[i] = 0.f;
    float* th = temphist + 2;
    for (int i = -radius; i <= radius; i++) {
        const int y = py + i;
        if (y <= 0 || y >= rows - 1) continue;
        for (int j = -radius; j <= radius; j++) {
            const int x = px + j;
            if (x <= 0 || x >= cols - 1) continue;
            const float dx = img[(size_t)y * cols + x + 1] - img[(size_t)y * cols + x - 1];
            const float dy = img[(size_t)(y - 1) * cols + x] - img[(size_t)(y + 1) * cols + x];
            const float w = expf((i * i + j * j) * expf_scale);
            const float ori = atan2_deg(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy);
            int bin = cv_round((n / 360.f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            th[bin] += w * mag;
        }
    }
    th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
    float maxval = 0;
    for (int i = 0; i < n; i++) {
        hist[i] = (th[i - 2] + th[i + 2]) * (1.f / 16.f) + (th[i - 1] + th[i + 1]) * (4.f / 16.f) + th[i] * (6.f / 16.f);
        if (i == 0 || hist[i] > maxval) maxval = hist[i];
    }
    return maxval;
}

/* S10: calcSIFTDescriptor */
static void calc_sift_descriptor(const float* img, int rows, int cols, float ptx, float pty, float ori, float scl, float* dst) {
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    const int px = cv_round(ptx), py = cv_round(pty);
    float cos_t = cosf(ori * (float)(M_PI / 180));
    float sin_t = sinf(ori * (float)(M_PI / 180));
    const float bins_per_rad = n / 360.f;
    const float exp_scale = -1.f / (d * d * 0.5f);
    const float hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = cv_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    const int rmax = (int)sqrt(((double)cols) * cols + ((double)rows) * rows);
    if (radius > rmax) radius = rmax;
    cos_t /= hist_width;
    sin_t /= hist_width;
    float hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
    memset(hist, 0, sizeof(hist));
    for (int i = -radius; i <= radius; i++)
        for (int j = -radius; j <= radius; j++) {
            const float c_rot = j * cos_t - i * sin_t;
            const float r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f;
            float cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1) {
                const float dx = img[(size_t)r * cols + c + 1] - img[(size_t)r * cols + c - 1];
                const float dy = img[(size_t)(r - 1) * cols + c] - img[(size_t)(r + 1) * cols + c];
                const float w = expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
                float obin = (atan2_deg(dy, dx) - ori) * bins_per_rad;
                const float mag = sqrtf(dx * dx + dy * dy) * w;
                const int r0 = cv_floor(rbin), c0 = cv_floor(cbin);
                int o0 = cv_floor(obin);
                rbin -= r0; cbin -= c0; obin -= o0;
                if (o0 < 0) o0 += n;
                if (o0 >= n) o0 -= n;
                const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
                const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
                const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
                const float v_rco111 = v_rc11 * obin, v_rco110 = v_rc11 - v_rco111;
                const float v_rco101 = v_rc10 * obin, v_rco100 = v_rc10 - v_rco101;
                const float v_rco011 = v_rc01 * obin, v_rco010 = v_rc01 - v_rco011;
                const float v_rco001 = v_rc00 * obin, v_rco000 = v_rc00 - v_rco001;
                const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
                hist[idx] += v_rco000; hist[idx + 1] += v_rco001;
                hist[idx + (n + 2)] += v_rco010; hist[idx + (n + 3)] += v_rco011;
                hist[idx + (d + 2) * (n + 2)] += v_rco100; hist[idx + (d + 2) * (n + 2) + 1] += v_rco101;
                hist[idx + (d + 3) * (n + 2)] += v_rco110; hist[idx + (d + 3) * (n + 2) + 1] += v_rco111;
            }
        }
    for (int i = 0; i < d; i++)
        for (int j = 0; j < d; j++) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; k++) dst[(i * d + j) * n + k] = hist[idx + k];
        }
    const int len = d * d * n;
    float nrm2 = 0;
    for (int k = 0; k < len; k++) nrm2 += dst[k] * dst[k];
    const float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < len; k++) {
        const float val = dst[k] < thr ? dst[k] : thr;
        dst[k] = val;
        nrm2 += val * val;
    }
    nrm2 = SIFT_INT_DESCR_FCTR / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    for (int k = 0; k < len; k++) dst[k] = (float)sat_u8(dst[k] * nrm2);
}

/* S1's resize alone, for tests that check the base layer of a large image in row bands */
void oracle_upsample2_linear(const uint8_t* src, int rows, int cols, float* dst) { upsample2_linear(src, rows, cols, dst); }

/* S9: KeyPoint12_LessThan of KeyPointsFilter::removeDuplicatedSorted (keypoint.cpp) */
static int kp_less(const void* pa, const void* pb) {
    const okp *a = (const okp*)pa, *b = (const okp*)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->y != b->y) return a->y < b->y ? -1 : 1;
    if (a->size != b->size) return a->size > b->size ? -1 : 1;
    if (a->angle != b->angle) return a->angle < b->angle ? -1 : 1;
    if (a->response != b->response) return a->response > b->response ? -1 : 1;
    if (a->octave != b->octave) return a->octave > b->octave ? -1 : 1;
    return 0;
}

void oracle_sift_free(oracle_sift* S) {
    if (!S) return;
    if (S->gauss) for (int i = 0; i < S->n_octaves * (S->n_layers + 3); ++i) free(S->gauss[i]);
    if (S->dog) for (int i = 0; i < S->n_octaves * (S->n_layers + 2); ++i) free(S->dog[i]);
    free(S->gauss); free(S->dog); free(S->rows); free(S->cols); free(S->kp); free(S->desc);
    free(S);
}

/* SIFT_Impl::detectAndCompute(image, noArray(), keypoints, descriptors, false) */
oracle_sift* oracle_sift_run(const uint8_t* image, int rows, int cols, int n_layers, double contrast_thr,
                             double edge_thr, double sigma, int keep_pyramid) {
    oracle_sift* S = (oracle_sift*)calloc(1, sizeof(oracle_sift));
    S->n_layers = n_layers;
    /* S1: createInitialImage */
    const int brows = rows * 2, bcols = cols * 2;
    float* dbl = (float*)malloc(sizeof(float) * (size_t)brows * bcols);
    upsample2_linear(image, rows, cols, dbl);
    const float sig_diff = sqrtf(fmaxf((float)sigma * (float)sigma - SIFT_INIT_SIGMA * SIFT_INIT_SIGMA * 4, 0.01f));
    float* base = (float*)malloc(sizeof(float) * (size_t)brows * bcols);
    oracle_gauss_blur(dbl, brows, bcols, sig_diff, base);
    free(dbl);
    /* S2 */
    const int mn = brows < bcols ? brows : bcols;
    int n_oct = cv_round_d(log((double)mn) / log(2.) - 2) - SIFT_FIRST_OCTAVE;
    if (n_oct < 1) n_oct = 1;
    S->n_octaves = n_oct;
    S->rows = (int*)malloc(sizeof(int) * n_oct);
    S->cols = (int*)malloc(sizeof(int) * n_oct);
    S->gauss = (float**)calloc((size_t)n_oct * (n_layers + 3), sizeof(float*));
    S->dog = (float**)calloc((size_t)n_oct * (n_layers + 2), sizeof(float*));
    /* S3: buildGaussianPyramid */
    double sig[16];
    sig[0] = sigma;
    const double k = pow(2., 1. / n_layers);
    for (int i = 1; i < n_layers + 3; i++) {
        const double sig_prev = pow(k, (double)(i - 1)) * sigma, sig_total = sig_prev * k;
        sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
    }
    for (int o = 0; o < n_oct; o++) {
        S->rows[o] = o == 0 ? brows : S->rows[o - 1] / 2;
        S->cols[o] = o == 0 ? bcols : S->cols[o - 1] / 2;
        if (S->rows[o] < 1 || S->cols[o] < 1) { S->n_octaves = n_oct = o; break; }
        const size_t px = (size_t)S->rows[o] * S->cols[o];
        for (int i = 0; i < n_layers + 3; i++) {
            float* dst;
            if (o == 0 && i == 0) dst = base;
            else {
                dst = (float*)malloc(sizeof(float) * px);
                if (i == 0) {
                    /* resize(..., INTER_NEAREST) to (cols/2, rows/2): dst(y,x) = src(2y,2x) */
                    const float* src = S->gauss[(o - 1) * (n_layers + 3) + n_layers];
                    const int sc = S->cols[o - 1];
                    for (int y = 0; y < S->rows[o]; ++y)
                        for (int x = 0; x < S->cols[o]; ++x) dst[(size_t)y * S->cols[o] + x] = src[(size_t)(2 * y) * sc + 2 * x];
                } else {
                    oracle_gauss_blur(S->gauss[o * (n_layers + 3) + i - 1], S->rows[o], S->cols[o], sig[i], dst);
                }
            }
            S->gauss[o * (n_layers + 3) + i] = dst;
        }
        /* S5: buildDoGPyramid */
        for (int i = 0; i < n_layers + 2; i++) {
            float* d = (float*)malloc(sizeof(float) * px);
            const float *a = S->gauss[o * (n_layers + 3) + i + 1], *b = S->gauss[o * (n_layers + 3) + i];
            for (size_t p = 0; p < px; ++p) d[p] = a[p] - b[p];
            S->dog[o * (n_layers + 2) + i] = d;
        }
    }
    /* S6: findScaleSpaceExtrema */
    const int threshold = (int)floor(0.5 * contrast_thr / n_layers * 255);
    int cap = 1024, nk = 0;
    okp* kps = (okp*)malloc(sizeof(okp) * cap);
    for (int o = 0; o < n_oct; o++) {
        const int rws = S->rows[o], cls = S->cols[o];
        for (int i = 1; i <= n_layers; i++) {
            const int idx = o * (n_layers + 2) + i;
            const float *img = S->dog[idx], *prev = S->dog[idx - 1], *next = S->dog[idx + 1];
#pragma omp parallel for schedule(dynamic, 8)
            for (int r = SIFT_IMG_BORDER; r < rws - SIFT_IMG_BORDER; r++) {
                for (int c = SIFT_IMG_BORDER; c < cls - SIFT_IMG_BORDER; c++) {
                    const float val = img[(size_t)r * cls + c];
                    if (!(fabsf(val) > threshold)) continue;
                    int is_max = val > 0, is_min = val < 0;
                    for (int dr = -1; dr <= 1 && (is_max || is_min); ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            const size_t p = (size_t)(r + dr) * cls + c + dc;
                            const float a = img[p], b = prev[p], cc = next[p];
                            if (is_max && !(val >= a && val >= b && val >= cc)) is_max = 0;
                            if (is_min && !(val <= a && val <= b && val <= cc)) is_min = 0;
                        }
                    if (!is_max && !is_min) continue;
                    okp kpt;
                    int r1 = r, c1 = c, layer = i;
                    if (!adjust_local_extrema(S, &kpt, o, &layer, &r1, &c1, (float)contrast_thr, (float)edge_thr, (float)sigma)) continue;
                    const float scl_octv = kpt.size * 0.5f / (1 << o);
                    float hist[SIFT_ORI_HIST_BINS];
                    const int n = SIFT_ORI_HIST_BINS;
                    const float omax = calc_orientation_hist(S->gauss[o * (n_layers + 3) + layer], rws, cls, c1, r1,
                                                             cv_round(SIFT_ORI_RADIUS * scl_octv), SIFT_ORI_SIG_FCTR * scl_octv, hist);
                    const float mag_thr = (float)(omax * SIFT_ORI_PEAK_RATIO);
                    for (int j = 0; j < n; j++) {
                        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
                        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= mag_thr) {
                            float bin = j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2 * hist[j] + hist[r2]);
                            bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
                            kpt.angle = 360.f - (float)((360.f / n) * bin);
                            if (fabsf(kpt.angle - 360.f) < FLT_EPSILON) kpt.angle = 0.f;
#pragma omp critical
                            {
                                if (nk == cap) { cap *= 2; kps = (okp*)realloc(kps, sizeof(okp) * cap); }
                                kps[nk++] = kpt;
                            }
                        }
                    }
                }
            }
        }
    }
    /* S9: removeDuplicatedSorted, then the firstOctave < 0 rescale */
    qsort(kps, nk, sizeof(okp), kp_less);
    int j = 0;
    for (int i = 1; i < nk; i++) {
        const okp *a = &kps[j], *b = &kps[i];
        if (a->x != b->x || a->y != b->y || a->size != b->size || a->angle != b->angle) kps[++j] = kps[i];
    }
    if (nk > 0) nk = j + 1;
    for (int i = 0; i < nk; i++) {
        const float scale = 1.f / (float)(1 << -SIFT_FIRST_OCTAVE);
        kps[i].octave = (kps[i].octave & ~255) | ((kps[i].octave + SIFT_FIRST_OCTAVE) & 255);
        kps[i].x *= scale; kps[i].y *= scale; kps[i].size *= scale;
    }
    S->kp = kps;
    S->n_kp = nk;
    /* S10: calcDescriptors */
    S->desc = (float*)malloc(sizeof(float) * 128 * (size_t)(nk > 0 ? nk : 1));
#pragma omp parallel for schedule(dynamic, 16)
    for (int i = 0; i < nk; i++) {
        const okp* kpt = &kps[i];
        int octave = kpt->octave & 255;
        const int layer = (kpt->octave >> 8) & 255;
        octave = octave < 128 ? octave : (-128 | octave);
        const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
        const float size = kpt->size * scale;
        const int o = octave - SIFT_FIRST_OCTAVE;
        float angle = 360.f - kpt->angle;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        calc_sift_descriptor(S->gauss[o * (n_layers + 3) + layer], S->rows[o], S->cols[o], kpt->x * scale, kpt->y * scale,
                             angle, size * 0.5f, S->desc + (size_t)i * 128);
    }
    if (!keep_pyramid) {
        for (int i = 0; i < n_oct * (n_layers + 3); ++i) { free(S->gauss[i]); S->gauss[i] = NULL; }
        for (int i = 0; i < n_oct * (n_layers + 2); ++i) { free(S->dog[i]); S->dog[i] = NULL; }
    }
    return S;
}

int oracle_sift_count(const oracle_sift* S) { return S->n_kp; }
int oracle_sift_octaves(const oracle_sift* S) { return S->n_octaves; }
void oracle_sift_octave_size(const oracle_sift* S, int o, int* rows, int* cols) { *rows = S->rows[o]; *cols = S->cols[o]; }
/* kp: [n][7] = x, y, size, angle, response, octave, class_id(-1); desc: [n][128] */
void oracle_sift_get(const oracle_sift* S, float* kp, float* desc) {
    for (int i = 0; i < S->n_kp; ++i) {
        kp[7 * i + 0] = S->kp[i].x; kp[7 * i + 1] = S->kp[i].y; kp[7 * i + 2] = S->kp[i].size;
        kp[7 * i + 3] = S->kp[i].angle; kp[7 * i + 4] = S->kp[i].response;
        kp[7 * i + 5] = (float)S->kp[i].octave; kp[7 * i + 6] = -1.f;
    }
    if (desc && S->n_kp) memcpy(desc, S->desc, sizeof(float) * 128 * (size_t)S->n_kp);
}
/* kind 0: Gaussian layer (0..n_layers+2), kind 1: DoG layer (0..n_layers+1); needs keep_pyramid */
int oracle_sift_layer(const oracle_sift* S, int kind, int o, int i, float* out) {
    const float* p = kind == 0 ? S->gauss[o * (S->n_layers + 3) + i] : S->dog[o * (S->n_layers + 2) + i];
    if (!p) return -1;
    memcpy(out, p, sizeof(float) * (size_t)S->rows[o] * S->cols[o]);
    return 0;
}
