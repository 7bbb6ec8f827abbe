/*
 * TEST INFRASTRUCTURE -- the CPU BASELINE's matcher (bench.py's cpu_baseline leg), not the checker.
 *
 * oracle_knn2 (knn_oracle.c) is the checker: one scalar loop, a sqrtf per pair, written for obviousness. As a
 * BASELINE it undersells the CPU it stands for: cv2.BFMatcher(NORM_L2).knnMatch (reference match_finder.py:182-183)
 * runs OpenCV's batchDistance over CV_32F rows with SIMD normL2Sqr and takes square roots of the winners only. This
 * file is that shape of computation: f32 rows, squared distances accumulated in vector lanes, train rows walked in
 * cache-sized tiles with the queries inside (a tile is re-used from L2 by every query of the thread), strict-< insertion
 * in ascending train index (the lower index wins ties), sqrtf of the two winners at the end. The vector width follows
 * the machine it RUNS on (AVX-512 / AVX2+FMA / plain C bodies picked by cpuid at the call), not the one
 * it was compiled on -- liboracle.so is built in the container and travels to the GPU box.
 *
 * For SIFT's integer-valued rows every partial sum is an integer below 2^24, so the lane-wise sums equal the scalar
 * sum bit for bit and the result is oracle_knn2's, index for index and distance for distance (tests/test_oracle_knn.py
 * and bench.py assert it). PARITY UNPINNED like the rest of oracle/ (no OpenCV here).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE_ROWS 512      /* 512 rows x 128 floats = 256 KB: stays in a core's L2 while the thread's queries pass over it */

#if defined(__x86_64__)
#include <immintrin.h>
#define HAVE_X86 1
#else
#define HAVE_X86 0
#endif

/* one query against the rows [r0, r1) of the train matrix: running (squared distance, index) top-2, ascending rows,
 * strict <. Three bodies, one per instruction set; d is a multiple of 16. */
typedef void (*tile_fn)(const float* a, const float* t, int64_t r0, int64_t r1, int d, float* s1, float* s2, int32_t* j1,
                        int32_t* j2);

#define INSERT(s, j)                                                        \
    if ((s) < *s2) { /* most rows fail this one compare */                  \
        if ((s) < *s1) { *s2 = *s1; *j2 = *j1; *s1 = (s); *j1 = (int32_t)(j); } \
        else { *s2 = (s); *j2 = (int32_t)(j); }                             \
    }

static void tile_scalar(const float* a, const float* t, int64_t r0, int64_t r1, int d, float* s1, float* s2, int32_t* j1,
                        int32_t* j2) {
    for (int64_t j = r0; j < r1; ++j) {
        const float* b = t + j * d;
        float acc[4] = {0, 0, 0, 0};
        for (int k = 0; k < d; k += 4)
            for (int u = 0; u < 4; ++u) {
                const float e = a[k + u] - b[k + u];
                acc[u] += e * e;
            }
        const float s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        INSERT(s, j)
    }
}

#if HAVE_X86
__attribute__((target("avx2,fma")))
static void tile_avx2(const float* a, const float* t, int64_t r0, int64_t r1, int d, float* s1, float* s2, int32_t* j1,
                      int32_t* j2) {
    for (int64_t j = r0; j < r1; ++j) {
        const float* b = t + j * d;
        __m256 acc0 = _mm256_setzero_ps(), acc1 = _mm256_setzero_ps();
        for (int k = 0; k < d; k += 16) {
            const __m256 e0 = _mm256_sub_ps(_mm256_loadu_ps(a + k), _mm256_loadu_ps(b + k));
            const __m256 e1 = _mm256_sub_ps(_mm256_loadu_ps(a + k + 8), _mm256_loadu_ps(b + k + 8));
            acc0 = _mm256_fmadd_ps(e0, e0, acc0);
            acc1 = _mm256_fmadd_ps(e1, e1, acc1);
        }
        const __m256 v = _mm256_add_ps(acc0, acc1);
        __m128 h = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
        h = _mm_add_ps(h, _mm_movehl_ps(h, h));
        h = _mm_add_ss(h, _mm_shuffle_ps(h, h, 1));
        const float s = _mm_cvtss_f32(h);
        INSERT(s, j)
    }
}

__attribute__((target("avx512f")))
static void tile_avx512(const float* a, const float* t, int64_t r0, int64_t r1, int d, float* s1, float* s2, int32_t* j1,
                        int32_t* j2) {
    if (d == 128) {   /* SIFT: the query stays in eight registers */
        __m512 qa[8];
        for (int k = 0; k < 8; ++k) qa[k] = _mm512_loadu_ps(a + 16 * k);
        for (int64_t j = r0; j < r1; ++j) {
            const float* b = t + j * 128;
            __m512 acc0 = _mm512_setzero_ps(), acc1 = _mm512_setzero_ps();
            for (int k = 0; k < 8; k += 2) {
                const __m512 e0 = _mm512_sub_ps(qa[k], _mm512_loadu_ps(b + 16 * k));
                const __m512 e1 = _mm512_sub_ps(qa[k + 1], _mm512_loadu_ps(b + 16 * k + 16));
                acc0 = _mm512_fmadd_ps(e0, e0, acc0);
                acc1 = _mm512_fmadd_ps(e1, e1, acc1);
            }
            const float s = _mm512_reduce_add_ps(_mm512_add_ps(acc0, acc1));
            INSERT(s, j)
        }
        return;
    }
    for (int64_t j = r0; j < r1; ++j) {
        const float* b = t + j * d;
        __m512 acc = _mm512_setzero_ps();
        for (int k = 0; k < d; k += 16) {
            const __m512 e = _mm512_sub_ps(_mm512_loadu_ps(a + k), _mm512_loadu_ps(b + k));
            acc = _mm512_fmadd_ps(e, e, acc);
        }
        const float s = _mm512_reduce_add_ps(acc);
        INSERT(s, j)
    }
}
#endif

#if HAVE_X86
/* two queries per pass over the tile (d = 128): a train row's eight loads serve both, which halves the L2 traffic
 * that bounds the one-query body */
__attribute__((target("avx512f")))
static void tile2_avx512(const float* a0, const float* a1, const float* t, int64_t r0, int64_t r1, float* best, int32_t* idx) {
    __m512 qa[8], qb[8];
    for (int k = 0; k < 8; ++k) { qa[k] = _mm512_loadu_ps(a0 + 16 * k); qb[k] = _mm512_loadu_ps(a1 + 16 * k); }
    float *s1 = best, *s2 = best + 1;
    int32_t *j1 = idx, *j2 = idx + 1;
    float *u1 = best + 2, *u2 = best + 3;
    int32_t *v1 = idx + 2, *v2 = idx + 3;
    for (int64_t j = r0; j < r1; ++j) {
        const float* b = t + j * 128;
        __m512 acc0 = _mm512_setzero_ps(), acc1 = _mm512_setzero_ps();
        for (int k = 0; k < 8; ++k) {
            const __m512 row = _mm512_loadu_ps(b + 16 * k);
            const __m512 e0 = _mm512_sub_ps(qa[k], row), e1 = _mm512_sub_ps(qb[k], row);
            acc0 = _mm512_fmadd_ps(e0, e0, acc0);
            acc1 = _mm512_fmadd_ps(e1, e1, acc1);
        }
        const float sa = _mm512_reduce_add_ps(acc0), sb = _mm512_reduce_add_ps(acc1);
        INSERT(sa, j)
        if (sb < *u2) {
            if (sb < *u1) { *u2 = *u1; *v2 = *v1; *u1 = sb; *v1 = (int32_t)j; }
            else { *u2 = sb; *v2 = (int32_t)j; }
        }
    }
}
#endif

static tile_fn pick_tile_fn(void) {
#if HAVE_X86
    __builtin_cpu_init();
    if (__builtin_cpu_supports("avx512f")) return tile_avx512;
    if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return tile_avx2;
#endif
    return tile_scalar;
}

void oracle_knn2_fast(const float* q, int64_t nq, const float* t, int64_t nt, int d, int32_t* idx, float* dist,
                      int threads) {
    if (d % 16 != 0) {   /* other lengths: not the baseline's case */
        for (int64_t i = 0; i < nq; ++i) { idx[2 * i] = idx[2 * i + 1] = -1; dist[2 * i] = dist[2 * i + 1] = INFINITY; }
        return;
    }
    const tile_fn tile = pick_tile_fn();
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    float* best = (float*)malloc(sizeof(float) * 2 * (size_t)(nq > 0 ? nq : 1));   /* squared distances */
    for (int64_t i = 0; i < 2 * nq; ++i) { best[i] = INFINITY; idx[i] = -1; }
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
#ifdef _OPENMP
        const int nth = omp_get_num_threads(), me = omp_get_thread_num();
#else
        const int nth = 1, me = 0;
#endif
        /* a contiguous block of queries per thread; every thread walks the tiles in ascending order, so each query sees
         * the train rows in ascending index */
        const int64_t q0 = nq * me / nth, q1 = nq * (me + 1) / nth;
        for (int64_t r0 = 0; r0 < nt; r0 += TILE_ROWS) {
            const int64_t r1 = r0 + TILE_ROWS < nt ? r0 + TILE_ROWS : nt;
            int64_t i = q0;
#if HAVE_X86
            if (tile == tile_avx512 && d == 128)
                for (; i + 1 < q1; i += 2) tile2_avx512(q + i * d, q + (i + 1) * d, t, r0, r1, best + 2 * i, idx + 2 * i);
#endif
            for (; i < q1; ++i)
                tile(q + i * d, t, r0, r1, d, best + 2 * i, best + 2 * i + 1, idx + 2 * i, idx + 2 * i + 1);
        }
    }
    for (int64_t i = 0; i < 2 * nq; ++i) dist[i] = idx[i] >= 0 ? sqrtf(best[i]) : INFINITY;
    free(best);
    (void)threads;
}
