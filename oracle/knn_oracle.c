/*
 * TEST INFRASTRUCTURE -- CPU oracle for the kNN(k=2) + ratio stage. Never imported by the
 * product path (image_matching_amd/); used only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors and
 * its arithmetic lives in OpenCV (cv2, version not pinned, absent from this image). This file
 * restates the published algorithm of cv2.BFMatcher(NORM_L2).knnMatch(k=2) (OpenCV 4.x
 * modules/features2d/src/matchers.cpp + modules/core/src/batch_distance.cpp) as the reference
 * calls it at image_processing/src/image_processing/match_finder.py:182-183, and the ratio
 * loop at match_finder.py:191-194.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Spec M: dist(q,t) = sqrt(sum_d (q_d - t_d)^2), f32 accumulation in index order;
 * per query an ascending scan over the train rows with strict-< insertion into a sorted list of
 * two, so ties keep the lower train index first. Rows get min(2, nt) entries; a missing entry
 * is idx -1, dist +inf. (match_finder.py:182-183) */
void oracle_knn2(const float* q, int64_t nq, const float* t, int64_t nt, int d, int32_t* idx, float* dist,
                 int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < nq; ++i) {
        const float* a = q + i * d;
        float d1 = INFINITY, d2 = INFINITY;
        int32_t j1 = -1, j2 = -1;
        for (int64_t j = 0; j < nt; ++j) {
            const float* b = t + j * d;
            float s = 0.f;
            for (int k = 0; k < d; ++k) {
                const float e = a[k] - b[k];
                s += e * e;
            }
            const float dd = sqrtf(s);
            if (dd < d1) {
                d2 = d1; j2 = j1; d1 = dd; j1 = (int32_t)j;
            } else if (dd < d2) {
                d2 = dd; j2 = (int32_t)j;
            }
        }
        idx[2 * i] = j1; idx[2 * i + 1] = j2;
        dist[2 * i] = d1; dist[2 * i + 1] = d2;
    }
    (void)threads;
}

/* u8 storage variant for large synthetic sets (same arithmetic: values are widened to f32). */
void oracle_knn2_u8(const uint8_t* q, int64_t nq, const uint8_t* t, int64_t nt, int d, int32_t* idx, float* dist,
                    int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < nq; ++i) {
        const uint8_t* a = q + i * d;
        float d1 = INFINITY, d2 = INFINITY;
        int32_t j1 = -1, j2 = -1;
        for (int64_t j = 0; j < nt; ++j) {
            const uint8_t* b = t + j * d;
            /* integer sum == the f32 sum: every partial sum is an integer below 2^24 */
            int32_t s = 0;
            for (int k = 0; k < d; ++k) {
                const int32_t e = (int32_t)a[k] - (int32_t)b[k];
                s += e * e;
            }
            const float dd = sqrtf((float)s);
            if (dd < d1) {
                d2 = d1; j2 = j1; d1 = dd; j1 = (int32_t)j;
            } else if (dd < d2) {
                d2 = dd; j2 = (int32_t)j;
            }
        }
        idx[2 * i] = j1; idx[2 * i + 1] = j2;
        dist[2 * i] = d1; dist[2 * i + 1] = d2;
    }
    (void)threads;
}

/* Lowe ratio test as the reference's Python evaluates it (match_finder.py:191-194):
 * `m.distance < points_quality * n.distance` -- f32 distances widened to double, double product.
 * Returns the number of survivors, written in query order. Rows with fewer than two
 * neighbours are skipped here (the reference's `for m,n in matches` would raise; the facade
 * keeps that behaviour, see image_matching_amd/match_finder.py). */
int64_t oracle_ratio(const int32_t* idx, const float* dist, int64_t nq, double ratio, int32_t* good_q,
                     int32_t* good_t, float* good_d) {
    int64_t g = 0;
    for (int64_t i = 0; i < nq; ++i) {
        if (idx[2 * i] < 0 || idx[2 * i + 1] < 0) continue;
        if ((double)dist[2 * i] < ratio * (double)dist[2 * i + 1]) {
            good_q[g] = (int32_t)i;
            good_t[g] = idx[2 * i];
            good_d[g] = dist[2 * i];
            ++g;
        }
    }
    return g;
}
