// Brute-force kNN (k = 2) for float descriptors of any length up to 128 -- the matcher of the AKAZE path (SURVEY 8f
// rank 3): cv2.BFMatcher(cv2.NORM_L2).knnMatch(descriptors_1, roi.dp, k=2) (match_finder.py:182-183) on the 64-float
// M-SURF descriptors the checked-in detector produces. These are not small integers, so the int8 MFMA kernel of knn.hip
// (exact only for 0..255 values) does not apply; this one computes sum_k (q_k - t_k)^2 in f32 in index order, one
// sequential chain per pair -- the order of oracle_knn2 -- so distances, indices and ties are bit-identical to the oracle.
// fp32 VALU-bound: 1.5 packed instructions per pair and dimension, operands from LDS tiles stored dimension-major.
#include <math.h>
#include "common.h"

#define KF_TQ 64
#define KF_TT 64

struct kf_top2 { float d1, d2; int j1, j2; };
typedef float kf2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void kf_insert(kf_top2& b, float dd, int j) {   // (dist, index) lexicographic
    if (dd < b.d1 || (dd == b.d1 && j < b.j1)) { b.d2 = b.d1; b.j2 = b.j1; b.d1 = dd; b.j1 = j; }
    else if (dd < b.d2 || (dd == b.d2 && j < b.j2)) { b.d2 = dd; b.j2 = j; }
}

// grid (query tiles, train chunks); partial: [chunk][nq] of kf_top2
__global__ void __launch_bounds__(256)
knn2_f32_kernel(const float* __restrict__ q, int64_t nq, const float* __restrict__ t, int64_t nt, int d, int tiles_per_chunk,
                kf_top2* __restrict__ partial) {
    extern __shared__ float lds[];
    float* qs = lds;                      // [d][64]
    float* ts = lds + (size_t)d * KF_TQ;  // [d][64]
    const int tid = threadIdx.x, tq = tid & 15, tt = tid >> 4;
    const int64_t q0 = (int64_t)blockIdx.x * KF_TQ;
    // Tiles go into LDS dimension-major. With consecutive lanes on consecutive DIMENSIONS of a row every lane of a wave wrote
    // the same LDS bank (addresses 64 floats apart: a 64-way conflict, 40 % of the kernel); consecutive lanes on consecutive
    // ROWS write side by side, and each lane's 16-byte global reads walk its row, whose lines stay in the vector cache.
    auto load_tile = [&](float* dstl, const float* __restrict__ src, int64_t r0, int64_t n_rows) {
        if ((d & 3) == 0) {
            const int d4 = d >> 2;
            for (int i = tid; i < 64 * d4; i += 256) {
                const int r = i & 63, kq = i >> 6;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r0 + r < n_rows) v = *(const float4*)(src + (r0 + r) * d + 4 * kq);
                dstl[(4 * kq) * 64 + r] = v.x; dstl[(4 * kq + 1) * 64 + r] = v.y;
                dstl[(4 * kq + 2) * 64 + r] = v.z; dstl[(4 * kq + 3) * 64 + r] = v.w;
            }
        } else {
            for (int i = tid; i < 64 * d; i += 256) {
                const int r = i & 63, k = i >> 6;
                dstl[k * 64 + r] = r0 + r < n_rows ? src[(r0 + r) * d + k] : 0.f;
            }
        }
    };
    static_assert(KF_TQ == 64 && KF_TT == 64, "load_tile walks 64-row tiles");
    load_tile(qs, q, q0, nq);
    kf_top2 best[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { best[a].d1 = best[a].d2 = INFINITY; best[a].j1 = best[a].j2 = -1; }
    const int64_t tile0 = (int64_t)blockIdx.y * tiles_per_chunk;
    for (int ti = 0; ti < tiles_per_chunk; ++ti) {
        const int64_t t0 = (tile0 + ti) * KF_TT;
        if (t0 >= nt) break;
        __syncthreads();
        load_tile(ts, t, t0, nt);
        __syncthreads();
        // two train rows per instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_add_f32: the same subtraction, product and
        // sum per pair as the scalar form, in the same order -- 1.5 instead of 3 instructions per pair and dimension)
        kf2 s2[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) { s2[a][0] = kf2{0.f, 0.f}; s2[a][1] = kf2{0.f, 0.f}; }
        for (int k = 0; k < d; ++k) {
            const float4 qa = *(const float4*)(qs + k * KF_TQ + tq * 4);
            const float4 tb = *(const float4*)(ts + k * KF_TT + tt * 4);
            const float qv[4] = {qa.x, qa.y, qa.z, qa.w};
            const kf2 t01 = {tb.x, tb.y}, t23 = {tb.z, tb.w};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const kf2 q2 = {qv[a], qv[a]};
                const kf2 e01 = q2 - t01, e23 = q2 - t23;
                s2[a][0] += e01 * e01;
                s2[a][1] += e23 * e23;
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {   // ascending train index within the thread
            const int64_t j = t0 + tt * 4 + b;
            if (j < nt) {
#pragma unroll
                for (int a = 0; a < 4; ++a) kf_insert(best[a], sqrtf(s2[a][b >> 1][b & 1]), (int)j);
            }
        }
    }
    // merge the 16 lists of each query
    __syncthreads();
    kf_top2* m = (kf_top2*)lds;   // [64 queries][16]
#pragma unroll
    for (int a = 0; a < 4; ++a) m[(tq * 4 + a) * 16 + tt] = best[a];
    __syncthreads();
    if (tid < KF_TQ && q0 + tid < nq) {
        kf_top2 r = m[tid * 16];
        for (int i = 1; i < 16; ++i) {
            const kf_top2 o = m[tid * 16 + i];
            if (o.j1 >= 0) kf_insert(r, o.d1, o.j1);
            if (o.j2 >= 0) kf_insert(r, o.d2, o.j2);
        }
        partial[(int64_t)blockIdx.y * nq + q0 + tid] = r;
    }
}

__global__ void __launch_bounds__(256)
knn2_f32_finish_kernel(const kf_top2* __restrict__ partial, int n_chunks, int64_t nq, int* __restrict__ idx, float* __restrict__ dist) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq) return;
    kf_top2 r = partial[i];
    for (int c = 1; c < n_chunks; ++c) {
        const kf_top2 o = partial[(int64_t)c * nq + i];
        if (o.j1 >= 0) kf_insert(r, o.d1, o.j1);
        if (o.j2 >= 0) kf_insert(r, o.d2, o.j2);
    }
    idx[2 * i] = r.j1; idx[2 * i + 1] = r.j2;
    dist[2 * i] = r.d1; dist[2 * i + 1] = r.d2;
}

struct imatch_fset {
    imatch_ctx* ctx;
    float* desc;   // device [n][d], from the context's block pool
    int64_t n;
    int d;
};

// q_host [nq][d] is uploaded, or the queries are already on the device (q_dev); the train rows are either uploaded too
// (t_host) or already resident (t_dev)
static int knn2_f32_run(imatch_ctx* ctx, const float* q_host, int64_t nq, const float* t_host, const float* t_dev, int64_t nt, int d,
                        int32_t* idx, float* dist, const float* q_dev = nullptr) {
    if (nq > INT32_MAX / 2 || nt > INT32_MAX / 2) return imatch_set_err(ctx, IMATCH_ESIZE, "set too large");
    if (nq == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n_tiles = (nt + KF_TT - 1) / KF_TT;
    const int q_tiles = (int)((nq + KF_TQ - 1) / KF_TQ);
    int n_chunks = (int)std::min<int64_t>(std::max<int64_t>(1, (2048 + q_tiles - 1) / q_tiles), std::max<int64_t>(1, n_tiles / 4));
    n_chunks = std::min(n_chunks, 1024);
    const int tiles_per_chunk = (int)std::max<int64_t>(1, (n_tiles + n_chunks - 1) / n_chunks);
    n_chunks = (int)std::max<int64_t>(1, (n_tiles + tiles_per_chunk - 1) / tiles_per_chunk);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t bq = q_dev ? 0 : up((size_t)nq * d * 4), bt = t_host ? up((size_t)std::max<int64_t>(nt, 1) * d * 4) : 0;
    const size_t bp = up((size_t)n_chunks * nq * sizeof(kf_top2)), bi = up((size_t)nq * 8), bd = up((size_t)nq * 8);
    int rc = imatch_ws_reserve(ctx, bq + bt + bp + bi + bd);
    if (rc) return rc;
    char* p = (char*)ctx->ws;
    const float* qd = q_dev ? q_dev : (const float*)p;
    float* q_up = (float*)p; p += bq;
    const float* td = t_host ? (const float*)p : t_dev;
    p += bt;
    kf_top2* part = (kf_top2*)p; p += bp;
    int* idx_d = (int*)p; p += bi;
    float* dist_d = (float*)p;
    rc = imatch_hs_reserve(ctx, bq + bt + bi + bd);
    if (rc) return rc;
    if (!q_dev) memcpy(ctx->hs, q_host, (size_t)nq * d * 4);
    if (t_host && nt) memcpy((char*)ctx->hs + bq, t_host, (size_t)nt * d * 4);
    const size_t up_bytes = bq + ((t_host && nt) ? (size_t)nt * d * 4 : 0);
    if (up_bytes) HIP_TRY(ctx, hipMemcpyAsync(q_up, ctx->hs, up_bytes, hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = std::max((size_t)2 * d * 64 * 4, (size_t)64 * 16 * sizeof(kf_top2));
    hipLaunchKernelGGL(knn2_f32_kernel, dim3(q_tiles, n_chunks), dim3(256), lds, ctx->stream, qd, nq, td, nt, d, tiles_per_chunk, part);
    hipLaunchKernelGGL(knn2_f32_finish_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream, part, n_chunks, nq,
                       idx_d, dist_d);
    HIP_TRY(ctx, hipGetLastError());
    // results come back through the pinned block, behind the uploads
    char* hres = (char*)ctx->hs + bq + bt;
    HIP_TRY(ctx, hipMemcpyAsync(hres, idx_d, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hres + bi, dist_d, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(idx, hres, (size_t)nq * 8);
    memcpy(dist, hres + bi, (size_t)nq * 8);
    return IMATCH_OK;
}

extern "C" {

int imatch_knn2_f32(imatch_ctx* ctx, const float* q, int64_t nq, const float* t, int64_t nt, int d, int32_t* idx, float* dist) {
    if (!ctx || nq < 0 || nt < 0 || d < 1 || d > 128 || (nq > 0 && (!q || !idx || !dist)) || (nt > 0 && !t))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    static const float none = 0.f;
    return knn2_f32_run(ctx, q, nq, nt ? t : &none, nullptr, nt, d, idx, dist);
}

int imatch_fset_from_host(imatch_ctx* ctx, const float* desc, int64_t n, int d, imatch_fset** out) {
    if (!ctx || !out || n < 0 || d < 1 || d > 128 || (n > 0 && !desc)) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    if (n > INT32_MAX / 2) return imatch_set_err(ctx, IMATCH_ESIZE, "set too large");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    imatch_fset* f = (imatch_fset*)calloc(1, sizeof *f);
    if (!f) return imatch_set_err(ctx, IMATCH_ENOMEM, "host allocation failed");
    f->ctx = ctx; f->n = n; f->d = d;
    const size_t bytes = (size_t)std::max<int64_t>(n, 1) * d * 4;
    f->desc = (float*)imatch_pool_alloc(ctx, bytes);
    if (!f->desc) {
        free(f);
        return imatch_set_err(ctx, IMATCH_ENOMEM, "device block of %zu bytes", bytes);
    }
    if (n) {
        int rc = imatch_hs_reserve(ctx, bytes);
        if (rc) { imatch_pool_free(ctx, f->desc); free(f); return rc; }
        memcpy(ctx->hs, desc, bytes);
        hipError_t e = hipMemcpyAsync(f->desc, ctx->hs, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            imatch_pool_free(ctx, f->desc);
            free(f);
            return imatch_set_err(ctx, IMATCH_EHIP, "upload: %s", hipGetErrorString(e));
        }
    }
    *out = f;
    return IMATCH_OK;
}

int64_t imatch_fset_count(const imatch_fset* f) { return f ? f->n : 0; }

void imatch_fset_free(imatch_fset* f) {
    if (!f) return;
    imatch_pool_free(f->ctx, f->desc);
    free(f);
}

int imatch_knn2_fset_akaze(imatch_ctx* ctx, uint64_t serial, const imatch_fset* t, int32_t* idx, float* dist) {
    if (!ctx || !t || t->ctx != ctx || !idx || !dist) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    const float* qd = nullptr;
    int64_t nq = 0;
    if (imatch_akaze_last_dev(ctx, serial, &qd, &nq) != IMATCH_OK)
        return imatch_set_err(ctx, IMATCH_EINVAL, "AKAZE extraction %llu is not this context's last one: its descriptors are no "
                              "longer on the device (use imatch_knn2_fset with the host matrix)", (unsigned long long)serial);
    if (t->d != 64) return imatch_set_err(ctx, IMATCH_EINVAL, "train set has %d-float rows, AKAZE's descriptors have 64", t->d);
    return knn2_f32_run(ctx, nullptr, nq, nullptr, t->desc, t->n, t->d, idx, dist, qd);
}

int imatch_knn2_fset(imatch_ctx* ctx, const float* q, int64_t nq, const imatch_fset* t, int32_t* idx, float* dist) {
    if (!ctx || !t || t->ctx != ctx || nq < 0 || (nq > 0 && (!q || !idx || !dist)))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    return knn2_f32_run(ctx, q, nq, nullptr, t->desc, t->n, t->d, idx, dist);
}

}  // extern "C"
