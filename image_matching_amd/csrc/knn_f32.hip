// Brute-force kNN (k = 2) for float descriptors of any length up to 128 -- the matcher of the AKAZE path (SURVEY 8f
// rank 3): cv2.BFMatcher(cv2.NORM_L2).knnMatch(descriptors_1, roi.dp, k=2) (match_finder.py:182-183) on the 64-float
// M-SURF descriptors the checked-in detector produces. These are not small integers, so the int8 MFMA kernel of knn.hip
// (exact only for 0..255 values) does not apply; this one computes sum_k (q_k - t_k)^2 in f32 in index order, one
// sequential chain per pair -- the order of oracle_knn2 -- so distances, indices and ties are bit-identical to the oracle.
// fp32 VALU-bound: 1.5 packed instructions per pair and dimension, operands from LDS tiles stored dimension-major.
#include <math.h>
#include "common.h"
#include <vector>

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

// ---- the same answer by filter and refine: f16 matrix cores find the few rows that CAN be a query's two nearest, the exact
// f32 chain runs on those only ------------------------------------------------------------------------------------------
// ssd(q, t) = |q|^2 + |t|^2 - 2 q.t with q.t from v_mfma_f32_16x16x32_f16 on rows rounded to f16 is an APPROXIMATION a of the
// oracle's value s (the sequential f32 sum of (q_k - t_k)^2), but one with a bound: f16 keeps 11 significant bits (values
// below 2^-14 an absolute 2^-25), the products are summed in f32, so |a - s| <= e_q + e_t with
//     e_row = |row|^2 * 2^-9 + |row| * 2^-20
// (twice the first-order bound 2^-10 |q||t| + 2^-25 sqrt(d) (|q| + |t|), which leaves room for the f32 summation errors on
// both sides; the second term is the f16 SUBNORMAL range, which conversion and MFMA both keep on gfx950 -- rows whose every
// component is subnormal in f16 go through the filter exactly, tests/test_knn_f32_gpu.py "tiny": were they flushed, every
// row would be a candidate and the lists would overflow). Every lane keeps the two smallest UPPER bounds U = a + e it has seen for its query: two rows lie at or below
// U2, so the query's true second-best does too, and a row can only belong to the true top-2 if its LOWER bound a - e is
// <= U2 (the factor 1 + 2^-18 covers rows whose s is larger but whose sqrtf rounds to the same float -- the order is on the
// distance, ties to the lower index). Such rows are appended to the query's candidate list; U2 only falls while the map is
// walked, so the list is a superset of what the final bound would admit (a few hundred rows per query with 10 chunks of a
// 30 k-row set, of 30 k). The refinement evaluates the oracle's own chain on the candidates and takes (distance, index)
// top-2 -- bit-identical to knn2_f32_kernel by construction, at 1/5 of its time on a frame against a crop's descriptors.
// Sets with values beyond +-1024 or non-finite ones, query sets of more than 16384 rows, train sets of fewer than 4096 and
// a candidate list that overflows all take knn2_f32_kernel.
typedef _Float16 kh8 __attribute__((ext_vector_type(8)));
typedef float kf4 __attribute__((ext_vector_type(4)));
#define KF_CAND_CAP 1024
#define KF_STAGE 512      // candidates a wave stages in LDS before it writes them out (a tile adds at most 256)
#define KF_MAXABS 1024.f

// one wave per row: f16 copy padded to dpad (a multiple of 32) dimensions, |row|^2, the row's error budget; rows n .. n_pad
// become zero rows with an infinite norm (never a candidate). flag |= 1 when a value is not finite or beyond KF_MAXABS.
// frag: rows are stored in the ORDER THE MFMA's A operand is read -- per tile of 16 rows and k-step of 32 dimensions one
// 1 KiB block in which lane (row & 15) + 16 * ((k >> 3) & 3) finds its eight halves at 16 * lane bytes: a fragment load is
// then one contiguous kilobyte per wave. (Row-major, consecutive lanes sat 128 bytes apart: sixteen cache lines per
// quarter wave, and the vector cache's tag rate, not the arithmetic, was the filter kernel.)
__global__ void __launch_bounds__(256)
kf_prep_kernel(const float* __restrict__ src, int64_t n, int64_t n_pad, int d, int dpad, _Float16* __restrict__ dst,
               float* __restrict__ norm, float* __restrict__ err, int* __restrict__ flag, int frag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    float s = 0.f;
    bool bad = false;
    for (int k = lane; k < dpad; k += 64) {
        float v = 0.f;
        if (row < n && k < d) {
            v = src[row * d + k];
            const unsigned u = __float_as_uint(v) & 0x7fffffffu;
            bad = bad || u > __float_as_uint(KF_MAXABS);   // also NaN and infinity (larger bit patterns)
        }
        const int64_t at = frag ? ((((row >> 4) * (dpad >> 5) + (k >> 5)) * 64 + ((k >> 3) & 3) * 16 + (row & 15)) * 8 + (k & 7))
                                : row * dpad + k;
        dst[at] = (_Float16)v;
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        norm[row] = row < n ? s : INFINITY;
        err[row] = row < n ? s * (1.f / 512.f) + sqrtf(s) * (1.f / 1048576.f) : 0.f;
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicOr(flag, 1);
}

// grid (query tiles of 64 * KF_QB, train chunks); a wave owns KF_QB blocks of 16 queries (the B operand), train tiles of 16
// rows are the A operand, loaded once per tile for all of the wave's blocks: the loop is a chain of L2 round trips, and what
// it costs is set by how many trips a wave makes, not by its arithmetic.
// Bounds: every workgroup first walks the set's first four tiles WITHOUT emitting (64 rows give every query a first bound:
// a chunk that starts from nothing emits every row it sees until it has seen two), and the lanes publish their bounds into
// gb[query] (float bits, atomicMin: the values are >= 0) and read the chip-wide minimum back -- after every tile at first,
// then every fourth. Any published value is the second smallest upper bound over SOME set of distinct rows, hence at or
// above the query's true second-best: a stale or foreign bound only admits more candidates, never fewer than needed.
#ifndef KF_QB
#define KF_QB 1         // blocks of 16 queries per wave (2 and 4 were measured: no faster, more candidates)
#endif
#ifndef KF_XCH_MASK
#define KF_XCH_MASK 3     // an old chunk exchanges bounds every KF_XCH_MASK + 1 tiles
#endif
#ifndef KF_YOUNG
#define KF_YOUNG 4        // tiles at a chunk's start that exchange every tile
#endif
#ifndef KF_DEFER
#define KF_DEFER 0        // 1: old chunks fold a request in at the NEXT exchange point instead of at the bottom of the same tile
                          // (measured: the staler bounds admit half as many candidates again, and nothing is gained)
#endif
template <int NKS>   // k-steps of 32 dimensions: dpad / 32
__global__ void __launch_bounds__(256)
kf_filter_kernel(const _Float16* __restrict__ qh, const float* __restrict__ qn, const float* __restrict__ qe, int64_t nq,
                 const _Float16* __restrict__ th, const float* __restrict__ tn, const float* __restrict__ te, int64_t nt,
                 int dpad, int tiles_per_chunk, int* __restrict__ gb, int* __restrict__ cnt, int* __restrict__ list) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int64_t q_wave0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * KF_QB);
    int64_t q[KF_QB];
    bool q_ok[KF_QB];
    kh8 bq[KF_QB][NKS];
    float nq_l[KF_QB], eq_l[KF_QB];
    int* gq[KF_QB];
#pragma unroll
    for (int b = 0; b < KF_QB; ++b) {
        q[b] = q_wave0 + 16 * b + c;
        q_ok[b] = q[b] < nq;
        const int64_t qr = q_ok[b] ? q[b] : nq - 1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) bq[b][ks] = *(const kh8*)(qh + qr * dpad + ks * 32 + 8 * g);
        nq_l[b] = qn[qr];
        eq_l[b] = qe[qr];
        gq[b] = gb + qr;
    }
    const int64_t ntiles = (nt + 15) >> 4;
    // a tile's operands: the lane's A fragments, the norms and error budgets of its four rows
    struct TileOps { kh8 a[NKS]; kf4 n4, e4; };
    auto tile_load = [&](int64_t tile, TileOps& o) {
        const int64_t r0 = tile * 16;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) o.a[ks] = *(const kh8*)(th + ((tile * NKS + ks) * 64 + lane) * 8);   // fragment order: see kf_prep_kernel
        o.n4 = *(const kf4*)(tn + r0 + 4 * g);
        o.e4 = *(const kf4*)(te + r0 + 4 * g);
    };
    // upper bounds U[4] and lower bounds L[4] of the lane's four rows of a tile, for query block b
    auto tile_bounds = [&](const TileOps& o, int b, float (&U)[4], float (&L)[4]) {
        kf4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.a[ks], bq[b][ks], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = (nq_l[b] + o.n4[i]) - 2.f * acc[i];
            const float e = eq_l[b] + o.e4[i];
            U[i] = fmaxf(a + e, 0.f);
            L[i] = a - e;
        }
    };
    auto top2 = [](float& U1, float& U2, float U) {
        const float t = fmaxf(U1, U);
        U1 = fminf(U1, U);
        U2 = fminf(U2, t);
    };
    // ---- the seed: a bound from the first rows of the set, merged over the four lanes that share a query (disjoint rows)
    float bound[KF_QB];
    {
        float S1[KF_QB], S2[KF_QB], U[4], L[4];
#pragma unroll
        for (int b = 0; b < KF_QB; ++b) S1[b] = S2[b] = INFINITY;
        TileOps so[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tile_load(u < ntiles ? u : ntiles - 1, so[u]);   // all in flight together
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= ntiles) break;
#pragma unroll
            for (int b = 0; b < KF_QB; ++b) {
                tile_bounds(so[u], b, U, L);
#pragma unroll
                for (int i = 0; i < 4; ++i) top2(S1[b], S2[b], U[i]);
            }
        }
#pragma unroll
        for (int b = 0; b < KF_QB; ++b) {
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float o1 = __shfl_xor(S1[b], off), o2 = __shfl_xor(S2[b], off);
                const float hi = fmaxf(S1[b], o1);
                S1[b] = fminf(S1[b], o1);
                S2[b] = fminf(hi, fminf(S2[b], o2));
            }
            bound[b] = S2[b];
        }
    }
    // The exchange makes nobody wait once a chunk is under way: the chip-wide words are asked for at one exchange point
    // (fetch) and folded in at the NEXT one (take), a run of tiles later; a lane publishes only what improves on everything it
    // has seen, with an atomic whose result nobody reads.
    int g_seen[KF_QB], g_pending[KF_QB];
    auto fetch = [&]() {
#pragma unroll
        for (int b = 0; b < KF_QB; ++b) g_pending[b] = __hip_atomic_load(gq[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    auto publish = [&](int b, float mine) {
        const int m = __float_as_int(fminf(bound[b], mine));
        if (q_ok[b] && m < g_seen[b]) {
            atomicMin(gq[b], m);
            g_seen[b] = m;
        }
    };
    auto take = [&]() {
#pragma unroll
        for (int b = 0; b < KF_QB; ++b) {
            g_seen[b] = min(g_seen[b], g_pending[b]);
            bound[b] = fminf(bound[b], __int_as_float(g_pending[b]));
        }
    };
#pragma unroll
    for (int b = 0; b < KF_QB; ++b) g_seen[b] = 0x7f7f7f7f;
    fetch();
#pragma unroll
    for (int b = 0; b < KF_QB; ++b) publish(b, INFINITY);
    take();
    // ---- the chunk: own running top-2 of upper bounds (distinct rows of this chunk), candidates against min(bound, U2)
    __shared__ unsigned stg_all[4][KF_STAGE];   // (row << 5) | query of the wave
    unsigned* const stg = stg_all[wave];
    int n_st = 0;                               // wave-uniform
    // Candidates are staged per wave in LDS (places from ballot prefix counts) and go to the queries' lists 64 at a time: an
    // append straight from the loop is an atomic whose result the store waits for, and with half the tiles holding a
    // candidate for one of the wave's queries that wait, not the arithmetic, was the kernel.
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < n_st; k += 64) {
            const unsigned v = stg[k];
            const int64_t qq = q_wave0 + (int64_t)(v & 31u);
            const int slot = atomicAdd(cnt + qq, 1);
            if (slot < KF_CAND_CAP) list[qq * KF_CAND_CAP + slot] = (int)(v >> 5);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        n_st = 0;
    };
    float U1[KF_QB], U2[KF_QB];
#pragma unroll
    for (int b = 0; b < KF_QB; ++b) U1[b] = U2[b] = INFINITY;
    const int64_t tile0 = (int64_t)blockIdx.y * tiles_per_chunk;
    TileOps cur, nxt;
    if (tile0 < ntiles) tile_load(tile0, cur);
    for (int ti = 0; ti < tiles_per_chunk; ++ti) {
        const int64_t tile = tile0 + ti;
        if (tile >= ntiles) break;
        const int64_t r0 = tile * 16;
        // the next tile's operands are asked for before this tile's arithmetic
        const bool more = ti + 1 < tiles_per_chunk && tile + 1 < ntiles;
        if (more) tile_load(tile + 1, nxt);
        const bool young = ti < KF_YOUNG || !KF_DEFER;
        const bool xch = ti < KF_YOUNG || (ti & KF_XCH_MASK) == 0;   // every tile while the bounds are young, then every fourth
        if (xch) {                                  // old chunks: what the previous exchange point asked for, and the next request;
            if (!young) take();                     // young ones cannot wait eight tiles for their first bounds: they ask here
            fetch();                                // and take at the bottom of the same tile
        }
#pragma unroll
        for (int b = 0; b < KF_QB; ++b) {
            float U[4], L[4];
            tile_bounds(cur, b, U, L);
#pragma unroll
            for (int i = 0; i < 4; ++i) top2(U1[b], U2[b], U[i]);
            if (xch) publish(b, U2[b]);
            const float bb = fminf(bound[b], U2[b]);
            const float thr = bb + bb * (1.f / 262144.f);
            bool any = false;
            bool emit[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                emit[i] = q_ok[b] && r0 + 4 * g + i < nt && L[i] <= thr;
                any = any || emit[i];
            }
            if (__builtin_amdgcn_ballot_w64(any) != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(emit[i]);
                    if (m == 0) continue;
                    if (emit[i])
                        stg[n_st + __popcll(m & ((1ull << lane) - 1ull))] = ((unsigned)(r0 + 4 * g + i) << 5) | (unsigned)(16 * b + c);
                    n_st += __popcll(m);
                }
                if (n_st > KF_STAGE - 256) flush();
            }
        }
        if (xch && young) take();
        if (more) cur = nxt;
    }
    flush();
}

// one wave per query: the oracle's chain on every candidate, (distance, index) top-2; a list that overflowed sets flag bit 1
__global__ void __launch_bounds__(256)
kf_refine_kernel(const float* __restrict__ q, int64_t nq, const float* __restrict__ t, int d, const int* __restrict__ cnt,
                 const int* __restrict__ list, int* __restrict__ idx, float* __restrict__ dist, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq) return;
    int n = cnt[qi];
    if (n > KF_CAND_CAP) {
        if (lane == 0) atomicOr(flag, 2);
        n = KF_CAND_CAP;
    }
    const float* qrow = q + qi * d;
    kf_top2 b;
    b.d1 = b.d2 = INFINITY; b.j1 = b.j2 = -1;
    for (int cix = lane; cix < n; cix += 64) {
        const int j = list[qi * KF_CAND_CAP + cix];
        const float* trow = t + (int64_t)j * d;
        float s = 0.f;
        if ((d & 3) == 0) {
            for (int k = 0; k < d; k += 4) {
                const float4 a = *(const float4*)(qrow + k), v = *(const float4*)(trow + k);
                float e = a.x - v.x; s += e * e;
                e = a.y - v.y; s += e * e;
                e = a.z - v.z; s += e * e;
                e = a.w - v.w; s += e * e;
            }
        } else {
            for (int k = 0; k < d; ++k) {
                const float e = qrow[k] - trow[k];
                s += e * e;
            }
        }
        kf_insert(b, sqrtf(s), j);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        kf_top2 ot;
        ot.d1 = __shfl_xor(b.d1, o); ot.d2 = __shfl_xor(b.d2, o);
        ot.j1 = __shfl_xor(b.j1, o); ot.j2 = __shfl_xor(b.j2, o);
        if (ot.j1 >= 0) kf_insert(b, ot.d1, ot.j1);
        if (ot.j2 >= 0) kf_insert(b, ot.d2, ot.j2);
    }
    if (lane == 0) {
        idx[2 * qi] = b.j1; idx[2 * qi + 1] = b.j2;
        dist[2 * qi] = b.d1; dist[2 * qi + 1] = b.d2;
    }
}

struct imatch_fset {
    imatch_ctx* ctx;
    float* desc;   // device [n][d], from the context's block pool
    int64_t n;
    int d;
    // the filter's view of the set (null when the set is too small for it to pay): f16 rows padded to dpad dimensions and to
    // a multiple of 16 rows, in the A operand's fragment order (kf_prep_kernel), |row|^2 and the row's error budget; usable when no value is beyond KF_MAXABS
    _Float16* desc_h;
    float *norm, *err;
    int dpad;
    bool filter_ok;
};

// q_host [nq][d] is uploaded, or the queries are already on the device (q_dev); the train rows are either uploaded too
// (t_host) or already resident (t_dev)
static int knn2_f32_filter(imatch_ctx* ctx, const float* q_host, const float* q_dev, int64_t nq, const imatch_fset* fs,
                           int32_t* idx, float* dist, bool* done);

static int knn2_f32_run(imatch_ctx* ctx, const float* q_host, int64_t nq, const float* t_host, const float* t_dev, int64_t nt, int d,
                        int32_t* idx, float* dist, const float* q_dev = nullptr, const imatch_fset* fs = nullptr) {
    if (nq > INT32_MAX / 2 || nt > INT32_MAX / 2) return imatch_set_err(ctx, IMATCH_ESIZE, "set too large");
    if (nq == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    static const int filter_env = getenv("IMATCH_KNNF_FILTER") ? atoi(getenv("IMATCH_KNNF_FILTER")) : 1;   // 0: A/B runs
    ctx->knnf_last_path = 0;
    if (fs && fs->filter_ok && filter_env && nq <= 16384 && nt < ((int64_t)1 << 27)) {   // (a staged candidate keeps its row in 27 bits)
        bool done = false;
        const int rc = knn2_f32_filter(ctx, q_host, q_dev, nq, fs, idx, dist, &done);
        ctx->knnf_last_path = done ? 1 : 2;
        if (rc || done) return rc;   // not done: a value out of the f16 path's range or a candidate list overflowed
    }
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

// *done = false (and rc 0) when the call has to take the exhaustive kernel after all
static int knn2_f32_filter(imatch_ctx* ctx, const float* q_host, const float* q_dev, int64_t nq, const imatch_fset* fs,
                           int32_t* idx, float* dist, bool* done) {
    *done = false;
    const int d = fs->d, dpad = fs->dpad;
    const int64_t nt = fs->n, nq_pad = (nq + 64 * KF_QB - 1) / (64 * KF_QB) * (64 * KF_QB);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_q = q_dev ? 0 : up((size_t)nq * d * 4), b_qh = up((size_t)nq_pad * dpad * 2), b_qn = up((size_t)nq_pad * 4);
    const size_t b_cnt = up((size_t)nq_pad * 4 + 4), b_list = up((size_t)nq * KF_CAND_CAP * 4), b_res = up((size_t)nq * 8);
    int rc = imatch_ws_reserve(ctx, b_q + b_qh + 3 * b_qn + b_cnt + b_list + 2 * b_res);
    if (rc) return rc;
    char* p = (char*)ctx->ws;
    float* q_up = (float*)p; p += b_q;
    _Float16* qh = (_Float16*)p; p += b_qh;
    float* qn = (float*)p; p += b_qn;
    float* qe = (float*)p; p += b_qn;
    int* cnt = (int*)p; p += b_cnt;           // [nq_pad] counters, then the flag word
    int* flag = cnt + nq_pad;
    int* gb = (int*)p; p += b_qn;             // [nq_pad] chip-wide bounds (float bits)
    int* list = (int*)p; p += b_list;
    int* idx_d = (int*)p; p += b_res;
    float* dist_d = (float*)p;
    rc = imatch_hs_reserve(ctx, b_q + 2 * b_res + 256);
    if (rc) return rc;
    const float* qd = q_dev;
    if (!q_dev) {
        memcpy(ctx->hs, q_host, (size_t)nq * d * 4);
        HIP_TRY(ctx, hipMemcpyAsync(q_up, ctx->hs, (size_t)nq * d * 4, hipMemcpyHostToDevice, ctx->stream));
        qd = q_up;
    }
    HIP_TRY(ctx, hipMemsetAsync(cnt, 0, b_cnt, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(gb, 0x7f, b_qn, ctx->stream));   // 0x7f7f7f7f = 3.4e38: no bound yet
    hipLaunchKernelGGL(kf_prep_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, ctx->stream, qd, nq, nq_pad, d, dpad, qh, qn, qe,
                       flag, 0);
    // enough workgroups for the chip (the loop is a chain of L2 round trips: it needs waves), chunks of at least 16 tiles (a chunk
    // spends four more on the seed)
    const int q_tiles = (int)(nq_pad / (64 * KF_QB));
    const int64_t ntiles = (nt + 15) / 16;
    static const int wg_target = getenv("IMATCH_KNNF_WGS") ? atoi(getenv("IMATCH_KNNF_WGS")) : 1024;
    static const int min_tiles = getenv("IMATCH_KNNF_MINTILES") ? atoi(getenv("IMATCH_KNNF_MINTILES")) : 16;
    int n_chunks = (int)std::min<int64_t>(std::max<int64_t>(1, (wg_target + q_tiles - 1) / q_tiles), std::max<int64_t>(1, ntiles / min_tiles));
    const int tiles_per_chunk = (int)((ntiles + n_chunks - 1) / n_chunks);
    n_chunks = (int)((ntiles + tiles_per_chunk - 1) / tiles_per_chunk);
#define KF_FILTER_ARGS dim3(q_tiles, n_chunks), dim3(256), 0, ctx->stream, (const _Float16*)qh, (const float*)qn, (const float*)qe, nq, \
                       (const _Float16*)fs->desc_h, (const float*)fs->norm, (const float*)fs->err, nt, dpad, tiles_per_chunk, gb, cnt, list
    switch (dpad >> 5) {
        case 1: hipLaunchKernelGGL(kf_filter_kernel<1>, KF_FILTER_ARGS); break;
        case 2: hipLaunchKernelGGL(kf_filter_kernel<2>, KF_FILTER_ARGS); break;
        case 3: hipLaunchKernelGGL(kf_filter_kernel<3>, KF_FILTER_ARGS); break;
        default: hipLaunchKernelGGL(kf_filter_kernel<4>, KF_FILTER_ARGS); break;
    }
#undef KF_FILTER_ARGS
    hipLaunchKernelGGL(kf_refine_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, ctx->stream, qd, nq, (const float*)fs->desc, d,
                       (const int*)cnt, (const int*)list, idx_d, dist_d, flag);
    HIP_TRY(ctx, hipGetLastError());
    char* hres = (char*)ctx->hs + b_q;
    HIP_TRY(ctx, hipMemcpyAsync(hres, idx_d, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hres + b_res, dist_d, (size_t)nq * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hres + 2 * b_res, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (getenv("IMATCH_KNNF_DEBUG")) {   // development: candidates per query
        std::vector<int> hc((size_t)nq);
        (void)hipMemcpy(hc.data(), cnt, (size_t)nq * 4, hipMemcpyDeviceToHost);
        long long tot = 0; int mx = 0;
        for (int v : hc) { tot += v; mx = std::max(mx, v); }
        fprintf(stderr, "[imatch] float kNN filter: %lld queries x %lld rows, %d chunks of %d tiles: %.1f candidates per query, at most %d\n",
                (long long)nq, (long long)nt, n_chunks, tiles_per_chunk, (double)tot / (double)nq, mx);
    }
    if (*(const int*)(hres + 2 * b_res) != 0) return IMATCH_OK;   // out-of-range query values (1) or a full candidate list (2)
    memcpy(idx, hres, (size_t)nq * 8);
    memcpy(dist, hres + b_res, (size_t)nq * 8);
    *done = true;
    return IMATCH_OK;
}

#define KF_FILTER_MIN_ROWS 4096

// the filter's view of a resident set (see kf_filter_kernel); leaves filter_ok false when the set does not qualify
static int fset_prepare_filter(imatch_ctx* ctx, imatch_fset* f) {
    f->filter_ok = false;
    if (f->n < KF_FILTER_MIN_ROWS) return IMATCH_OK;
    f->dpad = (f->d + 31) & ~31;
    const int64_t n_pad = (f->n + 15) & ~(int64_t)15;
    const size_t b_h = ((size_t)n_pad * f->dpad * 2 + 255) & ~(size_t)255, b_n = ((size_t)n_pad * 4 + 255) & ~(size_t)255;
    char* blk = (char*)imatch_pool_alloc(ctx, b_h + 2 * b_n + 256);
    if (!blk) return imatch_set_err(ctx, IMATCH_ENOMEM, "device block of %zu bytes", b_h + 2 * b_n + 256);
    f->desc_h = (_Float16*)blk;
    f->norm = (float*)(blk + b_h);
    f->err = (float*)(blk + b_h + b_n);
    int* flag = (int*)(blk + b_h + 2 * b_n);
    hipError_t e = hipMemsetAsync(flag, 0, 4, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(kf_prep_kernel, dim3((unsigned)((n_pad + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)f->desc, f->n, n_pad,
                           f->d, f->dpad, f->desc_h, f->norm, f->err, flag, 1);
        e = hipGetLastError();
    }
    int h_flag = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return imatch_set_err(ctx, IMATCH_EHIP, "filter view: %s", hipGetErrorString(e));
    f->filter_ok = h_flag == 0;
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
    const int rcf = fset_prepare_filter(ctx, f);
    if (rcf) {
        imatch_pool_free(ctx, f->desc_h);
        imatch_pool_free(ctx, f->desc);
        free(f);
        return rcf;
    }
    *out = f;
    return IMATCH_OK;
}

int64_t imatch_fset_count(const imatch_fset* f) { return f ? f->n : 0; }

void imatch_fset_free(imatch_fset* f) {
    if (!f) return;
    imatch_pool_free(f->ctx, f->desc_h);   // one block: f16 rows, norms, error budgets
    imatch_pool_free(f->ctx, f->desc);
    free(f);
}

int imatch_knn2_f32_debug_last_path(imatch_ctx* ctx) { return ctx ? ctx->knnf_last_path : 0; }

int imatch_knn2_fset_akaze(imatch_ctx* ctx, uint64_t serial, const imatch_fset* t, int32_t* idx, float* dist) {
    if (!ctx || !t || t->ctx != ctx || !idx || !dist) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    const float* qd = nullptr;
    int64_t nq = 0;
    if (imatch_akaze_last_dev(ctx, serial, &qd, &nq) != IMATCH_OK)
        return imatch_set_err(ctx, IMATCH_EINVAL, "AKAZE extraction %llu is not this context's last one: its descriptors are no "
                              "longer on the device (use imatch_knn2_fset with the host matrix)", (unsigned long long)serial);
    if (t->d != 64) return imatch_set_err(ctx, IMATCH_EINVAL, "train set has %d-float rows, AKAZE's descriptors have 64", t->d);
    return knn2_f32_run(ctx, nullptr, nq, nullptr, t->desc, t->n, t->d, idx, dist, qd, t);
}

int imatch_knn2_fset(imatch_ctx* ctx, const float* q, int64_t nq, const imatch_fset* t, int32_t* idx, float* dist) {
    if (!ctx || !t || t->ctx != ctx || nq < 0 || (nq > 0 && (!q || !idx || !dist)))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    return knn2_f32_run(ctx, q, nq, nullptr, t->desc, t->n, t->d, idx, dist, nullptr, t);
}

}  // extern "C"
