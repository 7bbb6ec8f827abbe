// GaussianBlur for the SIFT scale space on gfx950 (device code shared by sift.hip and tools/blur_bench.hip).
//
// blur_march_kernel: a workgroup owns a strip of W output columns and MARCHES down it in chunks of 32
// rows. Per chunk it (1) writes the 32 new input rows (prefetched into registers during the previous
// chunk's arithmetic) to LDS as interleaved row pairs, (2) runs the row pass on them -- two rows per
// v_pk_fma_f32 -- into a 64-row ring of row-pass results, (3) runs the column pass for the 32 output
// rows whose taps are now all in the ring. Compared with one tile per workgroup this never re-reads or
// recomputes the vertical halo (2R of 32+2R rows), keeps the next chunk's HBM loads in flight behind the
// current chunk's arithmetic, and all 256 threads have exactly one row-pass item and one column-pass item
// per chunk. Each output is still one ascending-tap fma chain per pass: bit-identical to the scalar form
// the CPU oracle computes (oracle/sift_oracle.c, GaussianBlur of sift.dispatch.cpp's pyramid).
//
// UP = true reads the u8 image and forms createInitialImage's x2 INTER_LINEAR value on the fly (the 2x
// f32 base is never stored). With u8 sources every product and sum of that interpolation is exact in
// f32, and clamped source indices reproduce OpenCV's border rule exactly, so any evaluation order gives
// the oracle's bits.
//
// dec != nullptr also writes the next octave's layer 0 (every second pixel of this output).
//
// The grid is one-dimensional and XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, so
// id -> (id % 8) * per + id / 8 makes the workgroups that are resident together on one XCD neighbouring
// strips, whose shared halo columns then hit that XCD's L2 instead of HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int R>
struct GK { float k[2 * R + 1]; };

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int reflect101(int p, int len) {
    if (p < 0) p = -p;                       // one fold each way covers every halo narrower than the image
    if (p >= len) p = 2 * (len - 1) - p;
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// value (dy, dx) of the x2 INTER_LINEAR upsample of a u8 image of urows x ucols
__device__ __forceinline__ float upsample2_value(const uint8_t* __restrict__ s, int urows, int ucols, int ustride, int dy,
                                                 int dx) {
    const int i = dx >> 1, j = dy >> 1;
    int xa, xb, ya, yb;
    float wxa, wya;
    if (dx & 1) { xa = i; xb = min(i + 1, ucols - 1); wxa = 0.75f; } else { xa = max(i - 1, 0); xb = i; wxa = 0.25f; }
    if (dy & 1) { ya = j; yb = min(j + 1, urows - 1); wya = 0.75f; } else { ya = max(j - 1, 0); yb = j; wya = 0.25f; }
    const uint8_t* s0 = s + (size_t)ya * ustride;
    const uint8_t* s1 = s + (size_t)yb * ustride;
    const float h0 = s0[xa] * wxa + s0[xb] * (1.f - wxa);
    const float h1 = s1[xa] * wxa + s1[xb] * (1.f - wxa);
    return h0 * wya + h1 * (1.f - wya);
}

#define BM_CH 32  // rows per chunk
// Workgroups per CU the kernel is compiled for. R = 13 at 4 (128 VGPRs) spills its row-pass window and runs
// 0.75 ms per 16384^2 layer against 0.61 ms at 3; R = 10 is the other way round (0.50 vs 0.56 ms).
#ifndef BM_MINB_BIG
#define BM_MINB_BIG 3
#endif
// Output rows are not read again by this launch: nontemporal stores keep them from displacing the halo
// columns neighbouring strips share through L2 (about 3 % per layer).
#ifndef BM_NT_STORE
#define BM_NT_STORE 1
#endif

// ABL (tools/blur_bench.hip only): 1 = row pass reduced to its centre tap, 2 = column pass likewise.
template <int R, int W, bool UP, int ABL = 0>
__global__ void __launch_bounds__(256, (W == 64 ? (R >= 12 ? BM_MINB_BIG : 4) : 2))
blur_march_kernel(const void* __restrict__ src_, float* __restrict__ dst, float* __restrict__ dec, int rows, int cols,
                  int ustride, GK<R> gk, int nt, int nstrip, int nseg, size_t src_zs, size_t zs) {
    static_assert(R >= 1 && R <= 16, "the 64-row ring holds 32 + 2R rows");
    static_assert(W == 64 || W == 128, "strip width");
    constexpr int N = 2 * R + 1, CH = BM_CH, NP = CH / 2;
    constexpr int XO = (R + 3) & ~3, P = W + 2 * XO, NX4 = P / 4;
    constexpr int NITEM = NP * NX4, NIT = (NITEM + 255) / 256;
    // `in` holds row pairs interleaved: element (row 2*yp + par, col) is float ((yp * PL + col) * 2 + par), so one
    // 16-byte read yields (y,x) (y+1,x) (y,x+1) (y+1,x+1). The pair pitch 2*PL is 4 mod 8 dwords: the 16-lane
    // groups of a ds_read_b128 then take their two row pairs from complementary 16-byte slots (conflict-free;
    // a pitch of 0 mod 8 is 2-way). The array is float4-typed so the reads are provably aligned (ds_read_b128,
    // not ds_read2_b64, which runs at half the rate and banks mod 32).
    constexpr int PL = P + 2;
    __shared__ float4 in4[NP * PL / 2];
    __shared__ float ring[64 * W];
    float* const in = (float*)in4;

    const int tid = threadIdx.x;
    int strip, seg;
    {
        const int total = nstrip * nseg, per = (total + 7) >> 3;
        const int id = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (id >= total) return;
        seg = id / nstrip;
        strip = id - seg * nstrip;
    }
    const float* srcf = UP ? nullptr : (const float*)((const char*)src_ + (size_t)blockIdx.z * src_zs);
    const uint8_t* srcu = UP ? (const uint8_t*)src_ + (size_t)blockIdx.z * src_zs : nullptr;
    dst = (float*)((char*)dst + (size_t)blockIdx.z * zs);
    if (dec) dec = (float*)((char*)dec + (size_t)blockIdx.z * zs);
    const int urows = rows >> 1, ucols = cols >> 1;   // UP: the u8 image
    const int x0 = strip * W, y_start = seg * CH * nt, y_end = min(rows, y_start + CH * nt);
    const int xl = x0 - XO;                          // image column of LDS column 0
    const bool colfast = xl >= 0 && xl + P <= cols && (UP || (cols & 3) == 0);

    // ---- chunk input: rows [yin, yin + 2 * np) -> registers (fast path) ...
    float4 pa[NIT], pb[NIT];
    // UP: the chunk's first row has the parity of R (y_start and the chunk height are even), so a row pair
    // needs u8 rows j-1, j, j+1 (even first row 2j) or j, j+1 (odd first row 2j+1); the raw bytes wait in
    // registers and are interpolated when they land.
    constexpr int UPR = (R & 1) ? 2 : 3;
    unsigned ub[NIT][UPR][4];
    auto issue = [&](int yin, int np) {
        if (UP) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = min(tid + it * 256, np * NX4 - 1);
                const int yp = item / NX4, x4 = item - yp * NX4;
                const int j = (yin + 2 * yp) >> 1, i0 = (xl + 4 * x4) >> 1;   // xl + 4*x4 is a multiple of 4
                const int c[4] = {max(i0 - 1, 0), i0, i0 + 1, min(i0 + 2, ucols - 1)};
#pragma unroll
                for (int q = 0; q < UPR; ++q) {
                    const int sr = (R & 1) ? min(j + q, urows - 1) : min(max(j - 1 + q, 0), urows - 1);
                    const uint8_t* sp = srcu + (size_t)sr * ustride;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) ub[it][q][cc] = sp[c[cc]];
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = min(tid + it * 256, np * NX4 - 1);
                const int yp = item / NX4, x4 = item - yp * NX4;
                const float* g = srcf + (size_t)(yin + 2 * yp) * cols + (xl + 4 * x4);
                pa[it] = *(const float4*)g;
                pb[it] = *(const float4*)(g + cols);
            }
        }
    };
    // ... -> LDS
    auto land = [&](int np) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (UP) {
                float h[UPR][4];
#pragma unroll
                for (int q = 0; q < UPR; ++q) {
                    const float s0 = (float)ub[it][q][0], s1 = (float)ub[it][q][1], s2 = (float)ub[it][q][2],
                                s3 = (float)ub[it][q][3];
                    h[q][0] = s0 * 0.25f + s1 * 0.75f;
                    h[q][1] = s1 * 0.75f + s2 * 0.25f;
                    h[q][2] = s1 * 0.25f + s2 * 0.75f;
                    h[q][3] = s2 * 0.75f + s3 * 0.25f;
                }
                float r0[4], r1[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    if (R & 1) {   // rows 2j+1, 2j+2
                        r0[cc] = h[0][cc] * 0.75f + h[1][cc] * 0.25f;
                        r1[cc] = h[0][cc] * 0.25f + h[1][cc] * 0.75f;
                    } else {       // rows 2j, 2j+1
                        r0[cc] = h[0][cc] * 0.25f + h[1][cc] * 0.75f;
                        r1[cc] = h[1][cc] * 0.75f + h[UPR - 1][cc] * 0.25f;
                    }
                }
                pa[it] = make_float4(r0[0], r0[1], r0[2], r0[3]);
                pb[it] = make_float4(r1[0], r1[1], r1[2], r1[3]);
            }
            const int item = tid + it * 256;
            if (item < np * NX4) {
                const int yp = item / NX4, x4 = item - yp * NX4;
                float4* l = &in4[yp * (PL / 2) + 2 * x4];
                l[0] = make_float4(pa[it].x, pb[it].x, pa[it].y, pb[it].y);
                l[1] = make_float4(pa[it].z, pb[it].z, pa[it].w, pb[it].w);
            }
        }
    };
    // border chunks and border strips: element-wise through reflect-101, loads batched before the LDS writes
    auto fill_slow = [&](int yin, int np) {
        constexpr int NE = (2 * NP * P + 255) / 256;
        float e[NE];
        const int ne = 2 * np * P;
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            const int f = min(tid + it * 256, ne - 1);
            const int yp = f / (2 * P), rem = f - yp * (2 * P);
            const int gy = reflect101(yin + 2 * yp + (rem & 1), rows), gx = reflect101(xl + (rem >> 1), cols);
            e[it] = UP ? upsample2_value(srcu, urows, ucols, ustride, gy, gx) : srcf[(size_t)gy * cols + gx];
        }
#pragma unroll
        for (int it = 0; it < NE; ++it) {
            const int f = tid + it * 256;
            if (f < ne) {
                const int yp = f / (2 * P), rem = f - yp * (2 * P);
                in[yp * (2 * PL) + rem] = e[it];
            }
        }
    };
    auto chunk_fast = [&](int yin, int np) { return colfast && yin >= 0 && yin + 2 * np <= rows; };

    // row pass over np row pairs of `in`; results go to ring rows (rel0 + 2*yp, +1) & 63
    auto row_pass = [&](int np, int rel0) {
        constexpr int E0 = (XO - R) & ~1, E1 = (XO + R + 4 + 1) & ~1, OFF = XO - R - E0;
        for (int item = tid; item < np * (W / 4); item += 256) {
            const int yp = item / (W / 4), xq = (item - yp * (W / 4)) * 4;
            f2 v[E1 - E0];
#pragma unroll
            for (int i = 0; i < (E1 - E0) / 2; ++i) {
                // the empty asm keeps the 16-byte read whole: when the first tap sits in the upper half (odd
                // XO - R) the compiler otherwise re-slices the window into misaligned ds_read2_b64 pairs
                f4 q = *(const f4*)&in4[yp * (PL / 2) + (xq + E0) / 2 + i];
                asm("" : "+v"(q));
                v[2 * i] = f2{q.x, q.y};
                v[2 * i + 1] = f2{q.z, q.w};
            }
            f2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
            for (int t = (ABL & 1) ? R : 0; t < ((ABL & 1) ? R + 1 : N); ++t) {
                const f2 k = {gk.k[t <= R ? t : 2 * R - t], gk.k[t <= R ? t : 2 * R - t]};   // symmetric: R + 1 distinct taps to hold
                a0 = __builtin_elementwise_fma(k, v[OFF + t], a0);
                a1 = __builtin_elementwise_fma(k, v[OFF + t + 1], a1);
                a2 = __builtin_elementwise_fma(k, v[OFF + t + 2], a2);
                a3 = __builtin_elementwise_fma(k, v[OFF + t + 3], a3);
            }
            const int rr = rel0 + 2 * yp;
            *(float4*)&ring[(rr & 63) * W + xq] = make_float4(a0.x, a1.x, a2.x, a3.x);
            *(float4*)&ring[((rr + 1) & 63) * W + xq] = make_float4(a0.y, a1.y, a2.y, a3.y);
        }
    };

    // ---- prologue: ring rows rel 0 .. 2R-1 = image rows y_start - R .. y_start + R - 1
    const int nchunk = (y_end - y_start + CH - 1) / CH;
    bool have = false;   // registers hold the next chunk's rows
    {
        const int yin = y_start - R;
        if (chunk_fast(yin, R)) { issue(yin, R); land(R); } else fill_slow(yin, R);
        // the first chunk's rows are asked for before the prologue's row pass, not after it: on a camera frame a workgroup
        // marches through ONE chunk, and the second round trip to memory was a fifth of the kernel
        if (chunk_fast(y_start + R, NP)) { issue(y_start + R, NP); have = true; }
        __syncthreads();
        row_pass(R, 0);
        __syncthreads();
    }
    for (int s = 0; s < nchunk; ++s) {
        const int yin = y_start + R + CH * s;
        // everyone is past the previous row pass (barrier below it), so `in` is free; the barrier after
        // this fill also orders the previous column pass's ring reads before the next row pass's ring writes
        if (have) land(NP); else fill_slow(yin, NP);
        have = false;
        if (s + 1 < nchunk && chunk_fast(yin + CH, NP)) { issue(yin + CH, NP); have = true; }
        __syncthreads();
        row_pass(NP, 2 * R + CH * s);
        __syncthreads();
        // column pass: thread = 2 adjacent columns x 4 output rows
        for (int item = tid; item < (W / 2) * (CH / 4); item += 256) {
            const int cp = item % (W / 2), r4 = (item / (W / 2)) * 4;
            f2 v[4 + 2 * R];
#pragma unroll
            for (int i = 0; i < 4 + 2 * R; ++i) v[i] = *(const f2*)&ring[((CH * s + r4 + i) & 63) * W + 2 * cp];
            f2 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f2{0.f, 0.f};
#pragma unroll
            for (int t = (ABL & 2) ? R : 0; t < ((ABL & 2) ? R + 1 : N); ++t) {
                const f2 k = {gk.k[t <= R ? t : 2 * R - t], gk.k[t <= R ? t : 2 * R - t]};   // symmetric: R + 1 distinct taps to hold
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_elementwise_fma(k, v[t + j], acc[j]);
            }
            const int gx = x0 + 2 * cp;
            const bool pair_ok = ((cols & 1) == 0) && gx + 1 < cols;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y_start + CH * s + r4 + j;
                if (gy >= y_end || gx >= cols) continue;
                const size_t o = (size_t)gy * cols + gx;
                if (pair_ok) {
                    if (BM_NT_STORE) __builtin_nontemporal_store(acc[j], (f2*)&dst[o]);
                    else *(f2*)&dst[o] = acc[j];
                } else {
                    dst[o] = acc[j].x;
                    if (gx + 1 < cols) dst[o + 1] = acc[j].y;
                }
                if (dec && !(j & 1) && (gy >> 1) < (rows >> 1) && (gx >> 1) < (cols >> 1))
                    dec[(size_t)(gy >> 1) * (cols >> 1) + (gx >> 1)] = acc[j].x;
            }
        }
    }
}

// Host launcher. nt = chunks of 32 rows a workgroup marches through (0: chosen here so that the launch
// still has several workgroups per CU slot).
template <int R, int W, bool UP, int ABL = 0>
static inline void blur_march_launch(hipStream_t st, const void* src, float* dst, float* dec, int rows, int cols, int ustride,
                                     const float* k, int batch, size_t src_zs, size_t zs, int nt = 0) {
    GK<R> gk;
    for (int i = 0; i < 2 * R + 1; ++i) gk.k[i] = k[i];
    const int nstrip = (cols + W - 1) / W, nchunk = (rows + BM_CH - 1) / BM_CH;
    if (nt <= 0) {
        // at least 16 workgroups per CU (256 CUs, 4 resident each) before the march gets longer; at most 16 chunks
        // (tools/blur_bench.hip: 8192^2 is 8 % faster at nt = 8 than at 16, 16384^2 the other way round)
        nt = 16;
        while (nt > 1 && (long long)nstrip * ((nchunk + nt - 1) / nt) * batch < 4096) nt >>= 1;
    }
    const int nseg = (nchunk + nt - 1) / nt, total = nstrip * nseg;
    hipLaunchKernelGGL((blur_march_kernel<R, W, UP, ABL>), dim3((unsigned)((total + 7) & ~7), 1, (unsigned)batch), dim3(256), 0, st,
                       src, dst, dec, rows, cols, ustride, gk, nt, nstrip, nseg, src_zs, zs);
}

// ---- three layers in one launch (camera frames) ----------------------------------------------------------------------
// On a 640x480 frame every blur launch is a few microseconds of arithmetic behind a launch boundary, and layers 1..nl of
// an octave are a dependent chain the next octave waits for. blur3_tile_kernel makes layers 1, 2 and 3 of an octave from
// layer 0 in ONE launch: a workgroup owns a 64 x 32 tile, loads it with a halo of R1 + R2 + R3 pixels (reflect-101
// applied to layer 0's coordinates), and runs the three separable blurs in LDS on shrinking regions -- layer 1 on the
// tile + (R2 + R3), layer 2 on the tile + R3, layer 3 on the tile. Halo pixels are recomputed by the neighbouring tiles
// (2 to 4 times the arithmetic of the marching kernel, on images where arithmetic is not what costs). Every in-image
// value is the same ascending-tap fma chain over the same inputs as in blur_march_kernel, so the layers are
// bit-identical; cells of an intermediate layer that lie OUTSIDE the image are overwritten with copies of their
// reflect-101 sources before the next blur reads them (recomputing them from mirrored inputs would visit the taps in
// mirrored order, which is not bit-identical).
#define B3_NT 512   // threads: two waves per SIMD, so that one wave's LDS round trips hide behind the other's arithmetic
// Tile shapes: 64 x 32 where the launch fills the chip (a batch, a map, a frame's first octave: the halo is recomputed 3.5
// times over), 32 x 16 where it does not (a frame's octaves 1-3: a few dozen to a few hundred tiles either way, and a
// workgroup with a quarter of the cells is done sooner: 17.4 / 16.7 / 16.2 -> 15.5 / 10.0 / 9.7 us for the three launches the
// next octave waits for; 32 x 8 and 16 x 16 are no shorter, profiles/r3_small_experiments.txt). Same cells, same chains.

template <int R>
__device__ __forceinline__ void b3_row_pass(const float* __restrict__ S, float* __restrict__ Tm, const GK<R>& gk, int r0, int r1,
                                            int c0, int c1, int P) {
    constexpr int N = 2 * R + 1, RA = (R + 3) & ~3, OFF = RA - R, NV = (OFF + N + 3 + 3) / 4;
    const int G = (c1 - c0) >> 2, items = (r1 - r0) * G;
    for (int item = threadIdx.x; item < items; item += B3_NT) {
        const int rr = item / G, g = item - rr * G;
        const int r = r0 + rr, c = c0 + 4 * g;
        const float4* src = (const float4*)(S + r * P + c - RA);
        float v[NV * 4];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float4 f = src[q];
            v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float kk = gk.k[t <= R ? t : 2 * R - t];
            a0 = fmaf(kk, v[OFF + t], a0);
            a1 = fmaf(kk, v[OFF + t + 1], a1);
            a2 = fmaf(kk, v[OFF + t + 2], a2);
            a3 = fmaf(kk, v[OFF + t + 3], a3);
        }
        *(float4*)(Tm + r * P + c) = make_float4(a0, a1, a2, a3);
    }
}

// D (in LDS) and, for cells of the workgroup's own tile inside the image, dst (+ dec) in memory
template <int R, int TW, int TH>
__device__ __forceinline__ void b3_col_pass(const float* __restrict__ Tm, float* __restrict__ D, const GK<R>& gk, int r0, int r1,
                                            int c0, int c1, int P, float* __restrict__ dst, float* __restrict__ dec, int Y0,
                                            int X0, int ty0, int tx0, int rows, int cols) {
    constexpr int N = 2 * R + 1;
    const int W = c1 - c0, items = ((r1 - r0) >> 2) * W;
    for (int item = threadIdx.x; item < items; item += B3_NT) {
        const int rg = item / W, cc = item - rg * W;
        const int r = r0 + 4 * rg, c = c0 + cc;
        const float* col = Tm + (r - R) * P + c;
        float v[N + 3];
#pragma unroll
        for (int j = 0; j < N + 3; ++j) v[j] = col[j * P];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float kk = gk.k[t <= R ? t : 2 * R - t];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = fmaf(kk, v[t + jj], acc[jj]);
        }
        const int gx = X0 + c;
        const bool mine_x = gx >= tx0 && gx < tx0 + TW && gx < cols;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            D[(r + jj) * P + c] = acc[jj];
            const int gy = Y0 + r + jj;
            if (mine_x && gy >= ty0 && gy < ty0 + TH && gy < rows) {
                dst[(size_t)gy * cols + gx] = acc[jj];
                if (dec && !(gy & 1) && !(gx & 1) && (gy >> 1) < (rows >> 1) && (gx >> 1) < (cols >> 1))
                    dec[(size_t)(gy >> 1) * (cols >> 1) + (gx >> 1)] = acc[jj];
            }
        }
    }
}

// cells of the region outside the image <- their reflect-101 sources (inside the image and inside the region)
// -- those within `reach` (the next blur's radius) of the image: nothing in-image reads the others, and their sources
// could lie outside the region
__device__ __forceinline__ void b3_mirror(float* __restrict__ D, int r0, int r1, int c0, int c1, int P, int Y0, int X0, int rows,
                                          int cols, int reach) {
    const int W = c1 - c0, n = (r1 - r0) * W;
    for (int p = threadIdx.x; p < n; p += B3_NT) {
        const int rr = p / W, cc = p - rr * W;
        const int r = r0 + rr, c = c0 + cc;
        const int gy = Y0 + r, gx = X0 + c;
        const bool outside = (unsigned)gy >= (unsigned)rows || (unsigned)gx >= (unsigned)cols;
        const bool near = gy >= -reach && gy < rows + reach && gx >= -reach && gx < cols + reach;
        if (outside && near) {
            const int sr = reflect101(gy, rows) - Y0, sc = reflect101(gx, cols) - X0;
            if (sr >= r0 && sr < r1 && sc >= c0 && sc < c1) D[r * P + c] = D[sr * P + sc];
        }
    }
}

template <int R1, int R2, int R3, int TW, int TH>
__global__ void __launch_bounds__(B3_NT)
blur3_tile_kernel(const float* __restrict__ src, float* __restrict__ dst1, float* __restrict__ dst2, float* __restrict__ dst3,
                  float* __restrict__ dec, int rows, int cols, GK<R1> k1, GK<R2> k2, GK<R3> k3, int tiles_x, size_t zs) {
    constexpr int RA1 = (R1 + 3) & ~3;
    constexpr int H1 = R2 + R3, H2 = R3;                     // halo of layers 1 and 2 around the tile
    constexpr int HX1 = (H1 + 3) & ~3, HX2 = (H2 + 3) & ~3;  // ... in x, rounded up to whole float4 groups
    constexpr int YB = H1 + R1, XB = HX1 + RA1;              // the tile's origin inside the buffers
    constexpr int RB = TH + 2 * YB, P = TW + 2 * XB;
    static_assert(((R2 + 3) & ~3) <= XB - HX2 && ((R3 + 3) & ~3) <= XB, "row-pass windows stay inside the buffer");
    static_assert((TH + 2 * H1) % 4 == 0 && (TH + 2 * H2) % 4 == 0, "column pass walks four rows at a time");
    __shared__ float4 sbuf4[RB * P / 4], tbuf4[RB * P / 4];
    float* const S = (float*)sbuf4;
    float* const Tm = (float*)tbuf4;
    src = (const float*)((const char*)src + (size_t)blockIdx.z * zs);
    dst1 = (float*)((char*)dst1 + (size_t)blockIdx.z * zs);
    dst2 = (float*)((char*)dst2 + (size_t)blockIdx.z * zs);
    dst3 = (float*)((char*)dst3 + (size_t)blockIdx.z * zs);
    if (dec) dec = (float*)((char*)dec + (size_t)blockIdx.z * zs);
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int ty0 = ty * TH, tx0 = tx * TW;
    const int Y0 = ty0 - YB, X0 = tx0 - XB;
    // layer 0 with its halo: every buffer cell, through reflect-101 (a halo wider than a small image folds repeatedly)
    // does any region poke out of the image? (workgroup-uniform)
    const bool border = Y0 < 0 || Y0 + RB > rows || X0 < 0 || X0 + P > cols;
    // (all of a round's loads are issued before its first LDS write: one load after the other was 30 of this kernel's 32 us)
    if (!border && (cols & 3) == 0) {
        constexpr int N4 = RB * P / 4, NL = (N4 + B3_NT - 1) / B3_NT;
        float4 v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int p4 = min((int)threadIdx.x + B3_NT * u, N4 - 1);
            const int r = p4 / (P / 4), c4 = p4 - r * (P / 4);
            v[u] = *(const float4*)(src + (size_t)(Y0 + r) * cols + (X0 + 4 * c4));
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const int p4 = (int)threadIdx.x + B3_NT * u;
            if (p4 < N4) sbuf4[p4] = v[u];
        }
    } else {
        for (int p0 = threadIdx.x; p0 < RB * P; p0 += B3_NT * 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = min(p0 + B3_NT * u, RB * P - 1);
                const int r = p / P, c = p - r * P;
                v[u] = src[(size_t)reflect101(Y0 + r, rows) * cols + reflect101(X0 + c, cols)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int p = p0 + B3_NT * u;
                if (p < RB * P) S[p] = v[u];
            }
        }
    }
    __syncthreads();
    b3_row_pass<R1>(S, Tm, k1, YB - H1 - R1, YB + TH + H1 + R1, XB - HX1, XB + TW + HX1, P);
    __syncthreads();
    b3_col_pass<R1, TW, TH>(Tm, S, k1, YB - H1, YB + TH + H1, XB - HX1, XB + TW + HX1, P, dst1, nullptr, Y0, X0, ty0, tx0, rows, cols);
    __syncthreads();
    if (ty0 - H1 < 0 || ty0 + TH + H1 > rows || tx0 - HX1 < 0 || tx0 + TW + HX1 > cols) {
        b3_mirror(S, YB - H1, YB + TH + H1, XB - HX1, XB + TW + HX1, P, Y0, X0, rows, cols, R2);
        __syncthreads();
    }
    b3_row_pass<R2>(S, Tm, k2, YB - H2 - R2, YB + TH + H2 + R2, XB - HX2, XB + TW + HX2, P);
    __syncthreads();
    b3_col_pass<R2, TW, TH>(Tm, S, k2, YB - H2, YB + TH + H2, XB - HX2, XB + TW + HX2, P, dst2, nullptr, Y0, X0, ty0, tx0, rows, cols);
    __syncthreads();
    if (ty0 - H2 < 0 || ty0 + TH + H2 > rows || tx0 - HX2 < 0 || tx0 + TW + HX2 > cols) {
        b3_mirror(S, YB - H2, YB + TH + H2, XB - HX2, XB + TW + HX2, P, Y0, X0, rows, cols, R3);
        __syncthreads();
    }
    b3_row_pass<R3>(S, Tm, k3, YB - R3, YB + TH + R3, XB, XB + TW, P);
    __syncthreads();
    b3_col_pass<R3, TW, TH>(Tm, S, k3, YB, YB + TH, XB, XB + TW, P, dst3, dec, Y0, X0, ty0, tx0, rows, cols);
}

template <int R1, int R2, int R3>
static inline void blur3_tile_launch(hipStream_t st, const float* src, float* dst1, float* dst2, float* dst3, float* dec, int rows,
                                     int cols, const float* k1, const float* k2, const float* k3, int batch, size_t zs) {
    GK<R1> g1;
    GK<R2> g2;
    GK<R3> g3;
    for (int i = 0; i < 2 * R1 + 1; ++i) g1.k[i] = k1[i];
    for (int i = 0; i < 2 * R2 + 1; ++i) g2.k[i] = k2[i];
    for (int i = 0; i < 2 * R3 + 1; ++i) g3.k[i] = k3[i];
    static const int small_env = getenv("IMATCH_BLUR3_SMALL") ? atoi(getenv("IMATCH_BLUR3_SMALL")) : 1;   // 0: 64 x 32 always
    const long long big_tiles = (long long)((cols + 63) / 64) * ((rows + 31) / 32) * batch;
    if (small_env && big_tiles < 512) {   // the launch would leave most of the chip idle: smaller, shorter workgroups
        const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 15) / 16;
        hipLaunchKernelGGL((blur3_tile_kernel<R1, R2, R3, 32, 16>), dim3((unsigned)(tiles_x * tiles_y), 1, (unsigned)batch),
                           dim3(B3_NT), 0, st, src, dst1, dst2, dst3, dec, rows, cols, g1, g2, g3, tiles_x, zs);
        return;
    }
    const int tiles_x = (cols + 63) / 64, tiles_y = (rows + 31) / 32;
    hipLaunchKernelGGL((blur3_tile_kernel<R1, R2, R3, 64, 32>), dim3((unsigned)(tiles_x * tiles_y), 1, (unsigned)batch), dim3(B3_NT),
                       0, st, src, dst1, dst2, dst3, dec, rows, cols, g1, g2, g3, tiles_x, zs);
}
