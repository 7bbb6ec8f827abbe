// SIFT detectAndCompute on gfx950: replaces the body of match_finder.find_kp_dp
// (reference match_finder.py:199-221 with the SIFT call of match_finder.py:205-209).
//
// Stages (all f32, the arithmetic order follows OpenCV's sift.simd.hpp so that the pyramid is
// bit-identical to the CPU oracle; -ffp-contract=off, every fma below is explicit):
//   blur<R>     separable Gaussian, blur.h: a workgroup marches down a 64-column strip, 32 rows at a
//               time, through LDS; the base blur forms the x2 bilinear upsample of the u8 image on the
//               fly (createInitialImage) and the blur of layer nOctaveLayers also writes the next
//               octave's layer 0 (every second pixel), so each Gaussian layer is read once and
//               written once and nothing else of the scale space is stored
//   detect      26-neighbour extremum scan; the DoG layers are never materialised: every consumer
//               (scan, refinement) forms gauss[i+1] - gauss[i] from the two Gaussian layers, the
//               same single f32 subtraction the oracle stores -> candidate list (atomic append)
//   refine      adjustLocalExtrema: <=5 Newton steps (3x3 Cramer solve), contrast and edge tests
//   orient      one wavefront per keypoint: 36-bin gradient histogram, smoothing, peaks
//   (host)      KeyPointsFilter::removeDuplicatedSorted order + first-octave rescale
//   describe    one wavefront per keypoint: 4x4x8 trilinear histogram, normalise, clamp, x512, u8
// Histograms accumulate in 64-bit fixed point (2^-20) through LDS atomics: integer adds commute,
// so results do not depend on the order in which lanes arrive.
#include "common.h"
#include "preproc.h"
#include "blur.h"
#include <algorithm>
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include <chrono>
static inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define MAX_OCT 16
#define MAX_LAY 8
#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define FIX_SCALE 1048576.0f
#define FIX_INV (1.0f / 1048576.0f)

struct PyrTable {
    float* gauss[MAX_OCT][MAX_LAY];
    int rows[MAX_OCT], cols[MAX_OCT];
    int n_oct, nl;
};

// Batched extraction: image b of a batch lives in the same arena layout shifted by b * zs bytes, and
// every kernel is launched with gridDim.z = batch size. ZP shifts a frame-0 pointer to this block's frame.
#define ZP(p) ((decltype(p))((char*)(p) + zo))
#define TG(o, i) ((float*)((char*)T.gauss[o][i] + zo))
typedef imatch_segtab SegTab;   // rows of image b: [off[b], off[b+1])

struct RefinedKp {   // output of refine, input of orient
    float x, y, size, response;
    int octave;      // packed: o + (layer << 8) + (cvRound((xi+0.5)*255) << 16)
    int o, layer, r, c;
};

struct KpRec {       // cv2.KeyPoint fields
    float x, y, size, angle, response;
    int octave;
};

struct imatch_sift_state {
    char* arena;
    size_t arena_bytes;
    PyrTable T;
    int h, w;
    int cand_hint, ref_hint, kp_hint;  // list lengths of the previous extraction: launch-grid and download-size guesses
    // Octave 1 needs only layer nl of octave 0, and an octave's six kernels depend on one another in a
    // chain: on large images octave 0's layers nl+1, nl+2 and scan (main stream) run beside ALL later octaves
    // (this second stream, forked after octave 0's layer nl, joined before the refinement). The later
    // octaves are a quarter of the bytes but mostly launch latency, which the overlap hides.
    // On a camera frame every launch is a few microseconds of latency and the chain is what costs: there the main
    // stream runs only what the NEXT octave waits for (layers 1..nl of every octave, then the small-octave kernel) and
    // the second stream runs each octave's layers nl+1, nl+2 and scan behind an event per octave (ev_oct).
    hipStream_t side;
    hipStream_t side2;           // chain-first order on large inputs: octaves >= 1's last two layers and scans (side has octave 0's)
    hipEvent_t ev_fork, ev_join, ev_join2;
    hipEvent_t ev_oct[MAX_OCT];
    bool have_ev_oct;
    // A batch's sorted records (7 MB for 64 frames) go to the host on a stream of their own, beside the descriptor kernel:
    // the host needs only the counters to launch it, the records for the keypoint table it fills afterwards.
    hipStream_t copy;
    hipEvent_t ev_sorted;
    // the one-workgroup-per-octave form of the small-octave kernel once timed out on this context: it is not used again
    bool no_tail_split;
    hipEvent_t ev_counts;   // recorded behind the count / record copies of a single frame (on the copy stream)
    hipEvent_t ev_image;    // ... and behind the copy of the equalised image that follows them there
};

void imatch_sift_state_free(imatch_ctx* ctx) {
    if (!ctx || !ctx->sift) return;
    if (ctx->sift->arena) (void)hipFree(ctx->sift->arena);
    if (ctx->sift->ev_counts) (void)hipEventDestroy(ctx->sift->ev_counts);
    if (ctx->sift->ev_image) (void)hipEventDestroy(ctx->sift->ev_image);
    if (ctx->sift->copy) {
        (void)hipStreamSynchronize(ctx->sift->copy);
        (void)hipEventDestroy(ctx->sift->ev_sorted);
        (void)hipStreamDestroy(ctx->sift->copy);
    }
    if (ctx->sift->side) {
        (void)hipStreamSynchronize(ctx->sift->side);
        (void)hipEventDestroy(ctx->sift->ev_fork);
        (void)hipEventDestroy(ctx->sift->ev_join);
        if (ctx->sift->have_ev_oct)
            for (int o = 0; o < MAX_OCT; ++o) (void)hipEventDestroy(ctx->sift->ev_oct[o]);
        (void)hipStreamDestroy(ctx->sift->side);
    }
    if (ctx->sift->side2) {
        (void)hipStreamSynchronize(ctx->sift->side2);
        (void)hipEventDestroy(ctx->sift->ev_join2);
        (void)hipStreamDestroy(ctx->sift->side2);
    }
    free(ctx->sift);
    ctx->sift = nullptr;
}

// ---- createInitialImage: u8 -> f32, x2 INTER_LINEAR ------------------------------------------
__global__ void __launch_bounds__(256)
upsample2_kernel(const uint8_t* __restrict__ src, int rows, int cols, int stride, float* __restrict__ dst, size_t src_zs,
                 size_t zs) {
    src += (size_t)blockIdx.z * src_zs;
    dst = (float*)((char*)dst + (size_t)blockIdx.z * zs);
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= 2 * cols || dy >= 2 * rows) return;
    float fx = (float)((dx + 0.5) * 0.5 - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= cols - 1) { fx = 0; sx = cols - 1; }
    float fy = (float)((dy + 0.5) * 0.5 - 0.5);
    int sy = (int)floorf(fy);
    fy -= sy;
    if (sy < 0) { fy = 0; sy = 0; }
    if (sy >= rows - 1) { fy = 0; sy = rows - 1; }
    const int sx1 = min(sx + 1, cols - 1), sy1 = min(sy + 1, rows - 1);
    const uint8_t* s0 = src + (size_t)sy * stride;
    const uint8_t* s1 = src + (size_t)sy1 * stride;
    const float a1 = fx, a0 = 1.f - a1;
    const float h0 = s0[sx] * a0 + s0[sx1] * a1;
    const float h1 = s1[sx] * a0 + s1[sx1] * a1;
    dst[(size_t)dy * (2 * cols) + dx] = h0 * (1.f - fy) + h1 * fy;
}

// ---- GaussianBlur ------------------------------------------------------------------------------------
// The templated radii run blur_march_kernel (blur.h).
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// any other radius (non-default sigma / nOctaveLayers): plain per-tap loops, same arithmetic order
#define RT_MAXR 20
struct GKrt { int r; float k[2 * RT_MAXR + 1]; };

__global__ void __launch_bounds__(256)
blur_rows_rt_kernel(const float* __restrict__ src, float* __restrict__ tmp, int rows, int cols, GKrt gk, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    src = ZP(src); tmp = ZP(tmp);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    const float* s = src + (size_t)y * cols;
    float acc = 0.f;
    for (int t = 0; t < 2 * gk.r + 1; ++t) acc = fmaf(gk.k[t], s[reflect101(x + t - gk.r, cols)], acc);
    tmp[(size_t)y * cols + x] = acc;
}

__global__ void __launch_bounds__(256)
blur_cols_rt_kernel(const float* __restrict__ tmp, float* __restrict__ dst, int rows, int cols, GKrt gk, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    tmp = ZP(tmp); dst = ZP(dst);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    float acc = 0.f;
    for (int t = 0; t < 2 * gk.r + 1; ++t) acc = fmaf(gk.k[t], tmp[(size_t)reflect101(y + t - gk.r, rows) * cols + x], acc);
    dst[(size_t)y * cols + x] = acc;
}

__global__ void __launch_bounds__(256)
decimate_kernel(const float* __restrict__ src, int scols, float* __restrict__ dst, int rows, int cols, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    src = ZP(src); dst = ZP(dst);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= cols || y >= rows) return;
    dst[(size_t)y * cols + x] = src[(size_t)(2 * y) * scols + 2 * x];
}

// ---- appending records to a global list ---------------------------------------------------------
// One atomicAdd on the list's single counter per candidate caps a kernel at ~100 appends/us for the
// whole chip (every op serialises in one L2 channel), which was 3.5 ms of the octave-0 scan. Waves
// therefore stage their records in LDS (slots from ballot prefix counts, no LDS atomics) and a
// workgroup reserves its range with ONE atomicAdd when it is done; a wave whose stage fills flushes
// on its own. The list order was never defined (the host sorts), so nothing downstream changes.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// wave-uniform n_staged > 0 or == 0; the whole wave calls
template <typename Rec>
__device__ __forceinline__ void stage_flush_wave(Rec* stage, int& n_staged, Rec* out, int cap, int* count) {
    if (n_staged == 0) return;
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(count, n_staged);
    base = __shfl(base, 0);
    wave_lds_fence();
    for (int j = lane; j < n_staged; j += 64)
        if ((unsigned)(base + j) < (unsigned)cap) out[base + j] = stage[j];   // (a poisoned counter is negative)
    wave_lds_fence();
    n_staged = 0;
}

template <typename Rec, int STAGE>
__device__ __forceinline__ void stage_append(Rec* stage, int& n_staged, bool hit, const Rec& rec, Rec* out, int cap,
                                             int* count) {
    const unsigned long long mk = __ballot(hit);
    if (mk == 0) return;  // wave-uniform
    const int lane = threadIdx.x & 63, np = __popcll(mk);
    if (n_staged + np > STAGE) stage_flush_wave(stage, n_staged, out, cap, count);
    if (hit) stage[n_staged + __popcll(mk & ((1ull << lane) - 1ull))] = rec;
    n_staged += np;
}

// Every wave of the NW-wave workgroup must call this, in converged control flow. wcnt: __shared__ int[NW + 1].
template <typename Rec, int NW = 4>
__device__ __forceinline__ void stage_flush_block(const Rec* stage, int n_staged, int* wcnt, Rec* out, int cap,
                                                  int* count) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wcnt[wv] = n_staged;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < NW; ++w) t += wcnt[w];
        wcnt[NW] = t ? atomicAdd(count, t) : 0;
    }
    __syncthreads();
    int base = wcnt[NW];
    for (int w = 0; w < wv; ++w) base += wcnt[w];
    for (int j = lane; j < n_staged; j += 64)
        if ((unsigned)(base + j) < (unsigned)cap) out[base + j] = stage[j];   // (a poisoned counter is negative)
}

// ---- findScaleSpaceExtrema: candidate scan ------------------------------------------------------
// Tile of 64x16 pixels per workgroup; the (nl+3) Gaussian layers of the tile with a 1-pixel halo are
// brought into LDS with asynchronous loads (each pixel is read from HBM once) and differenced in
// place into the nl+2 DoG layers, then every thread tests 4 rows x nl layers.
#define DT_W 64
#define DT_H 16
#define DT_PITCH 68
#define DT_LAYER ((DT_H + 2) * DT_PITCH)
#define DET_STAGE 128  // candidates a wave stages in LDS before it has to flush on its own
__global__ void __launch_bounds__(256)
detect_kernel_generic(PyrTable T, int o, float thr, uint2* __restrict__ cand, int cap, int* __restrict__ count, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    cand = ZP(cand); count = ZP(count);
    __shared__ float lds[MAX_LAY * DT_LAYER + 64];
    __shared__ uint2 stage_all[4][DET_STAGE];
    __shared__ int wcnt[5];
    const int cols = T.cols[o], rows = T.rows[o], nlay = T.nl + 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * DT_W + SIFT_IMG_BORDER, y0 = blockIdx.y * DT_H + SIFT_IMG_BORDER;
    const int npiece = (nlay * DT_LAYER + 63) / 64;
    for (int piece = wave; piece < npiece; piece += 4) {
        const int f = min(piece * 64 + lane, nlay * DT_LAYER - 1);
        const int l = f / DT_LAYER, rem = f - l * DT_LAYER;
        const int row = rem / DT_PITCH, col = rem - row * DT_PITCH;
        const int gy = min(max(y0 - 1 + row, 0), rows - 1), gx = min(max(x0 - 1 + col, 0), cols - 1);
        const float* g = TG(o, l) + (size_t)gy * cols + gx;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)g, (LDS_AS void*)(lds + piece * 64), 4, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int e = tid; e < DT_LAYER; e += 256) {   // a thread owns its elements through all layers
        float lo = lds[e];
        for (int l = 0; l + 1 < nlay; ++l) {
            const float hi = lds[(l + 1) * DT_LAYER + e];
            lds[l * DT_LAYER + e] = hi - lo;
            lo = hi;
        }
    }
    __syncthreads();
    const int c = x0 + lane;
    const bool col_ok = c < cols - SIFT_IMG_BORDER;
    int n_staged = 0;
    uint2* stage = stage_all[tid >> 6];
    for (int rr = 0; rr < 4; ++rr) {
        const int ly = 4 * (tid >> 6) + rr;  // tile row
        const int r = y0 + ly;
        if (r >= rows - SIFT_IMG_BORDER) break;  // wave-uniform
        for (int i = 1; i <= T.nl; ++i) {
            const float* img = lds + i * DT_LAYER + (ly + 1) * DT_PITCH + lane + 1;
            const float val = img[0];
            bool is_max = false, is_min = false;
            if (col_ok && fabsf(val) > thr) {
                const float* prev = img - DT_LAYER;
                const float* next = img + DT_LAYER;
                is_max = val > 0; is_min = val < 0;
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
                    for (int dc = -1; dc <= 1; ++dc) {
                        const int p = dr * DT_PITCH + dc;
                        const float a = img[p], b = prev[p], d = next[p];
                        is_max = is_max && val >= a && val >= b && val >= d;
                        is_min = is_min && val <= a && val <= b && val <= d;
                    }
                }
            }
            stage_append<uint2, DET_STAGE>(stage, n_staged, is_max || is_min,
                                           make_uint2((unsigned)((o << 8) | i), (unsigned)((r << 16) | c)), cand, cap, count);
        }
    }
    stage_flush_block(stage, n_staged, wcnt, cand, cap, count);
}

// nOctaveLayers = 3 (6 Gaussian -> 5 DoG layers), the default: a streaming scan with no LDS and no barrier. A wave
// owns a strip of 62 output columns (lanes 0 and 63 carry the halo columns) and walks `rpw` output rows
// (32 on a map: 6% halo re-reads; 8 on a camera frame, where the walk's serial load latency is the cost);
// per row and Gaussian layer it loads ONE value per lane (a coalesced 256-byte row segment), differences
// adjacent layers into the DoG values, gets the left and
// right neighbours from the adjacent lanes, and keeps a 3-row window of horizontal max/min in
// registers. Instead of 26 compares per pixel and layer, every layer's 3x3 max and min are built
// separably and shared by the three layers that need them: val is a maximum iff
// val >= max(M[i-1], M[i], M[i+1]), with M the 3x3 max that includes val itself.
#define DS_COLS 62
// The scans' grid is one-dimensional and XCD-aware like the blur's: consecutive workgroup ids go round-robin to the 8 XCDs,
// so id -> (id % 8) * per + id / 8 gives every XCD a contiguous run of (row group, strip) pairs. A wave's row segment starts
// at column 4 + 62 k (or 252 k, 126 k): not on a 128-byte line, so neighbouring strips share a line at each end; on one XCD
// the second strip finds it in that XCD's L2, where strips dealt round-robin fetched it once per XCD (FETCH_SIZE of the
// 4-byte-per-lane scan was 1.55 x the layers' bytes, calibrated with tools/fetch_calib.hip; profiles/r3_pmc.md).
__device__ __forceinline__ bool scan_block(int nsx, int nsy, int& sx, int& sy) {
    const int total = nsx * nsy, per = (total + 7) >> 3;
    const int id = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (id >= total || (int)(blockIdx.x >> 3) >= per) return false;
    sy = id / nsx;
    sx = id - sy * nsx;
    return true;
}

__global__ void __launch_bounds__(256)
detect_stream_kernel(PyrTable T, int o, int rpw, float thr, uint2* __restrict__ cand, int cap, int* __restrict__ count,
                     size_t zs, int nsx, int nsy) {
    int sx, sy;
    if (!scan_block(nsx, nsy, sx, sy)) return;
    const size_t zo = (size_t)blockIdx.z * zs;
    cand = ZP(cand); count = ZP(count);
    const int cols = T.cols[o], rows = T.rows[o];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = SIFT_IMG_BORDER + sx * DS_COLS + lane - 1;
    const int ry0 = SIFT_IMG_BORDER + (sy * 4 + wv) * rpw;
    __shared__ uint2 stage_all[4][DET_STAGE];
    __shared__ int wcnt[5];
    uint2* stage = stage_all[wv];
    int n_staged = 0;
    const bool wave_active = ry0 < rows - SIFT_IMG_BORDER;  // wave-uniform
    const int xc = min(max(x, 0), cols - 1);
    const float* p[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) p[l] = TG(o, l) + xc;
    const bool col_ok = lane >= 1 && lane <= DS_COLS && x < cols - SIFT_IMG_BORDER;
    float hm[5][3], hn[5][3], cv[3][3];
    // (Requesting row r + 2 before reducing row r + 1 was tried: 1.43 -> 1.58 ms on the 16384^2 octave; the walk
    // is not latency-bound at 8 waves per SIMD.)
    auto load_row = [&](int row, int slot) {
        const size_t off = (size_t)min(row, rows - 1) * cols;
        float g[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) g[l] = p[l][off];
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            const float v = g[l + 1] - g[l];
            const float a = __shfl_up(v, 1), d = __shfl_down(v, 1);
            hm[l][slot] = fmaxf(fmaxf(a, v), d);
            hn[l][slot] = fminf(fminf(a, v), d);
            if (l >= 1 && l <= 3) cv[l - 1][slot] = v;
        }
    };
    if (wave_active) {
        load_row(ry0 - 1, 0);
        load_row(ry0, 1);
    }
    for (int k = 0; wave_active && k < rpw; k += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = ry0 + k + u;            // output row: rows r-1, r, r+1 sit in the three slots
            load_row(r + 1, (u + 2) % 3);
            float M[5], m[5];
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                M[l] = fmaxf(fmaxf(hm[l][0], hm[l][1]), hm[l][2]);
                m[l] = fminf(fminf(hn[l][0], hn[l][1]), hn[l][2]);
            }
            const bool row_ok = col_ok && k + u < rpw && r < rows - SIFT_IMG_BORDER;
#pragma unroll
            for (int i = 1; i <= 3; ++i) {
                const float val = cv[i - 1][(u + 1) % 3];
                const bool is_max = val > thr && val >= fmaxf(fmaxf(M[i - 1], M[i]), M[i + 1]);
                const bool is_min = val < -thr && val <= fminf(fminf(m[i - 1], m[i]), m[i + 1]);
                stage_append<uint2, DET_STAGE>(stage, n_staged, row_ok && (is_max || is_min),
                                               make_uint2((unsigned)((o << 8) | i), (unsigned)((r << 16) | x)), cand, cap,
                                               count);
            }
        }
    }
    stage_flush_block(stage, n_staged, wcnt, cand, cap, count);
}

// The same scan with V adjacent columns per lane (V = 2 or 4: one 8- or 16-byte load per lane, row and layer). The scan is
// bound by instruction issue, not by bytes (about 150 instructions per wave and row for 62 pixels; profiles/): with V
// columns per lane the row's fixed costs -- address arithmetic, the exec-mask work around the rare appends, the loop -- and
// the lane exchanges (two per DoG layer and row, for the lane's outer columns only) are shared by V times the pixels. A wave
// loads 64 V columns starting at xb = 4 + (64 V - V) k (a multiple of V, so every load is aligned) and owns the outputs
// xb + 1 .. xb + 64 V - V; consecutive strips overlap by V columns. Needs cols % V == 0; same predicate, same records.
template <int V>
__global__ void __launch_bounds__(256)
detect_streamv_kernel(PyrTable T, int o, int rpw, float thr, uint2* __restrict__ cand, int cap, int* __restrict__ count,
                      size_t zs, int nsx, int nsy) {
    int sx, sy;
    if (!scan_block(nsx, nsy, sx, sy)) return;
    typedef float vf __attribute__((ext_vector_type(V)));
    constexpr int OUTW = 64 * V - V;
    const size_t zo = (size_t)blockIdx.z * zs;
    cand = ZP(cand); count = ZP(count);
    const int cols = T.cols[o], rows = T.rows[o];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int xb = (SIFT_IMG_BORDER - 1) + sx * OUTW;
    const int c0 = xb + V * lane;
    const int ry0 = SIFT_IMG_BORDER + (sy * 4 + wv) * rpw;
    __shared__ uint2 stage_all[4][DET_STAGE];
    __shared__ int wcnt[5];
    uint2* stage = stage_all[wv];
    int n_staged = 0;
    const bool wave_active = ry0 < rows - SIFT_IMG_BORDER;  // wave-uniform
    const int cl = min(c0, cols - V);                       // lanes past the row's end re-read its last group (never an output)
    const float* p[6];
#pragma unroll
    for (int l = 0; l < 6; ++l) p[l] = TG(o, l) + cl;
    bool col_ok[V];
#pragma unroll
    for (int e = 0; e < V; ++e) col_ok[e] = c0 + e > xb && c0 + e <= xb + OUTW && c0 + e < cols - SIFT_IMG_BORDER;
    float hm[5][3][V], hn[5][3][V], cv[3][3][V];
    auto load_row = [&](int row, int slot) {
        const size_t off = (size_t)min(row, rows - 1) * cols;
        vf g[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) g[l] = *(const vf*)(p[l] + off);
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            const vf v = g[l + 1] - g[l];
            const float left = __shfl_up(v[V - 1], 1), right = __shfl_down(v[0], 1);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float a = e > 0 ? v[e - 1] : left, d = e < V - 1 ? v[e + 1] : right;
                hm[l][slot][e] = fmaxf(fmaxf(a, v[e]), d);
                hn[l][slot][e] = fminf(fminf(a, v[e]), d);
                if (l >= 1 && l <= 3) cv[l - 1][slot][e] = v[e];
            }
        }
    };
    if (wave_active) {
        load_row(ry0 - 1, 0);
        load_row(ry0, 1);
    }
    for (int k = 0; wave_active && k < rpw; k += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int r = ry0 + k + u;            // output row: rows r-1, r, r+1 sit in the three slots
            load_row(r + 1, (u + 2) % 3);
            const bool row_ok = k + u < rpw && r < rows - SIFT_IMG_BORDER;
            bool hit[3][V];
            bool any = false;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float M[5], m[5];
#pragma unroll
                for (int l = 0; l < 5; ++l) {
                    M[l] = fmaxf(fmaxf(hm[l][0][e], hm[l][1][e]), hm[l][2][e]);
                    m[l] = fminf(fminf(hn[l][0][e], hn[l][1][e]), hn[l][2][e]);
                }
#pragma unroll
                for (int i = 1; i <= 3; ++i) {
                    const float val = cv[i - 1][(u + 1) % 3][e];
                    const bool is_max = val > thr && val >= fmaxf(fmaxf(M[i - 1], M[i]), M[i + 1]);
                    const bool is_min = val < -thr && val <= fminf(fminf(m[i - 1], m[i]), m[i + 1]);
                    hit[i - 1][e] = row_ok && col_ok[e] && (is_max || is_min);
                    any = any || hit[i - 1][e];
                }
            }
            if (__ballot(any) != 0) {   // wave-uniform; a few per cent of the rows
#pragma unroll
                for (int i = 1; i <= 3; ++i)
#pragma unroll
                    for (int e = 0; e < V; ++e)
                        stage_append<uint2, DET_STAGE>(stage, n_staged, hit[i - 1][e],
                                                       make_uint2((unsigned)((o << 8) | i), (unsigned)((r << 16) | (c0 + e))), cand,
                                                       cap, count);
            }
        }
    }
    stage_flush_block(stage, n_staged, wcnt, cand, cap, count);
}

// ---- the small octaves: one workgroup, LDS-resident ----------------------------------------------
// From the first octave of at most TAIL_MAXPX pixels on (60x80 for a 640x480 frame) a launch per layer
// is all latency: 7 launches per octave for a few thousand pixels. One 1024-thread workgroup instead
// walks those octaves start to finish: the current Gaussian layer, the row-pass intermediate and a
// ring of three DoG layers live in LDS; every Gaussian layer is also written to the global pyramid (refine,
// orientation and descriptors read it there). Same ascending-tap fma chains, same reflect-101, same
// extremum predicate as the per-layer kernels, so the result is bit-identical to them.
#define TAIL_MAXPX 4800
#define TAIL_STAGE 64
struct TailK { int r[MAX_LAY]; float k[MAX_LAY][2 * RT_MAXR + 1]; };  // slot i = blur that makes layer i
#define TAIL_LDS_BYTES ((6 * TAIL_MAXPX + TAIL_MAXPX / 4 + 64) * 4 + 16 * TAIL_STAGE * 8)

__global__ void __launch_bounds__(1024)
pyramid_tail_kernel(PyrTable T, int o_start, TailK gk, float thr, uint2* __restrict__ cand, int cap,
                    int* __restrict__ count, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    cand = ZP(cand); count = ZP(count);
    extern __shared__ float smem[];
    float* const g0 = smem;
    float* const tm = smem + 2 * TAIL_MAXPX;
    float* const dg = smem + 3 * TAIL_MAXPX;                // ring of 3 DoG layers
    float* const nx = smem + 6 * TAIL_MAXPX;                // next octave's layer 0
    float* const kt = nx + TAIL_MAXPX / 4;                  // taps of the current blur
    uint2* const stage = (uint2*)(kt + 64) + (threadIdx.x >> 6) * TAIL_STAGE;
    const int tid = threadIdx.x, nl = T.nl;
    int n_staged = 0;
    for (int o = o_start; o < T.n_oct; ++o) {
        const int rows = T.rows[o], cols = T.cols[o], n = rows * cols;
        if (o == o_start) {
            const float* up = TG(o - 1, nl);
            const int ucols = T.cols[o - 1];
            for (int p = tid; p < n; p += 1024) {
                const int y = p / cols, x = p - y * cols;
                const float v = up[(size_t)(2 * y) * ucols + 2 * x];
                g0[p] = v;
                TG(o, 0)[p] = v;
            }
        } else {
            for (int p = tid; p < n; p += 1024) {
                const float v = nx[p];
                g0[p] = v;
                TG(o, 0)[p] = v;
            }
        }
        __syncthreads();
        int cur = 0;
        const int ncols2 = cols >> 1, nrows2 = rows >> 1;
        const bool has_next = o + 1 < T.n_oct;
        for (int i = 1; i < nl + 3; ++i) {
            const int R = gk.r[i], N = 2 * R + 1;
            if (tid < N) kt[tid] = gk.k[i][tid];
            __syncthreads();
            const float* src = g0 + cur * TAIL_MAXPX;
            float* dst = g0 + (cur ^ 1) * TAIL_MAXPX;
            for (int p = tid; p < n; p += 1024) {
                const int y = p / cols, x = p - y * cols;
                const float* srow = src + y * cols;
                float acc = 0.f;
                for (int t = 0; t < N; ++t) acc = fmaf(kt[t], srow[reflect101(x + t - R, cols)], acc);
                tm[p] = acc;
            }
            __syncthreads();  // also: every detect below has finished reading the ring slot rewritten next
            float* dslot = dg + ((i - 1) % 3) * TAIL_MAXPX;
            float* gout = TG(o, i);
            for (int p = tid; p < n; p += 1024) {
                const int y = p / cols, x = p - y * cols;
                float acc = 0.f;
                for (int t = 0; t < N; ++t) acc = fmaf(kt[t], tm[reflect101(y + t - R, rows) * cols + x], acc);
                dst[p] = acc;
                gout[p] = acc;
                const float dv = acc - src[p];
                dslot[p] = dv;
                if (i == nl && has_next && !(y & 1) && !(x & 1) && (y >> 1) < nrows2 && (x >> 1) < ncols2)
                    nx[(y >> 1) * ncols2 + (x >> 1)] = acc;
            }
            __syncthreads();
            cur ^= 1;
            const int j = i - 2;  // DoG layers j-1, j, j+1 are in the ring: scan layer j
            if (j >= 1 && rows > 2 * SIFT_IMG_BORDER && cols > 2 * SIFT_IMG_BORDER) {
                const int iw = cols - 2 * SIFT_IMG_BORDER, ih = rows - 2 * SIFT_IMG_BORDER, ni = iw * ih;
                const float* lp = dg + ((j - 1) % 3) * TAIL_MAXPX;
                const float* lc = dg + (j % 3) * TAIL_MAXPX;
                const float* ln = dg + ((j + 1) % 3) * TAIL_MAXPX;
                for (int base = 0; base < ni; base += 1024) {  // workgroup-uniform trip count
                    const int p = base + tid;
                    bool hit = false;
                    int r = 0, c = 0;
                    if (p < ni) {
                        r = SIFT_IMG_BORDER + p / iw;
                        c = SIFT_IMG_BORDER + p % iw;
                        const int q = r * cols + c;
                        const float val = lc[q];
                        if (fabsf(val) > thr) {
                            bool is_max = val > 0, is_min = val < 0;
#pragma unroll
                            for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
                                for (int dc = -1; dc <= 1; ++dc) {
                                    const int qq = q + dr * cols + dc;
                                    const float a = lc[qq], b = lp[qq], d = ln[qq];
                                    is_max = is_max && val >= a && val >= b && val >= d;
                                    is_min = is_min && val <= a && val <= b && val <= d;
                                }
                            }
                            hit = is_max || is_min;
                        }
                    }
                    stage_append<uint2, TAIL_STAGE>(stage, n_staged, hit,
                                                    make_uint2((unsigned)((o << 8) | j), (unsigned)((r << 16) | c)), cand, cap,
                                                    count);
                }
            }
        }
        __syncthreads();  // nx and the ring are rewritten by the next octave
    }
    stage_flush_wave(stage, n_staged, cand, cap, count);
}

// ---- the small octaves, default kernel radii: the same walk, restructured for issue rate -----------
// pyramid_tail_kernel above spends ~15 instructions per tap (run-time radius, reflect-101 and a tap
// fetch inside the loop): 237 us for the four small octaves of a 640x480 frame, more than the four
// large octaves together. With the radii as template parameters the layer sits in LDS with a
// mirrored margin of TF_M pixels on every side, so no pass reflects; the row pass makes 4 adjacent
// outputs from aligned 16-byte LDS reads, the column pass 4 stacked outputs from one column walk,
// all taps unrolled from SGPRs. After each layer the margin is refilled with COPIES of the interior
// (recomputing a mirrored pixel would visit the taps in mirrored order, which is not bit-identical).
#define TF_M 13                      // margin rows/columns = largest radius
#define TF_XO 16                     // column of x = 0 inside a row (>= TF_M, multiple of 4)
#define TF_EXT 10560                 // floats per margin-extended buffer (60x80 needs 90 x 116, 80x60 110 x 96)
#define TF_LDS_BYTES ((2 * TF_EXT + 3 * TAIL_MAXPX + TAIL_MAXPX / 4) * 4 + 16 * TAIL_STAGE * 8 + 17 * 4)
struct TailFK { float k[5][2 * TF_M + 1]; };

__host__ __device__ inline int tf_pitch(int cols) { return (cols + TF_XO + TF_M + 4 + 3) & ~3; }
__host__ __device__ inline int tf_ext_rows(int rows) { return rows + 2 * TF_M + 4; }

template <int R>
__device__ __forceinline__ void tf_row_pass(const float* __restrict__ A, float* __restrict__ tm, const float* k, int rows,
                                            int cols, int P) {
    constexpr int N = 2 * R + 1, RA = (R + 3) & ~3, OFF = RA - R, NV = (OFF + N + 3 + 3) / 4;
    const int G = (cols + 3) >> 2, items = (rows + 2 * R) * G;
    for (int item = threadIdx.x; item < items; item += 1024) {
        const int yy = item / G, g = item - yy * G;
        const int row = yy - R + TF_M;                      // buffer row of image row yy - R
        const float4* src = (const float4*)(A + row * P + TF_XO + 4 * g - RA);
        float v[NV * 4];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float4 f = src[q];
            v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w;
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float kk = k[t];
            a0 = fmaf(kk, v[OFF + t], a0);
            a1 = fmaf(kk, v[OFF + t + 1], a1);
            a2 = fmaf(kk, v[OFF + t + 2], a2);
            a3 = fmaf(kk, v[OFF + t + 3], a3);
        }
        *(float4*)(tm + row * P + 4 * g) = make_float4(a0, a1, a2, a3);  // tm columns start at x = 0
    }
}

template <int R>
__device__ __forceinline__ void tf_col_pass(float* __restrict__ A, const float* __restrict__ tm, const float* k, int rows,
                                            int cols, int P, float* __restrict__ ring, float* __restrict__ gout,
                                            float* __restrict__ nx) {
    constexpr int N = 2 * R + 1;
    const int items = ((rows + 3) >> 2) * cols;
    const int ncols2 = cols >> 1, nrows2 = rows >> 1;
    for (int item = threadIdx.x; item < items; item += 1024) {
        const int yg = item / cols, x = item - yg * cols;
        const int y0 = 4 * yg;
        const float* col = tm + (y0 - R + TF_M) * P + x;
        float v[N + 3];
#pragma unroll
        for (int j = 0; j < N + 3; ++j) v[j] = col[j * P];
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < N; ++t) {
            const float kk = k[t];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[jj] = fmaf(kk, v[t + jj], acc[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int y = y0 + jj;
            if (y >= rows) break;
            float* a = A + (y + TF_M) * P + TF_XO + x;
            const float dv = acc[jj] - *a;
            *a = acc[jj];
            const int p = y * cols + x;
            ring[p] = dv;
            gout[p] = acc[jj];
            if (nx && !(y & 1) && !(x & 1) && (y >> 1) < nrows2 && (x >> 1) < ncols2) nx[(y >> 1) * ncols2 + (x >> 1)] = acc[jj];
        }
    }
}

// margin <- copies of the interior through reflect-101, M cells deep (the next layer's radius). Two index spaces with
// cheap arithmetic -- the side margins of the interior rows, then the bands above and below with their corners -- where
// one loop over the whole extended rectangle spent 3 us per layer of a 60 x 80 octave on divisions.
template <int M>
__device__ __forceinline__ void tf_mirror(float* __restrict__ A, int rows, int cols, int P) {
    for (int p = threadIdx.x; p < rows * (2 * M); p += 1024) {
        const int y = p / (2 * M), k = p - y * (2 * M);          // constant divisor
        const int x = k < M ? -1 - k : cols + (k - M);
        float* row = A + (y + TF_M) * P + TF_XO;
        row[x] = row[reflect101(x, cols)];
    }
    const int ew = cols + 2 * M;                                 // mostly <= 128: one trip
    for (int xx = threadIdx.x & 127; xx < ew; xx += 128) {
        const int x = xx - M, sx = reflect101(x, cols);
        for (int b = threadIdx.x >> 7; b < 2 * M; b += 8) {
            const int y = b < M ? -1 - b : rows + (b - M);
            A[(y + TF_M) * P + TF_XO + x] = A[(reflect101(y, rows) + TF_M) * P + TF_XO + sx];
        }
    }
}

#ifdef TAIL_TRACE   // development only (tools/dbg/tail_trace.py)
__device__ long long tail_trace[256];
__device__ int tail_trace_n;
#define TT() do { if (threadIdx.x == 0 && blockIdx.z == 0 && tt_n < 256) tail_trace[tt_n++] = (long long)wall_clock64(); } while (0)
extern "C" int imatch_dbg_tail_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tail_trace), sizeof(long long) * 256);
}
#else
#define TT() do {} while (0)
#endif
// SPLIT: one workgroup per octave (blockIdx.x), for single frames, where this kernel is a third of the scale-space chain:
// octave o + 1 needs only layer nl of octave o, so its workgroup starts as soon as that layer is in the global pyramid
// (a flag per octave in the image's counter block, release / acquire at agent scope) while octave o's workgroup goes on
// with its last two layers and scans. No dispatch order is assumed: the launcher uses this form only for grids of at most
// 32 workgroups (one CU each, of 256), so every workgroup becomes resident whatever the waiting ones do -- kernels of other
// streams that hold CUs meanwhile finish without any help from this one. A batch keeps the one-workgroup walk: there the
// waiting workgroups would hold CUs that other images' kernels can use, and the grid would no longer fit the chip.
template <int R1, int R2, int R3, int R4, int R5, bool SPLIT>
__global__ void __launch_bounds__(1024)
pyramid_tail_fast_kernel(PyrTable T, int o_start, TailFK gk, float thr, uint2* __restrict__ cand, int cap,
                         int* __restrict__ count, size_t zs, int withhold) {
    const size_t zo = (size_t)blockIdx.z * zs;
    cand = ZP(cand); count = ZP(count);
    static_assert(R1 <= TF_M && R2 <= TF_M && R3 <= TF_M && R4 <= TF_M && R5 <= TF_M, "margin too small");
    extern __shared__ float smem[];
    float* const A = smem;
    float* const tm = smem + TF_EXT;
    float* const dg = smem + 2 * TF_EXT;                    // ring of 3 DoG layers, [rows][cols]
    float* const nx = dg + 3 * TAIL_MAXPX;                  // next octave's layer 0
    uint2* const stage = (uint2*)(nx + TAIL_MAXPX / 4) + (threadIdx.x >> 6) * TAIL_STAGE;
    int* const wcnt = (int*)((uint2*)(nx + TAIL_MAXPX / 4) + 16 * TAIL_STAGE);
    const int tid = threadIdx.x;
    int n_staged = 0;
    int tt_n = 0; (void)tt_n;
    TT();
    int* const ready = count + 16;        // [octave]: layer nl of that octave is in the global pyramid (SPLIT)
    const int o_first = SPLIT ? o_start + (int)blockIdx.x : o_start, o_last = SPLIT ? o_first + 1 : T.n_oct;
    for (int o = o_first; o < o_last; ++o) {
        const int rows = T.rows[o], cols = T.cols[o], n = rows * cols, P = tf_pitch(cols);
        if (SPLIT && o > o_start) {
            if (tid == 0) {
                // bounded (about a second; the whole kernel takes 70 us when nothing holds it up): if the flag never comes
                // -- the workgroup before this one never ran -- the candidate counter's sign bit is set and the octave
                // recorded (count[4] = octave + 1), the kernels behind this one then find nothing to do, and the host
                // runs the extraction again with the one-workgroup walk, which waits for nobody
                int spins = 0;
                while (__hip_atomic_load(ready + (o - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > 1000000) {
                        atomicOr(count, (int)0x80000000);
                        atomicMax(count + 4, o + 1);
                        break;
                    }
                }
            }
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        {
            const float* up = TG(o - 1, T.nl);
            const int ucols = T.cols[o - 1];
            float* g0 = TG(o, 0);
            const bool from_global = SPLIT || o == o_start;
            float v[(TAIL_MAXPX + 1023) / 1024];   // all loads in flight before the first store
#pragma unroll
            for (int k = 0; k < (TAIL_MAXPX + 1023) / 1024; ++k) {
                const int p = tid + 1024 * k;
                if (p < n) {
                    const int y = p / cols, x = p - y * cols;
                    v[k] = from_global ? up[(size_t)(2 * y) * ucols + 2 * x] : nx[p];
                }
            }
#pragma unroll
            for (int k = 0; k < (TAIL_MAXPX + 1023) / 1024; ++k) {
                const int p = tid + 1024 * k;
                if (p < n) {
                    const int y = p / cols, x = p - y * cols;
                    A[(y + TF_M) * P + TF_XO + x] = v[k];
                    g0[p] = v[k];
                }
            }
        }
        __syncthreads();
        TT();
        tf_mirror<R1>(A, rows, cols, P);
        __syncthreads();
        TT();
        float* const nxo = (!SPLIT && o + 1 < T.n_oct) ? nx : nullptr;
        const bool scan_ok = rows > 2 * SIFT_IMG_BORDER && cols > 2 * SIFT_IMG_BORDER;
        auto scan = [&](int j) {  // DoG layers j-1, j, j+1 sit in the ring
            if (!scan_ok) return;
            const int iw = cols - 2 * SIFT_IMG_BORDER, ih = rows - 2 * SIFT_IMG_BORDER, ni = iw * ih;
            const float* lp = dg + ((j - 1) % 3) * TAIL_MAXPX;
            const float* lc = dg + (j % 3) * TAIL_MAXPX;
            const float* ln = dg + ((j + 1) % 3) * TAIL_MAXPX;
            for (int base = 0; base < ni; base += 1024) {  // workgroup-uniform trip count
                const int p = base + tid;
                bool hit = false;
                int r = 0, c = 0;
                if (p < ni) {
                    r = SIFT_IMG_BORDER + p / iw;
                    c = SIFT_IMG_BORDER + p % iw;
                    const int q = r * cols + c;
                    const float val = lc[q];
                    if (fabsf(val) > thr) {
                        bool is_max = val > 0, is_min = val < 0;
#pragma unroll
                        for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
                            for (int dc = -1; dc <= 1; ++dc) {
                                const int qq = q + dr * cols + dc;
                                const float a = lc[qq], b = lp[qq], d = ln[qq];
                                is_max = is_max && val >= a && val >= b && val >= d;
                                is_min = is_min && val <= a && val <= b && val <= d;
                            }
                        }
                        hit = is_max || is_min;
                    }
                }
                stage_append<uint2, TAIL_STAGE>(stage, n_staged, hit,
                                                make_uint2((unsigned)((o << 8) | j), (unsigned)((r << 16) | c)), cand, cap,
                                                count);
            }
        };
#define TF_LAYER(I, R, RNEXT, NX)                                                                                  \
        tf_row_pass<R>(A, tm, gk.k[I - 1], rows, cols, P);                                                         \
        __syncthreads(); /* also: the scan before this is done with the ring slot rewritten next */              \
        TT();                                                                                                      \
        tf_col_pass<R>(A, tm, gk.k[I - 1], rows, cols, P, dg + ((I - 1) % 3) * TAIL_MAXPX, TG(o, I), NX);     \
        __syncthreads();                                                                                           \
        TT();                                                                                                      \
        if (I < 5) { tf_mirror<RNEXT>(A, rows, cols, P); __syncthreads(); TT(); }
        TF_LAYER(1, R1, R2, nullptr)
        TF_LAYER(2, R2, R3, nullptr)
        if (SPLIT && o + 1 < T.n_oct) {
            // layer nl for the next octave's workgroup: own stores out, everybody's (barrier), then the flag
            tf_row_pass<R3>(A, tm, gk.k[2], rows, cols, P);
            __syncthreads();
            tf_col_pass<R3>(A, tm, gk.k[2], rows, cols, P, dg + 2 * TAIL_MAXPX, TG(o, 3), nullptr);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __syncthreads();
            // (withhold: the test knob IMATCH_DEBUG_TAIL_WITHHOLD -- this octave's flag is never published)
            if (tid == 0 && o != withhold) __hip_atomic_store(ready + o, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tf_mirror<R4>(A, rows, cols, P);
            __syncthreads();
        } else {
            TF_LAYER(3, R3, R4, nxo)
        }
        scan(1);
        TT();
        TF_LAYER(4, R4, R5, nullptr)
        scan(2);
        TT();
        TF_LAYER(5, R5, R5, nullptr)
        scan(3);
        TT();
#undef TF_LAYER
        __syncthreads();  // nx, A and the ring are rewritten by the next octave
    }
    stage_flush_block<uint2, 16>(stage, n_staged, wcnt, cand, cap, count);
}

// ---- adjustLocalExtrema ---------------------------------------------------------------------------
__device__ __forceinline__ void solve3(const float a[3][3], const float b[3], float x[3]) {
    float d = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
              a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (d == 0) { x[0] = x[1] = x[2] = 0; return; }
    d = 1 / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

__device__ __forceinline__ bool refine_one(const PyrTable& T, const size_t zo, const uint2 cd, float contrast_thr,
                                           float edge_thr, float sigma, RefinedKp& k) {
    const int octv = cd.x >> 8;
    int layer = cd.x & 255, r = cd.y >> 16, c = cd.y & 0xffff;
    const int nl = T.nl, rows = T.rows[octv], cols = T.cols[octv];
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale,
                cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    // DoG layer l = gauss[l+1] - gauss[l], formed here; img/prev/next name DoG layers by their index
#define AT(l, rr, cc) (TG(octv, (l) + 1)[(size_t)(rr) * cols + (cc)] - TG(octv, (l))[(size_t)(rr) * cols + (cc)])
    for (; i < SIFT_MAX_INTERP_STEPS; i++) {
        const int img = layer, prev = layer - 1, next = layer + 1;
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
        if (fabsf(xi) > (float)(INT_MAX / 3) || fabsf(xr) > (float)(INT_MAX / 3) || fabsf(xc) > (float)(INT_MAX / 3)) return false;
        c += __float2int_rn(xc); r += __float2int_rn(xr); layer += __float2int_rn(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= cols - SIFT_IMG_BORDER || r < SIFT_IMG_BORDER ||
            r >= rows - SIFT_IMG_BORDER)
            return false;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return false;
    float contr;
    {
        const int img = layer, prev = layer - 1, next = layer + 1;
        const float dD[3] = {(AT(img, r, c + 1) - AT(img, r, c - 1)) * deriv_scale,
                             (AT(img, r + 1, c) - AT(img, r - 1, c)) * deriv_scale,
                             (AT(next, r, c) - AT(prev, r, c)) * deriv_scale};
        const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
        contr = AT(img, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * nl < contrast_thr) return false;
        const float v2 = AT(img, r, c) * 2.f;
        const float dxx = (AT(img, r, c + 1) + AT(img, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (AT(img, r + 1, c) + AT(img, r - 1, c) - v2) * second_deriv_scale;
        const float dxy = (AT(img, r + 1, c + 1) - AT(img, r + 1, c - 1) - AT(img, r - 1, c + 1) + AT(img, r - 1, c - 1)) * cross_deriv_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return false;
    }
#undef AT
    k.x = (c + xc) * (1 << octv);
    k.y = (r + xr) * (1 << octv);
    k.octave = octv + (layer << 8) + ((int)rint(((double)xi + 0.5) * 255) << 16);
    k.size = sigma * powf(2.f, (layer + xi) / nl) * (1 << octv) * 2;
    k.response = fabsf(contr);
    k.o = octv; k.layer = layer; k.r = r; k.c = c;
    return true;
}

// The candidate count is read from device memory (the scan that produced it is still in flight when the
// host queues this kernel), so the grid is sized by a guess and strides.
__global__ void __launch_bounds__(256)
refine_kernel(PyrTable T, const uint2* __restrict__ cand, const int* __restrict__ n_cand_dev, int cap, float contrast_thr,
              float edge_thr, float sigma, RefinedKp* __restrict__ out, int* __restrict__ count, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    cand = ZP(cand); n_cand_dev = ZP(n_cand_dev); out = ZP(out); count = ZP(count);
    const int n_cand = min(*n_cand_dev, cap);
    for (int id = blockIdx.x * 256 + threadIdx.x; id < n_cand; id += gridDim.x * 256) {
        RefinedKp k;
        if (refine_one(T, zo, cand[id], contrast_thr, edge_thr, sigma, k)) out[atomicAdd(count, 1)] = k;
    }
}

// The descriptor stage's own atan2 / exp / sqrt: its samples only feed interpolation WEIGHTS (a last-place difference moves
// a weight by 1e-7 of itself), so the library functions' special-case handling and correctly rounded square root -- 70 of
// the 190 instructions of one sample -- are not needed there. Same minimax polynomial as the device library's atan2f
// (|error| < 2e-7 rad), reciprocal instead of a scaled division; finite inputs only (pixel differences).
__device__ __forceinline__ float atan2_deg_lean(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mx > 0.f ? mn * __builtin_amdgcn_rcpf(mx) : 0.f;
    const float u = t * t;
    float p = fmaf(u, 0.002642294391989708f, -0.015280019491910934f);
    p = fmaf(u, p, 0.04149937257170677f);
    p = fmaf(u, p, -0.07413642853498459f);
    p = fmaf(u, p, 0.10605181753635406f);
    p = fmaf(u, p, -0.14197120070457458f);
    p = fmaf(u, p, 0.19992350041866302f);
    p = fmaf(u, p, -0.3333311676979065f);
    float a = fmaf(t, u * p, t);
    a = ay > ax ? 1.5707963267948966f - a : a;
    a = x < 0.f ? 3.141592653589793f - a : a;
    a *= (float)(180.0 / M_PI);
    return y < 0.f ? 360.f - a : a;
}

__device__ __forceinline__ float atan2_deg(float y, float x) {
    float a = atan2f(y, x) * (float)(180.0 / M_PI);
    if (a < 0) a += 360.f;
    return a;
}

// ---- calcOrientationHist + peak picking: one wavefront per refined keypoint ---------------------
// The list length is read from device memory and the grid strides over it; a wave stages its output
// records, so the output counter sees one atomicAdd per workgroup.
#define ORI_STAGE 64
__global__ void __launch_bounds__(256)
orient_kernel(PyrTable T, const RefinedKp* __restrict__ ref, const int* __restrict__ n_ref_dev, int ref_cap,
              KpRec* __restrict__ out, int cap, int* __restrict__ count, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    ref = ZP(ref); n_ref_dev = ZP(n_ref_dev); out = ZP(out); count = ZP(count);
    __shared__ unsigned long long hacc[4][SIFT_ORI_HIST_BINS];
    __shared__ float hsm[4][SIFT_ORI_HIST_BINS + 4];
    __shared__ KpRec stage_all[4][ORI_STAGE];
    __shared__ int wcnt[5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = SIFT_ORI_HIST_BINS;
    KpRec* stage = stage_all[wave];
    int n_staged = 0;
    const int n_ref = min(*n_ref_dev, ref_cap);
    // Every wave walks its own keypoints: its histogram and staging rows are its own, so the phases below are ordered by
    // wave-level fences, not workgroup barriers (seven of them per keypoint, with the four waves' window sizes differing,
    // were a third of this kernel).
    for (int id = blockIdx.x * 4 + wave; id < n_ref; id += gridDim.x * 4) {  // wave-uniform trip count
        const bool active = true;
        if (lane < n) hacc[wave][lane] = 0ull;
        wave_lds_fence();
        RefinedKp k;
        if (active) {
            k = ref[id];
            const int rows = T.rows[k.o], cols = T.cols[k.o];
            const float* img = TG(k.o, k.layer);
            const float scl_octv = k.size * 0.5f / (1 << k.o);
            const int radius = __float2int_rn(4.5f * scl_octv);
            const float sigma = 1.5f * scl_octv;
            const float expf_scale = -1.f / (2.f * sigma * sigma);
            const int side = 2 * radius + 1;
            // p / side through the float reciprocal: (p + 0.5) / side is at least 0.5 / side away from an integer, far more
            // than the rounding of the product for any window that fits an image
            const float inv_side = __builtin_amdgcn_rcpf((float)side);
            for (int p = lane; p < side * side; p += 64) {
                const int pi = (int)(((float)p + 0.5f) * inv_side);
                const int i = pi - radius, j = p - pi * side - radius;
                const int y = k.r + i, x = k.c + j;
                if (y <= 0 || y >= rows - 1 || x <= 0 || x >= cols - 1) continue;
                const float dx = img[(size_t)y * cols + x + 1] - img[(size_t)y * cols + x - 1];
                const float dy = img[(size_t)(y - 1) * cols + x] - img[(size_t)(y + 1) * cols + x];
                const float w = __expf((i * i + j * j) * expf_scale);
                const float ori = atan2_deg_lean(dy, dx);
                const float mag = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
                int bin = __float2int_rn((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                atomicAdd(&hacc[wave][bin], (unsigned long long)(long long)__float2int_rn(w * mag * FIX_SCALE));
            }
        }
        wave_lds_fence();
        if (lane < n) hsm[wave][lane + 2] = (float)(long long)hacc[wave][lane] * FIX_INV;
        wave_lds_fence();
        if (lane == 0) {
            float* th = &hsm[wave][2];
            th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1];
        }
        wave_lds_fence();
        float hv = 0.f;
        if (lane < n) {
            const float* th = &hsm[wave][2];
            hv = (th[lane - 2] + th[lane + 2]) * (1.f / 16.f) + (th[lane - 1] + th[lane + 1]) * (4.f / 16.f) +
                 th[lane] * (6.f / 16.f);
        }
        float mx = lane < n ? hv : -FLT_MAX;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        wave_lds_fence();
        if (lane < n) hsm[wave][lane] = hv;  // smoothed histogram, plain indexing from here
        wave_lds_fence();
        bool peak = false;
        KpRec o;
        if (active && lane < n) {
            const float mag_thr = (float)(mx * 0.8f);
            const int j = lane, l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
            const float hl = hsm[wave][l], hj = hsm[wave][j], hr = hsm[wave][r2];
            if (hj > hl && hj > hr && hj >= mag_thr) {
                float bin = j + 0.5f * (hl - hr) / (hl - 2 * hj + hr);
                bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
                float angle = 360.f - (float)((360.f / n) * bin);
                if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
                o.x = k.x; o.y = k.y; o.size = k.size; o.angle = angle; o.response = k.response; o.octave = k.octave;
                peak = true;
            }
        }
        stage_append<KpRec, ORI_STAGE>(stage, n_staged, peak, o, out, cap, count);
        wave_lds_fence();  // hsm/hacc are rewritten by the next keypoint
    }
    stage_flush_block(stage, n_staged, wcnt, out, cap, count);
}

// ---- calcSIFTDescriptor: WPK wavefronts per (sorted) keypoint ------------------------------------
// The patch samples are spread over 64*WPK lanes (a camera frame has fewer keypoints than the chip has
// wave slots, so WPK = 4 there: the patch walk is a quarter as long); the 128 outputs are finished by
// the group's first wave. Fixed-point sums make the result independent of WPK.
#define DESC_HIST (6 * 6 * 10)
template <int WPK>
__global__ void __launch_bounds__(256)
describe_kernel(PyrTable T, const KpRec* __restrict__ kps, SegTab seg, int first_octave, int8_t* __restrict__ rows_i8,
                int* __restrict__ norm, int* __restrict__ tneg, size_t zs, const int* __restrict__ n_dev,
                float2* __restrict__ xy) {
    const size_t zo = (size_t)blockIdx.z * zs;
    kps = ZP(kps);
    int n_kp = seg.off[blockIdx.z + 1] - seg.off[blockIdx.z];
    // n_dev: the launch was queued BEFORE the host knew the image's keypoint count (a single frame: the segment is the
    // list's capacity); the count is the sort kernel's, read where it lies (-1: the sort left the list to the host)
    if (n_dev) n_kp = min(n_kp, *ZP(n_dev));
    constexpr int GROUPS = 4 / WPK, GL = 64 * WPK;
    if ((int)blockIdx.x * GROUPS >= n_kp) return;   // workgroup-uniform
    __shared__ unsigned long long hist[GROUPS][DESC_HIST];
    __shared__ unsigned queue[4][128];   // per wave: indices of samples inside the rotated grid, waiting for a full wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x / GL, gl = threadIdx.x % GL;
    const int id = blockIdx.x * GROUPS + wave;
    for (int i = gl; i < DESC_HIST; i += GL) hist[wave][i] = 0ull;
    __syncthreads();
    const bool active = id < n_kp;
    const int d = 4, n = 8;
    if (active) {
        const KpRec kp = kps[id];
        int octave = kp.octave & 255;
        const int layer = (kp.octave >> 8) & 255;
        octave = octave < 128 ? octave : (-128 | octave);
        const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
        const float size = kp.size * scale;
        const float ptx = kp.x * scale, pty = kp.y * scale;
        const int o = octave - first_octave;
        const float* img = TG(o, layer);
        const int rows = T.rows[o], cols = T.cols[o];
        float ori = 360.f - kp.angle;
        if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
        const float scl = size * 0.5f;
        const int px = __float2int_rn(ptx), py = __float2int_rn(pty);
        // Everything up to the sample loop runs once per WAVE on values every lane shares -- a third of a frame keypoint's
        // instructions before it was cut down: hardware sine / cosine (the angle in revolutions; the window's rotation is
        // good to 1e-6 rad), reciprocals for the divisions, an integer square root with a two-sided fix-up for the radius
        // clamp instead of a double-precision one, 32-bit remainders.
        float cos_t = __builtin_amdgcn_cosf(ori * (1.0f / 360.0f));
        float sin_t = __builtin_amdgcn_sinf(ori * (1.0f / 360.0f));
        const float bins_per_rad = n / 360.f;
        const float exp_scale = -1.f / (d * d * 0.5f);
        const float hist_width = 3.f * scl;
        int radius = __float2int_rn(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
        {   // min(radius, (int)sqrt(cols^2 + rows^2)): the sum stays below 2^31 for layers up to 32768 x 32768
            const unsigned v = (unsigned)cols * (unsigned)cols + (unsigned)rows * (unsigned)rows;
            unsigned rt = (unsigned)__builtin_amdgcn_sqrtf((float)v);
            while (rt * rt > v) --rt;
            while ((rt + 1) * (rt + 1) <= v) ++rt;
            radius = min(radius, (int)rt);
        }
        const float inv_hw = __builtin_amdgcn_rcpf(hist_width);
        cos_t *= inv_hw;
        sin_t *= inv_hw;
        const int side = 2 * radius + 1, total = side * side;
        // Lanes of one instruction must not sit on neighbouring pixels: those fall into the same spatial cell
        // and similar orientation bins, and same-address LDS atomics serialise (that was 2/3 of this kernel).
        // total is odd, so multiplying the sample number by a power of two is a bijection mod total; one near
        // 0.3*total sends consecutive lanes to different bands of the patch. Sums are fixed-point, so the
        // visiting order does not change a single bit of the result.
        const int spread = total >= 4 ? 1 << (31 - __clz(total >> 2)) : 1;   // the largest power of two <= total / 4
        // p = (q * spread) mod total, kept incrementally (q advances by GL: one add and a conditional subtract),
        // and p / side through the float reciprocal: (p + 0.5) / side is at least 0.5 / side away from an integer,
        // far more than the rounding of the product for any patch side that fits an image. The three integer
        // divisions per sample this replaces were a quarter of the kernel's instructions.
        const unsigned step = ((unsigned)GL * (unsigned)spread) % (unsigned)total;   // GL * spread < 2^8 * total / 4: 32 bits do
        unsigned p = ((unsigned)gl * (unsigned)spread) % (unsigned)total;
        const float inv_side = __builtin_amdgcn_rcpf((float)side);
        // About half of the square's samples fall outside the rotated 4x4 grid. Testing and accumulating in one
        // loop leaves those lanes idle through the ~200 instructions of the accumulation; instead every wave
        // QUEUES the samples that pass the test (their index p, slots from ballot prefix counts) and runs the
        // accumulation on 64 queued samples at a time, all lanes busy. The fixed-point sums do not depend on the order.
        unsigned* const wq = queue[threadIdx.x >> 6];
        int n_q = 0;   // wave-uniform
        // the keypoint, hence its layer, is the same for every lane of the wave: the layer becomes a buffer resource in
        // scalar registers (base made wave-uniform explicitly, else the compiler wraps every load in a waterfall loop), a
        // sample's four pixels are raw buffer loads at 32-bit byte offsets, one of them with an immediate
        // (readfirstlane returns int: both halves go through unsigned, or a low half with bit 31 set smears over the high one)
        // The resource starts at the first row the window can touch (row py - radius - 1), not at the layer's: a buffer
        // resource counts bytes in 32 bits and a 32768^2 layer of a 16384^2 map is exactly 2^32 of them -- with the layer's
        // base and size, num_records wrapped to 0 and every octave-0 descriptor of such a map was built from zeros (found
        // by the full-size test of configs[3]'s map leg). The window's rows are at most (2 radius + 3) x cols x 4 bytes
        // (15 MB for the widest layer), and the byte offsets below count from row_base.
        const int row_base = max(py - radius - 1, 0);
        const unsigned long long img_bits = (unsigned long long)(img + (size_t)row_base * (size_t)cols);
        const unsigned img_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)img_bits);
        const unsigned img_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(img_bits >> 32));
        void* const img_u = (void*)(((unsigned long long)img_hi << 32) | (unsigned long long)img_lo);
        const unsigned long long bytes_left = (unsigned long long)(rows - row_base) * (unsigned long long)cols * 4ull;
        const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc(
            img_u, 0, __builtin_amdgcn_readfirstlane((int)(unsigned)(bytes_left > 0xffffffffull ? 0xffffffffull : bytes_left)),
            0x00020000);
        const int py_rel = py - row_base;
        const unsigned cols4 = (unsigned)cols * 4u;
        unsigned long long* const hp = hist[wave];
        auto accumulate = [&](unsigned ps) {
            const int pi = (int)(((float)ps + 0.5f) * inv_side);
            const int i = pi - radius, j = (int)ps - pi * side - radius;
            const float c_rot = j * cos_t - i * sin_t;
            const float r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f;
            float cbin = c_rot + d / 2 - 0.5f;
            const unsigned at = ((unsigned)(py_rel + i) * (unsigned)cols + (unsigned)(px + j)) * 4u - 4u;   // bytes from row_base, pixel x - 1
            const float dx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(img_rs, (int)(at + 8u), 0, 0)) -
                             __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(img_rs, (int)at, 0, 0));
            const float dy = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(img_rs, (int)(at + 4u - cols4), 0, 0)) -
                             __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(img_rs, (int)(at + 4u + cols4), 0, 0));
            const float w = __expf((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (atan2_deg_lean(dy, dx) - ori) * bins_per_rad;
            // the sums are kept in 2^-20 fixed point: the scale goes in here once (a power of two: every product below
            // is the unscaled one times 2^20, bit for bit)
            const float mag = __builtin_amdgcn_sqrtf(dx * dx + dy * dy) * w * FIX_SCALE;
            const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
            int o0 = (int)floorf(obin);
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
            // every term is >= 0 (a magnitude times fractions in [0, 1)) and below 2^31: unsigned, high word zero
#define HADD(off, val) atomicAdd(&hp[idx + (off)], (unsigned long long)__float2uint_rn(val))
            HADD(0, v_rco000); HADD(1, v_rco001);
            HADD(n + 2, v_rco010); HADD(n + 3, v_rco011);
            HADD((d + 2) * (n + 2), v_rco100); HADD((d + 2) * (n + 2) + 1, v_rco101);
            HADD((d + 3) * (n + 2), v_rco110); HADD((d + 3) * (n + 2) + 1, v_rco111);
#undef HADD
        };
        for (int q0 = 0; q0 < total; q0 += GL) {      // wave-uniform trip count
            const int q = q0 + gl;
            bool ok = q < total;
            if (ok) {
                const int pi = (int)(((float)p + 0.5f) * inv_side);
                const int i = pi - radius, j = (int)p - pi * side - radius;
                const float c_rot = j * cos_t - i * sin_t;
                const float r_rot = j * sin_t + i * cos_t;
                const float rbin = r_rot + d / 2 - 0.5f;
                const float cbin = c_rot + d / 2 - 0.5f;
                const int r = py + i, c = px + j;
                ok = rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < rows - 1 && c > 0 && c < cols - 1;
            }
            const unsigned long long mk = __ballot(ok);
            if (ok) wq[n_q + __popcll(mk & ((1ull << lane) - 1ull))] = p;
            n_q += __popcll(mk);
            p += step;
            if (p >= (unsigned)total) p -= (unsigned)total;
            if (n_q >= 64) {
                wave_lds_fence();
                const unsigned ps = wq[lane];
                const unsigned carry = lane < n_q - 64 ? wq[64 + lane] : 0;
                wave_lds_fence();
                if (lane < n_q - 64) wq[lane] = carry;
                n_q -= 64;
                accumulate(ps);
            }
        }
        if (n_q > 0) {
            wave_lds_fence();
            if (lane < n_q) accumulate(wq[lane]);
        }
    }
    __syncthreads();
    if (gl >= 64) return;  // the group's first wave finishes
    // fold the circular orientation bins, normalise, clamp, renormalise, quantise
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int el = lane + 64 * e;
        const int i = el >> 5, j = (el >> 3) & 3, k = el & 7;
        const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
        long long s = (long long)hist[wave][idx + k];
        if (k < 2) s += (long long)hist[wave][idx + n + k];
        v[e] = (float)s * FIX_INV;
    }
    float nrm2 = v[0] * v[0] + v[1] * v[1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nrm2 += __shfl_xor(nrm2, o);
    const float thr = sqrtf(nrm2) * 0.2f;
    v[0] = fminf(v[0], thr);
    v[1] = fminf(v[1], thr);
    nrm2 = v[0] * v[0] + v[1] * v[1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nrm2 += __shfl_xor(nrm2, o);
    nrm2 = 512.f / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    // The quantised descriptor goes straight into the matcher's row format (what pack_rows_kernel made of the u8 rows in a
    // pass of its own: 0.2 ms of a 64-frame batch): int8 values x - 128, the squared norm of that shifted row and its
    // negative (common.h, imatch_features).
    if (active) {
        const size_t row = (size_t)seg.off[blockIdx.z] + (size_t)id;
        int s2 = 0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int qi = min(max(__float2int_rn(v[e] * nrm2), 0), 255) - 128;
            rows_i8[row * 128 + lane + 64 * e] = (int8_t)qi;
            s2 += qi * qi;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
        if (lane == 0) {
            norm[row] = s2;
            tneg[row] = -s2;
            const KpRec k2 = kps[id];
            xy[row] = make_float2(k2.x, k2.y);   // the set's coordinate table (imatch_match_localize)
        }
    }
}

// ---- KeyPoint12_LessThan sort + removeDuplicatedSorted + first-octave rescale on the device ------------------------------
// A camera frame has a few thousand keypoint records: fetching them, sorting them on the host and sending them back held
// the GPU idle for ~115 us of a 0.7 ms frame. kp_sort_kernel does it where the records lie, one 1024-thread workgroup per
// image, everything in LDS:
//   1. keys (order-preserving bits of x and y, 8 B per record) -> LDS;
//   2. a counting sort on 256 buckets of x (x is the leading key and the bucket number is monotone in it): histogram by
//      LDS atomics, prefix, record numbers scattered bucket by bucket;
//   3. a record's place = its bucket's start + the number of bucket-mates that come before it (a dozen compares; records
//      of equal (x, y) -- the several orientations of one point -- go through the full comparator: size desc, angle asc,
//      response desc, octave desc, read from the records themselves);
//   4. duplicates (equal x, y, size, angle) are neighbours now and are dropped, the rest rescaled to the input image's
//      frame and written back to back (flags, a workgroup scan, strided writes).
// (The first version ranked every record against ALL others, 16 or 4 lanes per record over many workgroups: 14 us for a
// frame, 0.23 ms for a 64-frame batch.) Above the launch's limit (the previous extraction's count plus a margin, at most
// SORT_CAP) the kernel reports -1 and the host's radix sort runs on the raw records.
#define SORT_CAP 8192
#define SORT_NB 256
#define SORT_LDS_BYTES (SORT_CAP * (8 + 2 + 2 + 1 + 2) + (2 * SORT_NB + 2 + 16) * 4)

__device__ __forceinline__ unsigned ord_bits(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// a before b, keys equal
__device__ __forceinline__ bool kp_tie_before(const KpRec* __restrict__ rec, int ia, int ib) {
    const KpRec a = rec[ia], b = rec[ib];
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    if (a.octave != b.octave) return a.octave > b.octave;
    return ia < ib;
}

__global__ void __launch_bounds__(1024)
kp_sort_kernel(const KpRec* __restrict__ rec, const int* __restrict__ n_dev, int limit, float bucket_scale,
               KpRec* __restrict__ out, int* __restrict__ k_out, int first_octave, size_t zs) {
    const size_t zo = (size_t)blockIdx.z * zs;
    rec = ZP(rec); n_dev = ZP(n_dev); out = ZP(out); k_out = ZP(k_out);
    extern __shared__ unsigned long long sort_lds[];
    unsigned long long* const keys = sort_lds;                                   // [SORT_CAP]
    unsigned short* const member = (unsigned short*)(keys + SORT_CAP);            // record numbers, bucket by bucket
    unsigned short* const order = member + SORT_CAP;                              // record number at every sorted place
    unsigned short* const off_s = order + SORT_CAP;                               // output place of every sorted place
    unsigned char* const keep_s = (unsigned char*)(off_s + SORT_CAP);
    int* const hist = (int*)(keep_s + SORT_CAP);                                  // [SORT_NB] counts, then fill cursors
    int* const start = hist + SORT_NB;                                            // [SORT_NB + 1]
    int* const wsum = start + SORT_NB + 2;                                        // [16]
    const int tid = threadIdx.x;
    const int n = *n_dev;
    if (n > limit) {
        if (tid == 0) *k_out = -1;
        return;
    }
    constexpr int E = SORT_CAP / 1024;
    // 1. + 2a. keys and bucket counts (all of a thread's loads in flight before the first use)
    float2 xy[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = min(tid + 1024 * e, max(n - 1, 0));
        xy[e] = *(const float2*)&rec[i].x;
    }
    if (tid < SORT_NB) hist[tid] = 0;
    __syncthreads();
    int bkt[E], pos[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + 1024 * e;
        if (i < n) {
            keys[i] = ((unsigned long long)ord_bits(xy[e].x) << 32) | ord_bits(xy[e].y);
            bkt[e] = min(max((int)(xy[e].x * bucket_scale), 0), SORT_NB - 1);
            pos[e] = atomicAdd(&hist[bkt[e]], 1);
        }
    }
    __syncthreads();
    // 2b. bucket starts: exclusive prefix of the 256 counts by the first four waves
    {
        const int lane = tid & 63, wave = tid >> 6;
        int v = tid < SORT_NB ? hist[tid] : 0, incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        if (tid < SORT_NB && lane == 63) wsum[wave] = incl;
        __syncthreads();
        if (tid < SORT_NB) {
            int base = 0;
            for (int w = 0; w < wave; ++w) base += wsum[w];
            start[tid] = base + incl - v;
            if (tid == SORT_NB - 1) start[SORT_NB] = base + incl;
        }
    }
    __syncthreads();
    // 2c. members of every bucket side by side
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + 1024 * e;
        if (i < n) member[start[bkt[e]] + pos[e]] = (unsigned short)i;
    }
    __syncthreads();
    // 3. place = bucket start + bucket-mates before me
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + 1024 * e;
        if (i < n) {
            const unsigned long long mine = keys[i];
            const int j0 = start[bkt[e]], j1 = start[bkt[e] + 1];
            int before = j0;
            for (int j = j0; j < j1; ++j) {
                const int m = member[j];
                const unsigned long long k = keys[m];
                if (k < mine) ++before;
                else if (k == mine && m != i && kp_tie_before(rec, m, i)) ++before;
            }
            order[before] = (unsigned short)i;
        }
    }
    __syncthreads();
    // 4. keep flags (place i against place i - 1), scan, compaction
    KpRec mine_rec[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + 1024 * e;
        bool kp = false;
        if (i < n) {
            const KpRec c = rec[order[i]];
            mine_rec[e] = c;
            kp = true;
            if (i > 0) {
                const KpRec a = rec[order[i - 1]];
                kp = a.x != c.x || a.y != c.y || a.size != c.size || a.angle != c.angle;
            }
        }
        keep_s[i] = kp;
    }
    __syncthreads();
    const int base = tid * E;   // consecutive runs of E places per thread for the scan
    int mine = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) mine += keep_s[base + e];
    const int lane = tid & 63, wave = tid >> 6;
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = incl - mine;
    for (int w = 0; w < wave; ++w) off += wsum[w];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        off_s[base + e] = (unsigned short)off;
        off += keep_s[base + e];
    }
    if (tid == 1023) *k_out = off;
    __syncthreads();
    const float scale = 1.f / (float)(1 << -first_octave);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = tid + 1024 * e;
        if (i < n && keep_s[i]) {
            KpRec k = mine_rec[e];
            k.octave = (k.octave & ~255) | ((k.octave + first_octave) & 255);
            k.x *= scale; k.y *= scale; k.size *= scale;
            out[off_s[i]] = k;
        }
    }
}

// ---- host side -------------------------------------------------------------------------------------
// getGaussianKernel(ksize, sigma, CV_32F), bit-exact form (see oracle/sift_oracle.c oracle_gauss_kernel)
static int gauss_kernel(double sigma, float* k, int maxn) {
    int n = (int)lrint(sigma * 4 * 2 + 1) | 1;
    if (n > maxn) return -1;
    const int n2 = (n - 1) / 2;
    const double scale2x = -0.125 / (sigma * sigma);
    std::vector<double> vals(n2 + 1);
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

// GaussianBlur(src, dst, Size(), sigma). The templated radii take the marching kernel, which can also
//   - read the u8 image and upsample it on the fly (up_u8 != nullptr: src is ignored, rows x cols is the x2 size),
//   - write the next octave's layer 0 (dec != nullptr);
// other radii (non-default sigma / nOctaveLayers) run per-tap kernels through `tmp`; asked for either
// extra they set *fused = false, and the caller launches upsample / decimate itself.
static bool blur_radius_is_templated(int r) { return r == 4 || r == 5 || r == 6 || r == 8 || r == 10 || r == 13; }

template <int R>
static void launch_blur_fixed(hipStream_t st, const float* src, const uint8_t* up_u8, int ustride, size_t src_zs, float* dst,
                              float* dec, int rows, int cols, const float* k, int batch, size_t zs) {
    if (up_u8) blur_march_launch<R, 64, true>(st, up_u8, dst, dec, rows, cols, ustride, k, batch, src_zs, zs);
    else blur_march_launch<R, 64, false>(st, src, dst, dec, rows, cols, 0, k, batch, zs, zs);
}

static int gaussian_blur(imatch_ctx* ctx, hipStream_t st, const float* src, const uint8_t* up_u8, int ustride, size_t src_zs,
                         float* dst, float* dec, float* tmp, int rows, int cols, double sigma, int batch, size_t zs,
                         bool* fused) {
    float k[2 * RT_MAXR + 1];
    const int n = gauss_kernel(sigma, k, 2 * RT_MAXR + 1);
    if (n < 0) return imatch_set_err(ctx, IMATCH_ESIZE, "Gaussian sigma %g needs a kernel wider than %d", sigma, 2 * RT_MAXR + 1);
    const int r = n / 2;
    if (fused) *fused = true;
    switch (r) {
        case 4: launch_blur_fixed<4>(st, src, up_u8, ustride, src_zs, dst, dec, rows, cols, k, batch, zs); break;
        case 5: launch_blur_fixed<5>(st, src, up_u8, ustride, src_zs, dst, dec, rows, cols, k, batch, zs); break;
        case 6: launch_blur_fixed<6>(st, src, up_u8, ustride, src_zs, dst, dec, rows, cols, k, batch, zs); break;
        case 8: launch_blur_fixed<8>(st, src, up_u8, ustride, src_zs, dst, dec, rows, cols, k, batch, zs); break;
        case 10: launch_blur_fixed<10>(st, src, up_u8, ustride, src_zs, dst, dec, rows, cols, k, batch, zs); break;
        case 13: launch_blur_fixed<13>(st, src, up_u8, ustride, src_zs, dst, dec, rows, cols, k, batch, zs); break;
        default: {
            if (up_u8 || dec) {
                if (!fused) return imatch_set_err(ctx, IMATCH_EINVAL, "internal: fused blur needs a templated radius");
                *fused = false;
                if (up_u8) return IMATCH_OK;   // the caller materialises the x2 image and calls again with src
            }
            GKrt gk;
            gk.r = r;
            memcpy(gk.k, k, sizeof(float) * n);
            const dim3 grid((cols + 63) / 64, (rows + 3) / 4, batch);
            hipLaunchKernelGGL(blur_rows_rt_kernel, grid, dim3(256), 0, st, src, tmp, rows, cols, gk, zs);
            hipLaunchKernelGGL(blur_cols_rt_kernel, grid, dim3(256), 0, st, (const float*)tmp, dst, rows, cols, gk, zs);
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

static bool kp_less(const KpRec& a, const KpRec& b) {  // KeyPoint12_LessThan (keypoint.cpp)
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    if (a.octave != b.octave) return a.octave > b.octave;
    return false;
}

// Sort by KeyPoint12_LessThan. (x, y) decide almost every comparison, so the records are radix-sorted
// (LSD, 11-bit digits) on the 64-bit key made of the two floats' order-preserving bit patterns, and only
// runs of equal (x, y) - the several orientations of one point - go through the full comparator.
static void sort_keypoints(const KpRec* in, int n, KpRec* out) {
    auto ord = [](float f) {
        uint32_t u;
        memcpy(&u, &f, 4);
        return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    };
    std::vector<uint64_t> key((size_t)n), key2((size_t)n);
    std::vector<uint32_t> idx((size_t)n), idx2((size_t)n);
    for (int i = 0; i < n; ++i) {
        key[i] = ((uint64_t)ord(in[i].x) << 32) | ord(in[i].y);
        idx[i] = (uint32_t)i;
    }
    for (int shift = 0; shift < 64; shift += 11) {
        uint32_t hist[2049] = {0};
        for (int i = 0; i < n; ++i) hist[((key[i] >> shift) & 2047) + 1]++;
        bool single = false;
        for (int d = 1; d <= 2048; ++d) single = single || hist[d] == (uint32_t)n;
        if (single) continue;
        for (int d = 1; d <= 2048; ++d) hist[d] += hist[d - 1];
        for (int i = 0; i < n; ++i) {
            const uint32_t at = hist[(key[i] >> shift) & 2047]++;
            key2[at] = key[i];
            idx2[at] = idx[i];
        }
        key.swap(key2);
        idx.swap(idx2);
    }
    for (int i = 0; i < n; ++i) out[i] = in[idx[i]];
    for (int i = 0; i < n;) {
        int j = i + 1;
        while (j < n && key[j] == key[i]) ++j;
        if (j - i > 1) std::sort(out + i, out + j, kp_less);
        i = j;
    }
}

static size_t al(size_t b) { return (b + 255) & ~(size_t)255; }

extern "C" {

// One extraction of `batch` equally sized images: image b is img + b * img_pitch bytes. Every kernel is
// launched once for the whole batch (gridDim.z = batch) over `batch` copies of the per-image arena
// layout, zs bytes apart; on a camera frame the launches are latency-bound, so a batch costs little
// more than one image. The result is ONE descriptor set holding the images' rows back to back;
// offsets[b] .. offsets[b+1] are image b's rows (and keypoint records).
static int sift_extract(imatch_ctx* ctx, const uint8_t* img, int img_on_device, int batch, size_t img_pitch, int h, int w,
                        int stride, const imatch_sift_params* P, imatch_features** out, int64_t* offsets) {
    // IMATCH_DEBUG_STEPS=1: a line on stderr at every point where the host may wait (to place a hang)
    static const bool dbg_steps = getenv("IMATCH_DEBUG_STEPS") != nullptr;
#define DBG_STEP(msg) do { if (dbg_steps) { fprintf(stderr, "[imatch %d] sift_extract: %s\n", (int)getpid(), msg); fflush(stderr); } } while (0)
    DBG_STEP("enter");
    if (!ctx || !img || !P || !out || h <= 0 || w <= 0 || stride < w || batch < 1)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    if (batch > IMATCH_MAX_BATCH) return imatch_set_err(ctx, IMATCH_ESIZE, "batch %d above %d", batch, IMATCH_MAX_BATCH);
    if (batch > 1 && (!img_on_device || !offsets)) return imatch_set_err(ctx, IMATCH_EINVAL, "a batch needs device images and offsets");
    if (P->n_octave_layers < 1 || P->n_octave_layers + 3 > MAX_LAY)
        return imatch_set_err(ctx, IMATCH_ESIZE, "nOctaveLayers %d outside 1..%d", P->n_octave_layers, MAX_LAY - 3);
    if (h > 16384 || w > 16384) return imatch_set_err(ctx, IMATCH_ESIZE, "image larger than 16384 on a side");
    if (P->clahe && (P->clahe_tiles_x <= 0 || P->clahe_tiles_y <= 0 || P->clahe_tiles_x * P->clahe_tiles_y > 4096))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad CLAHE tile grid");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    const int nl = P->n_octave_layers, first_octave = -1;
    const int brows = 2 * h, bcols = 2 * w;
    int n_oct = (int)lrint(log((double)std::min(brows, bcols)) / log(2.) - 2) - first_octave;
    if (n_oct < 1) n_oct = 1;
    if (n_oct > MAX_OCT) n_oct = MAX_OCT;

    // ---- arena layout of ONE image (every offset is relative to the image's block)
    if (!ctx->sift) {
        ctx->sift = (imatch_sift_state*)calloc(1, sizeof(imatch_sift_state));
        if (!ctx->sift) return imatch_set_err(ctx, IMATCH_ENOMEM, "host alloc");
    }
    imatch_sift_state* S = ctx->sift;
    PyrTable T;
    memset(&T, 0, sizeof(T));
    T.nl = nl;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_img = take((size_t)h * w), o_clahe = take((size_t)h * w);
    const size_t o_cscr = take((size_t)(P->clahe ? P->clahe_tiles_x * P->clahe_tiles_y : 1) * 1280);
    // the x2 f32 image and the row-pass scratch exist only on the runtime-radius path
    double sig[MAX_LAY];
    sig[0] = P->sigma;
    {
        const double kf = pow(2., 1. / nl);
        for (int i = 1; i < nl + 3; i++) {
            const double sig_prev = pow(kf, (double)(i - 1)) * P->sigma, sig_total = sig_prev * kf;
            sig[i] = sqrt(sig_total * sig_total - sig_prev * sig_prev);
        }
    }
    const float sig_diff = sqrtf(fmaxf((float)P->sigma * (float)P->sigma - 0.5f * 0.5f * 4, 0.01f));
    bool all_templated = true;
    {
        float kk[2 * RT_MAXR + 1];
        int n = gauss_kernel(sig_diff, kk, 2 * RT_MAXR + 1);
        if (n < 0 || !blur_radius_is_templated(n / 2)) all_templated = false;
        for (int i = 1; i < nl + 3; ++i) {
            n = gauss_kernel(sig[i], kk, 2 * RT_MAXR + 1);
            if (n < 0 || !blur_radius_is_templated(n / 2)) all_templated = false;
        }
    }
    const size_t o_up = take(all_templated ? 0 : (size_t)brows * bcols * 4);
    const size_t o_tmp = take(all_templated ? 0 : (size_t)brows * bcols * 4);
    size_t o_g[MAX_OCT][MAX_LAY];
    for (int o = 0; o < n_oct; ++o) {
        T.rows[o] = o == 0 ? brows : T.rows[o - 1] / 2;
        T.cols[o] = o == 0 ? bcols : T.cols[o - 1] / 2;
        if (T.rows[o] < 1 || T.cols[o] < 1) { n_oct = o; break; }
        const size_t px = (size_t)T.rows[o] * T.cols[o] * 4;
        // Large layers are staggered: a 16384^2 layer is exactly 2^30 bytes, and with the layers back to back a
        // kernel's read and write streams (row y of layer i, row y of layer i+1; the scan's six layers) sit on the
        // same HBM channels at every instant (tools/blur_bench.hip `pipe`: R = 13 0.83 ms -> 0.68 ms with the gap;
        // inside the extraction the effect is within run-to-run noise, the gap costs 6 MB per octave).
        const size_t stagger = px >= ((size_t)16 << 20) ? (size_t)1057024 : 0;   // (a stagger on frame-sized layers changes nothing: profiles/r3_small_experiments.txt)
        for (int i = 0; i < nl + 3; ++i) { o_g[o][i] = take(px); (void)take(stagger); }
    }
    T.n_oct = n_oct;
    int cap = (int)std::min<size_t>((size_t)brows * bcols / 8 + 16384, (size_t)1 << 28);
    const size_t o_cnt = take(256);
    const size_t fixed_end = off;
    for (int attempt = 0; attempt < 5; ++attempt) {
        off = fixed_end;
        const size_t o_cand = take((size_t)cap * sizeof(uint2));
        const size_t o_ref = take((size_t)cap * sizeof(RefinedKp));
        const size_t o_kp = take((size_t)cap * 2 * sizeof(KpRec));
        const size_t zs = (off + 4095) & ~(size_t)4095;   // bytes between the blocks of consecutive images
        const size_t need = zs * (size_t)batch;
        if (need > S->arena_bytes) {
            DBG_STEP("arena grows: sync");
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            DBG_STEP("arena grows: free + malloc");
            if (S->arena) HIP_TRY(ctx, hipFree(S->arena));
            S->arena = nullptr;
            S->arena_bytes = 0;
            if (hipMalloc((void**)&S->arena, need) != hipSuccess)
                return imatch_set_err(ctx, IMATCH_ENOMEM, "SIFT arena of %zu bytes", need);
            S->arena_bytes = need;
        }
        char* A = S->arena;
        for (int o = 0; o < n_oct; ++o) {
            for (int i = 0; i < nl + 3; ++i) T.gauss[o][i] = (float*)(A + o_g[o][i]);
        }
        S->T = T;
        S->h = h;
        S->w = w;
        uint8_t* d_img = (uint8_t*)(A + o_img);
        uint8_t* d_clahe = (uint8_t*)(A + o_clahe);
        float* d_up = (float*)(A + o_up);
        float* d_tmp = (float*)(A + o_tmp);
        int* d_cnt = (int*)(A + o_cnt);
        uint2* d_cand = (uint2*)(A + o_cand);
        RefinedKp* d_ref = (RefinedKp*)(A + o_ref);
        KpRec* d_kp = (KpRec*)(A + o_kp);

        // ---- pre-processing + pyramid. Device images with dense rows are read where they lie.
        const uint8_t* d_src = d_img;
        int src_stride = w;
        size_t src_zs = zs;
        if (img_on_device && (stride == w || batch > 1)) {
            d_src = img;
            src_stride = stride;
            src_zs = img_pitch;
        } else if (img_on_device) {
            HIP_TRY(ctx, hipMemcpy2DAsync(d_img, (size_t)w, img, (size_t)stride, (size_t)w, (size_t)h, hipMemcpyDeviceToDevice,
                                          ctx->stream));
        } else {
            DBG_STEP("arena ready: upload");
            int rcu = imatch_stage_upload(ctx, img, h, w, stride, d_img);   // consumed before this call's synchronisation
            DBG_STEP("upload queued");
            if (rcu) return rcu;
        }
        bool cnt_cleared = false;   // the image's counter block (64 ints): cleared by the CLAHE tile kernel on its way, or here
        if (P->clahe) {
            (void)imatch_stage_begin(ctx, IMATCH_STAGE_CLAHE, ctx->stream);
            int rc = imatch_clahe_dev(ctx, d_src, h, w, src_stride, P->clahe_clip, P->clahe_tiles_x, P->clahe_tiles_y, d_clahe,
                                      A + o_cscr, batch, src_zs, zs, d_cnt, &cnt_cleared);
            if (rc) return rc;
            (void)imatch_stage_end(ctx, IMATCH_STAGE_CLAHE, ctx->stream);
            d_src = d_clahe;
            src_stride = w;
            src_zs = zs;
        }
        if (cnt_cleared) {
        } else if (batch == 1) HIP_TRY(ctx, hipMemsetAsync(d_cnt, 0, 256, ctx->stream));
        else HIP_TRY(ctx, hipMemset2DAsync(d_cnt, zs, 0, 256, (size_t)batch, ctx->stream));
        { const int rcs = imatch_stage_begin(ctx, IMATCH_STAGE_SCALE_SPACE, ctx->stream); if (rcs) return rcs; }
        bool fused = false;
        int rc = gaussian_blur(ctx, ctx->stream, nullptr, d_src, src_stride, src_zs, T.gauss[0][0], nullptr, d_tmp, brows, bcols,
                               sig_diff, batch, zs, &fused);
        if (rc) return rc;
        if (!fused) {
            hipLaunchKernelGGL(upsample2_kernel, dim3((bcols + 63) / 64, (brows + 3) / 4, batch), dim3(256), 0, ctx->stream,
                               d_src, h, w, src_stride, d_up, src_zs, zs);
            rc = gaussian_blur(ctx, ctx->stream, d_up, nullptr, 0, 0, T.gauss[0][0], nullptr, d_tmp, brows, bcols, sig_diff, batch,
                               zs, nullptr);
            if (rc) return rc;
        }
        const float thr = (float)(int)floor(0.5 * P->contrast_threshold / nl * 255);
        // octaves from o_tail on go to the single-workgroup kernel
        int o_tail = n_oct;
        static const size_t tail_px = getenv("IMATCH_SIFT_TAILPX") ? (size_t)atoi(getenv("IMATCH_SIFT_TAILPX")) : TAIL_MAXPX;
        for (int o = n_oct - 1; o >= 1 && (size_t)T.rows[o] * T.cols[o] <= std::min<size_t>(tail_px, TAIL_MAXPX); --o) o_tail = o;
        // fork/join (see imatch_sift_state); the runtime-radius blur shares d_tmp, so it stays on one stream.
        // An event record + cross-stream wait costs ~20 us (tools/launch_bench.hip), so small work stays serial.
        bool use_side = all_templated && n_oct > 1 && (size_t)brows * bcols * batch >= ((size_t)1 << 22);
        if (const char* e = getenv("IMATCH_SIFT_SIDE")) use_side = use_side && atoi(e) != 0;   // experiments
        // small inputs: the chain-splitting form (see imatch_sift_state)
        bool split = all_templated && n_oct > 1 && !use_side && o_tail >= 1 && nl == 3;
        if (const char* e = getenv("IMATCH_SIFT_SPLIT")) split = split && atoi(e) != 0;        // experiments
        // Large inputs (maps, batches), chain first: the same split of every octave -- layers 1..nl on the main stream, the
        // last two layers and the scan behind an event per octave -- with the per-layer marching kernels. The old order
        // (octave 0 whole on the main stream, ALL later octaves one after the other on the second) ended with the small
        // octaves' forty latency-bound launches running alone for 0.85 ms of the 8192^2 map's 6.7 ms after everything big had
        // finished; now that chain is through 0.9 ms after octave 0's layer nl and the streams end on chip-filling
        // kernels. Octave 0's tail goes to `side`, the later octaves' tails to `side2`, so that they run beside each other
        // (IMATCH_SIFT_ORDER: 0 = old order, 1 = both on `side`).
        const bool big = use_side;
        static const int order_env = getenv("IMATCH_SIFT_ORDER") ? atoi(getenv("IMATCH_SIFT_ORDER")) : 2;
        const bool chain_first = big && order_env != 0 && o_tail >= 1;
        if (chain_first) {
            use_side = false;
            split = true;
        }
        if ((use_side || split) && !S->side) {
            // Default priority. (Rounds 1-2 created this stream at the highest priority so that the later octaves' small
            // launches would get CU slots beside octave 0's big kernels. With TWO processes on one GPU -- the rehearsal
            // form of bench.py -- that deadlocked about one run in five: a rank's first extraction never finished; forty
            // runs without the priority were clean, and one process alone is no slower without it. Which queue or
            // kernel was stuck was never identified -- see DESIGN.md section 5 -- so the configuration is gone, knob
            // included.)
            DBG_STEP("second stream: create");
            HIP_TRY(ctx, hipStreamCreateWithFlags(&S->side, hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_fork, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_join, hipEventDisableTiming));
        }
        const bool three = chain_first && order_env >= 2;
        if (three && !S->side2) {
            HIP_TRY(ctx, hipStreamCreateWithFlags(&S->side2, hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_join2, hipEventDisableTiming));
        }
        if (split && !S->have_ev_oct) {
            for (int o = 0; o < MAX_OCT; ++o) HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_oct[o], hipEventDisableTiming));
            DBG_STEP("second stream and events created");
            S->have_ev_oct = true;
        }
        hipStream_t st2 = use_side ? S->side : ctx->stream;
        auto scan_octave = [&](int o, hipStream_t st) {
            const int rows = T.rows[o], cols = T.cols[o];
            if (rows <= 2 * SIFT_IMG_BORDER || cols <= 2 * SIFT_IMG_BORDER) return;
            const int iw = cols - 2 * SIFT_IMG_BORDER, ih = rows - 2 * SIFT_IMG_BORDER;
            if (nl == 3) {
                const size_t px = (size_t)iw * ih * batch;
                static const int rpw_env = getenv("IMATCH_SIFT_RPW") ? atoi(getenv("IMATCH_SIFT_RPW")) : 0;   // experiments
                const int rpw = rpw_env > 0 ? rpw_env : px >= ((size_t)1 << 26) ? 64 : px >= ((size_t)1 << 24) ? 32 : px >= ((size_t)1 << 20) ? 16 : 8;
                // columns per lane (profiles/r3_scanv_sweep.txt): 4 on the 32768^2 layers of a 16384^2 map (scale space 25.1 ->
                // 22.6 ms; 24.1 with 2), 2 elsewhere (8192^2 map 6.43 -> 6.11 ms; 6.26 with 4: fewer waves per SIMD), 1 where
                // the rows are odd or narrower than a strip (IMATCH_SIFT_SCANV = 1 / 2 / 4 forces a form where it is possible)
                static const int scanv_env = getenv("IMATCH_SIFT_SCANV") ? atoi(getenv("IMATCH_SIFT_SCANV")) : 0;
                int V = (cols % 4 == 0 && (size_t)iw * ih >= ((size_t)1 << 29)) ? 4 : (cols % 2 == 0 && iw >= 126) ? 2 : 1;
                if (scanv_env == 1 || (scanv_env == 2 && cols % 2 == 0) || (scanv_env == 4 && cols % 4 == 0)) V = scanv_env;
                const int ow = V == 1 ? DS_COLS : 63 * V;
                const int nsx = (iw + ow - 1) / ow, nsy = (ih + 4 * rpw - 1) / (4 * rpw);
                const dim3 grid((unsigned)((nsx * nsy + 7) & ~7), 1, batch);
                if (V == 4)
                    hipLaunchKernelGGL(detect_streamv_kernel<4>, grid, dim3(256), 0, st, T, o, rpw, thr, d_cand, cap, d_cnt, zs, nsx, nsy);
                else if (V == 2)
                    hipLaunchKernelGGL(detect_streamv_kernel<2>, grid, dim3(256), 0, st, T, o, rpw, thr, d_cand, cap, d_cnt, zs, nsx, nsy);
                else
                    hipLaunchKernelGGL(detect_stream_kernel, grid, dim3(256), 0, st, T, o, rpw, thr, d_cand, cap, d_cnt, zs, nsx, nsy);
            } else {
                hipLaunchKernelGGL(detect_kernel_generic, dim3((iw + DT_W - 1) / DT_W, (ih + DT_H - 1) / DT_H, batch), dim3(256),
                                   0, st, T, o, thr, d_cand, cap, d_cnt, zs);
            }
        };
        bool have_l0 = true;   // layer 0 of octave o was written by the previous octave's layer-nl blur
        // layers 1..3 of an octave in one launch where the radii are the default ones (blur3_tile_kernel)
        float k3[3][2 * RT_MAXR + 1];
        bool fuse3 = split && nl == 3 && !chain_first;
        for (int i = 1; i <= 3 && fuse3; ++i) {
            static const int want_r[4] = {0, 5, 6, 8};
            const int n = gauss_kernel(sig[i], k3[i - 1], 2 * RT_MAXR + 1);
            if (n / 2 != want_r[i]) fuse3 = false;
        }
        if (const char* e = getenv("IMATCH_SIFT_FUSE3")) fuse3 = fuse3 && atoi(e) != 0;        // experiments
        for (int o = 0; o < o_tail && split; ++o) {
            const int rows = T.rows[o], cols = T.cols[o];
            if (fuse3) {
                float* dec = o + 1 < o_tail ? T.gauss[o + 1][0] : nullptr;
                blur3_tile_launch<5, 6, 8>(ctx->stream, T.gauss[o][0], T.gauss[o][1], T.gauss[o][2], T.gauss[o][3], dec, rows, cols,
                                           k3[0], k3[1], k3[2], batch, zs);
            }
            for (int i = 1; i <= nl && !fuse3; ++i) {
                float* dec = (i == nl && o + 1 < o_tail) ? T.gauss[o + 1][0] : nullptr;
                rc = gaussian_blur(ctx, ctx->stream, T.gauss[o][i - 1], nullptr, 0, 0, T.gauss[o][i], dec, d_tmp, rows, cols, sig[i],
                                   batch, zs, nullptr);
                if (rc) return rc;
            }
            hipStream_t sb = (three && o > 0) ? S->side2 : S->side;
            HIP_TRY(ctx, hipEventRecord(S->ev_oct[o], ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(sb, S->ev_oct[o], 0));
            for (int i = nl + 1; i < nl + 3; ++i) {
                rc = gaussian_blur(ctx, sb, T.gauss[o][i - 1], nullptr, 0, 0, T.gauss[o][i], nullptr, d_tmp, rows, cols, sig[i],
                                   batch, zs, nullptr);
                if (rc) return rc;
            }
            scan_octave(o, sb);
        }
        for (int o = 0; o < o_tail && !split; ++o) {
            const int rows = T.rows[o], cols = T.cols[o];
            hipStream_t st = o == 0 ? ctx->stream : st2;
            if (o > 0 && !have_l0)
                hipLaunchKernelGGL(decimate_kernel, dim3((cols + 63) / 64, (rows + 3) / 4, batch), dim3(256), 0, st,
                                   (const float*)T.gauss[o - 1][nl], T.cols[o - 1], T.gauss[o][0], rows, cols, zs);
            for (int i = 1; i < nl + 3; ++i) {
                // the tail kernel decimates for itself, so only octaves that stay on this path get the fused output
                float* dec = (i == nl && o + 1 < o_tail) ? T.gauss[o + 1][0] : nullptr;
                bool fz = true;
                rc = gaussian_blur(ctx, st, T.gauss[o][i - 1], nullptr, 0, 0, T.gauss[o][i], dec, d_tmp, rows, cols, sig[i], batch, zs,
                                   &fz);
                if (rc) return rc;
                if (i == nl) {
                    have_l0 = dec != nullptr && fz;
                    if (o == 0 && use_side) {   // the later octaves need nothing else of octave 0
                        HIP_TRY(ctx, hipEventRecord(S->ev_fork, ctx->stream));
                        HIP_TRY(ctx, hipStreamWaitEvent(st2, S->ev_fork, 0));
                    }
                }
            }
            scan_octave(o, st);
        }
        if (split) {   // the small octaves follow layer nl of octave o_tail - 1 on the main stream; the scans join below
            st2 = ctx->stream;
        }
        bool used_tail_split = false;
        hipStream_t st_tail = st2;   // o_tail >= 1: the tail needs layer nl of octave o_tail - 1, which precedes it on st2 (or the fork)
        bool tail_fast = o_tail < n_oct && nl == 3 &&
                         (size_t)tf_ext_rows(T.rows[o_tail]) * tf_pitch(T.cols[o_tail]) <= TF_EXT;
        TailFK tfk;
        memset(&tfk, 0, sizeof(tfk));
        for (int i = 1; i < nl + 3 && tail_fast; ++i) {
            static const int want_r[6] = {0, 5, 6, 8, 10, 13};
            float kk[2 * RT_MAXR + 1];
            const int n = gauss_kernel(sig[i], kk, 2 * RT_MAXR + 1);
            if (n / 2 != want_r[i]) tail_fast = false;
            else memcpy(tfk.k[i - 1], kk, sizeof(float) * n);
        }
        DBG_STEP("large octaves queued");
        if (tail_fast) {
            static bool attr_set[64] = {false};
            if (!attr_set[ctx->device & 63]) {
                HIP_TRY(ctx, hipFuncSetAttribute((const void*)pyramid_tail_fast_kernel<5, 6, 8, 10, 13, false>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS_BYTES));
                HIP_TRY(ctx, hipFuncSetAttribute((const void*)pyramid_tail_fast_kernel<5, 6, 8, 10, 13, true>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, TF_LDS_BYTES));
                attr_set[ctx->device & 63] = true;
            }
            static const int tail_split_env = getenv("IMATCH_SIFT_TAILSPLIT") ? atoi(getenv("IMATCH_SIFT_TAILSPLIT")) : 1;
            const int n_tail = n_oct - o_tail;
            // One workgroup per octave only on a single frame's chain (split), where the kernel is a third of the scale
            // space and little else runs: on maps and batches (use_side) it would wait beside octave 0's chip-filling
            // kernels for CUs with 155 KB of free LDS, and gains nothing measurable there. Never again on a context
            // where the hand-off once timed out.
            used_tail_split = tail_split_env && !big && !S->no_tail_split && n_tail > 1 && n_tail <= 16 &&
                              batch * n_tail <= 32;
            // test knob (read on every call): the flag of this octave is never published, which forces the time-out path
            const char* wh = getenv("IMATCH_DEBUG_TAIL_WITHHOLD");
            const int withhold = wh ? o_tail + atoi(wh) : -1;
            if (used_tail_split)
                hipLaunchKernelGGL((pyramid_tail_fast_kernel<5, 6, 8, 10, 13, true>), dim3(n_tail, 1, batch), dim3(1024),
                                   TF_LDS_BYTES, st_tail, T, o_tail, tfk, thr, d_cand, cap, d_cnt, zs, withhold);
            else
                hipLaunchKernelGGL((pyramid_tail_fast_kernel<5, 6, 8, 10, 13, false>), dim3(1, 1, batch), dim3(1024),
                                   TF_LDS_BYTES, st_tail, T, o_tail, tfk, thr, d_cand, cap, d_cnt, zs, -1);
        } else if (o_tail < n_oct) {
            TailK tk;
            memset(&tk, 0, sizeof(tk));
            for (int i = 1; i < nl + 3; ++i) {
                const int n = gauss_kernel(sig[i], tk.k[i], 2 * RT_MAXR + 1);
                if (n < 0) return imatch_set_err(ctx, IMATCH_ESIZE, "Gaussian sigma %g needs a kernel wider than %d", sig[i], 2 * RT_MAXR + 1);
                tk.r[i] = n / 2;
            }
            static bool attr_set[64] = {false};
            if (!attr_set[ctx->device & 63]) {
                HIP_TRY(ctx, hipFuncSetAttribute((const void*)pyramid_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 TAIL_LDS_BYTES));
                attr_set[ctx->device & 63] = true;
            }
            hipLaunchKernelGGL(pyramid_tail_kernel, dim3(1, 1, batch), dim3(1024), TAIL_LDS_BYTES, st_tail, T, o_tail, tk,
                               thr, d_cand, cap, d_cnt, zs);
        }
        HIP_TRY(ctx, hipGetLastError());
        if (use_side || split) {
            HIP_TRY(ctx, hipEventRecord(S->ev_join, S->side));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, S->ev_join, 0));
        }
        if (three) {
            HIP_TRY(ctx, hipEventRecord(S->ev_join2, S->side2));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, S->ev_join2, 0));
        }
        DBG_STEP("small octaves queued");
        { const int rcs = imatch_stage_end(ctx, IMATCH_STAGE_SCALE_SPACE, ctx->stream); if (rcs) return rcs; }
        (void)imatch_stage_begin(ctx, IMATCH_STAGE_KEYPOINTS, ctx->stream);
        // ---- refine + orientation are queued behind the scan without waiting for its count
        // (grid from the previous extraction's candidate count -- the kernel strides: sized by the list's CAPACITY, a batch
        // launched forty thousand workgroups of which a few hundred found a candidate)
        const int cand_hint = std::min(cap, S->cand_hint > 0 ? S->cand_hint : cap);
        hipLaunchKernelGGL(refine_kernel, dim3(std::min(std::max((cand_hint + 255) / 256, 16), 1024), 1, batch), dim3(256), 0, ctx->stream, T,
                           (const uint2*)d_cand, (const int*)d_cnt, cap, (float)P->contrast_threshold,
                           (float)P->edge_threshold, (float)P->sigma, d_ref, d_cnt + 1, zs);
        const int ref_hint = S->ref_hint > 0 ? S->ref_hint : 4096;
        hipLaunchKernelGGL(orient_kernel, dim3(std::min(std::max((ref_hint + 3) / 4, 64), 4096), 1, batch), dim3(256), 0,
                           ctx->stream, T, (const RefinedKp*)d_ref, (const int*)(d_cnt + 1), cap, d_kp, 2 * cap, d_cnt + 2, zs);
        HIP_TRY(ctx, hipGetLastError());
        // frames: sorted, deduplicated and rescaled where they lie (sort_dedupe_kernel); maps keep the host's radix sort
        KpRec* d_sorted = (KpRec*)d_ref;   // the refined list is no longer needed once the orientations are out
        static const int dev_sort_env = getenv("IMATCH_SIFT_DEVSORT") ? atoi(getenv("IMATCH_SIFT_DEVSORT")) : 1;
        DBG_STEP("refinement and orientation queued");
        const int n_sort = std::min(SORT_CAP, S->kp_hint > 0 ? S->kp_hint : 4096);   // records the device sort is launched for
        bool dev_sort = dev_sort_env && (size_t)SORT_CAP * sizeof(KpRec) <= (size_t)cap * sizeof(RefinedKp) &&
                        (S->kp_hint > 0 ? S->kp_hint <= SORT_CAP + SORT_CAP / 4 + 1024 : (size_t)h * w <= ((size_t)1 << 21));
        if (dev_sort) {
            static bool attr_set[64] = {false};
            if (!attr_set[ctx->device & 63]) {
                HIP_TRY(ctx, hipFuncSetAttribute((const void*)kp_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 SORT_LDS_BYTES));
                attr_set[ctx->device & 63] = true;
            }
            // x of a record lies in [0, 2w) (base-octave coordinates): 256 buckets over that range
            hipLaunchKernelGGL(kp_sort_kernel, dim3(1, 1, batch), dim3(1024), SORT_LDS_BYTES, ctx->stream, (const KpRec*)d_kp,
                               (const int*)(d_cnt + 2), n_sort, (float)SORT_NB / (float)bcols, d_sorted, d_cnt + 3, first_octave,
                               zs);
            HIP_TRY(ctx, hipGetLastError());
        }
        (void)imatch_stage_end(ctx, IMATCH_STAGE_KEYPOINTS, ctx->stream);
        // ---- A single frame sorted on the device: its descriptors are queued before the host has seen a count (right behind the sort; the copies below leave on the copy stream). The
        // set is allocated for as many rows as the sort was launched for (the frame's keypoints are at most that many, or
        // the sort reports -1 and the launch does nothing), the kernels read the count where the sort left it, and the
        // host fixes the set's row count after the synchronisation. The GPU used to idle from the end of the sort to the
        // host's launch (the synchronisation's wake-up, the set's allocation, the launch: 35-45 us of a 0.65 ms frame).
        static const int spec_env = getenv("IMATCH_SIFT_SPECDESC") ? atoi(getenv("IMATCH_SIFT_SPECDESC")) : 1;
        struct SpecSet {   // freed on every way out on which nobody took it over
            imatch_features* f;
            ~SpecSet() { if (f) imatch_features_free(f); }
        } spec{nullptr};
        // ---- the extraction's one synchronisation: the counters and (a guess of) the records of every image
        int kp_hint = std::min(2 * cap, S->kp_hint > 0 ? S->kp_hint : 8192);
        if (dev_sort) kp_hint = std::min(kp_hint, SORT_CAP);
        const size_t cnt_bytes = al((size_t)batch * 32);
        DBG_STEP("pinned block: reserve");
        rc = imatch_hs_reserve(ctx, cnt_bytes + (size_t)batch * kp_hint * sizeof(KpRec));
        if (rc) return rc;
        DBG_STEP("pinned block reserved");
        int* h_cnt = (int*)ctx->hs;                              // [batch][8]: candidates, refined, records, sorted records, timed-out octave + 1
        KpRec* h_kp = (KpRec*)((char*)ctx->hs + cnt_bytes);      // [batch][kp_hint]
        // ends the record copy on the second stream on every way out of this attempt
        struct CopyGuard {
            hipStream_t s;
            bool pending;
            hipError_t wait() { const hipError_t e = pending ? hipStreamSynchronize(s) : hipSuccess; pending = false; return e; }
            ~CopyGuard() { (void)wait(); }
        } rec_copy{nullptr, false};
        const bool copy_aside = dev_sort && (size_t)batch * kp_hint * sizeof(KpRec) >= ((size_t)1 << 20);
        // a single frame sorted on the device: both copies (and the copy of the equalised image the set keeps) leave on
        // the copy stream, the descriptor launches follow the sort on the main stream at once -- the two copies used to sit
        // between the sort and the descriptor kernel, 30 us of a frame in which the GPU did nothing else
        const bool spec_frame = spec_env && batch == 1 && dev_sort && S->kp_hint > 0 && !copy_aside;
        if ((copy_aside || spec_frame) && !S->copy) {
            HIP_TRY(ctx, hipStreamCreateWithFlags(&S->copy, hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_sorted, hipEventDisableTiming));
        }
        if (copy_aside || spec_frame) {
            HIP_TRY(ctx, hipEventRecord(S->ev_sorted, ctx->stream));
            HIP_TRY(ctx, hipStreamWaitEvent(S->copy, S->ev_sorted, 0));
            rec_copy.s = S->copy;
            rec_copy.pending = true;
        }
        HIP_TRY(ctx, hipMemcpy2DAsync(h_cnt, 32, d_cnt, zs, 32, (size_t)batch, hipMemcpyDeviceToHost, spec_frame ? S->copy : ctx->stream));
        if (copy_aside || spec_frame) {
            HIP_TRY(ctx, hipMemcpy2DAsync(h_kp, (size_t)kp_hint * sizeof(KpRec), d_sorted, zs, (size_t)kp_hint * sizeof(KpRec),
                                          (size_t)batch, hipMemcpyDeviceToHost, S->copy));
        } else {
            HIP_TRY(ctx, hipMemcpy2DAsync(h_kp, (size_t)kp_hint * sizeof(KpRec), dev_sort ? d_sorted : d_kp, zs,
                                          (size_t)kp_hint * sizeof(KpRec), (size_t)batch, hipMemcpyDeviceToHost, ctx->stream));
        }
        // the host waits for the copies only: the descriptor launches run while it sorts out the counts
        bool wait_on_event = false;
        if (spec_frame) {
            if (!S->ev_counts) HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_counts, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventRecord(S->ev_counts, S->copy));
            wait_on_event = true;
            rc = imatch_features_alloc(ctx, n_sort, &spec.f);
            if (rc) return rc;
            if (P->clahe) {   // the set's copy of the equalised image: on the copy stream too; the main stream (so every later
                              // reader of the copy, and the next frame's CLAHE) waits for it behind the descriptor launches
                spec.f->image = (uint8_t*)imatch_pool_alloc(ctx, (size_t)h * w);
                if (!spec.f->image) return imatch_set_err(ctx, IMATCH_ENOMEM, "image copy");
                spec.f->img_h = h; spec.f->img_w = w;
                if (!S->ev_image) HIP_TRY(ctx, hipEventCreateWithFlags(&S->ev_image, hipEventDisableTiming));
                HIP_TRY(ctx, hipMemcpyAsync(spec.f->image, d_clahe, (size_t)h * w, hipMemcpyDeviceToDevice, S->copy));
                HIP_TRY(ctx, hipEventRecord(S->ev_image, S->copy));
            }
            SegTab seg1;
            memset(&seg1, 0, sizeof(seg1));
            seg1.n = 1;
            seg1.off[1] = n_sort;
            (void)imatch_stage_begin(ctx, IMATCH_STAGE_DESCRIBE, ctx->stream);
            hipLaunchKernelGGL(describe_kernel<4>, dim3(n_sort, 1, 1), dim3(256), 0, ctx->stream, T, (const KpRec*)d_sorted, seg1,
                               first_octave, spec.f->desc, spec.f->norm, spec.f->tneg, zs, (const int*)(d_cnt + 3), (float2*)spec.f->xy);
            HIP_TRY(ctx, hipGetLastError());
            (void)imatch_stage_end(ctx, IMATCH_STAGE_DESCRIBE, ctx->stream);
            rc = imatch_features_pad_counted_dev(ctx, spec.f, d_cnt + 3);
            if (rc) return rc;
            // everything the copy stream reads (counters, records, the equalised image) is the arena's: whatever the main
            // stream runs next comes after it
            if (P->clahe) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, S->ev_image, 0));
            else HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, S->ev_counts, 0));
        }
        const bool timing = getenv("IMATCH_TIMING") != nullptr;
        const double t_a = now_ms();
        DBG_STEP("chain queued: sync for the counts");
        if (wait_on_event) HIP_TRY(ctx, hipEventSynchronize(S->ev_counts));
        else HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        DBG_STEP("counts are here");
        const double t_b = now_ms();
        int max_cand = 0, max_ref = 0, max_kp = 0, timed_out = 0;
        for (int b = 0; b < batch; ++b) {
            if (h_cnt[8 * b] < 0) timed_out = std::max(timed_out, std::max(h_cnt[8 * b + 4], 1));
            max_cand = std::max(max_cand, h_cnt[8 * b]);
            max_ref = std::max(max_ref, h_cnt[8 * b + 1]);
            max_kp = std::max(max_kp, h_cnt[8 * b + 2]);
        }
        if (timed_out) {
            // the workgroup of octave timed_out - 1 of the small-octave kernel waited about a second for the octave before
            // it: some workgroup of that launch never became resident (something else held the CUs it needs). The
            // one-workgroup walk depends on nobody: run the extraction again with it, and keep to it on this context.
            HIP_TRY(ctx, rec_copy.wait());
            if (!used_tail_split)
                return imatch_set_err(ctx, IMATCH_EHIP, "scale space: candidate counter poisoned (octave %d) without the octave hand-off in use",
                                      timed_out - 1);
            fprintf(stderr, "[imatch] scale space, small octaves: octave %d waited in vain for octave %d's layer %d; running the "
                            "extraction again with the one-workgroup walk (kept from now on for this context)\n",
                    timed_out - 1, timed_out - 2, nl);
            S->no_tail_split = true;
            continue;
        }
        if (max_cand > cap) {  // rare: grow the lists and run the batch again
            cap = max_cand + max_cand / 8;
            continue;
        }
        if (max_kp > 2 * cap) {
            cap = max_kp;
            continue;
        }
        S->cand_hint = max_cand + max_cand / 4 + 256;
        S->ref_hint = max_ref + max_ref / 4 + 256;
        S->kp_hint = max_kp + max_kp / 4 + 1024;
        std::vector<int> n_kp((size_t)batch);
        for (int b = 0; b < batch; ++b) n_kp[b] = h_cnt[8 * b + 2];
        bool refetch = false;
        if (dev_sort && max_kp > n_sort) {   // more records than the device sort was launched for: the host sorts the raw ones
            dev_sort = false;
            refetch = true;
        }
        std::vector<int> K_dev((size_t)batch, 0);
        int max_fetch = max_kp;
        if (dev_sort) {
            max_fetch = 0;
            for (int b = 0; b < batch; ++b) {
                K_dev[b] = h_cnt[8 * b + 3];
                max_fetch = std::max(max_fetch, K_dev[b]);
            }
        }
        if (refetch || max_fetch > kp_hint) {  // the guess was short: fetch everything again at the right width
            DBG_STEP("records: second fetch");
            HIP_TRY(ctx, rec_copy.wait());
            kp_hint = std::max(kp_hint, max_fetch);
            rc = imatch_hs_reserve(ctx, cnt_bytes + (size_t)batch * kp_hint * sizeof(KpRec));  // may move hs
            if (rc) return rc;
            h_kp = (KpRec*)((char*)ctx->hs + cnt_bytes);
            HIP_TRY(ctx, hipMemcpy2DAsync(h_kp, (size_t)kp_hint * sizeof(KpRec), dev_sort ? d_sorted : d_kp, zs,
                                          (size_t)kp_hint * sizeof(KpRec), (size_t)batch, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        // ---- removeDuplicatedSorted + first-octave rescale on the host (K records of 24 bytes per image);
        // the sorted records go back into the same pinned rows
        SegTab seg;
        memset(&seg, 0, sizeof(seg));
        seg.n = batch;
        std::vector<int> K_of((size_t)batch, 0);
        auto finish_image = [&](int b, std::vector<KpRec>& kps) {
            KpRec* rec = h_kp + (size_t)b * kp_hint;
            if (n_kp[b] <= 0) return;
            kps.resize((size_t)n_kp[b]);
            sort_keypoints(rec, n_kp[b], kps.data());
            size_t j = 0;
            for (size_t i = 1; i < kps.size(); ++i) {
                const KpRec &a = kps[j], &c = kps[i];
                if (a.x != c.x || a.y != c.y || a.size != c.size || a.angle != c.angle) kps[++j] = kps[i];
            }
            const int K = (int)j + 1;
            const float scale = 1.f / (float)(1 << -first_octave);
            for (int i = 0; i < K; ++i) {
                KpRec& k = kps[i];
                k.octave = (k.octave & ~255) | ((k.octave + first_octave) & 255);
                k.x *= scale; k.y *= scale; k.size *= scale;
            }
            memcpy(rec, kps.data(), (size_t)K * sizeof(KpRec));
            K_of[b] = K;
        };
        if (dev_sort) {
            for (int b = 0; b < batch; ++b) K_of[b] = K_dev[b];
        } else if (batch >= 4) {  // the images' lists are independent: sort them on the context's host workers
            imatch_parallel_for(ctx, batch, [&](int b) {
                std::vector<KpRec> kps;
                finish_image(b, kps);
            });
        } else {
            std::vector<KpRec> kps;
            for (int b = 0; b < batch; ++b) finish_image(b, kps);
        }
        const double t_c = now_ms();
        int max_k = 0;
        for (int b = 0; b < batch; ++b) {
            seg.off[b + 1] = seg.off[b] + K_of[b];
            max_k = std::max(max_k, K_of[b]);
        }
        const int Ktot = seg.off[batch];
        imatch_features* f = nullptr;
        // the set queued before the synchronisation holds this frame's rows if the sort ran (dev_sort still true) --
        // Ktot <= n_sort then; its pad rows start at Ktot, so the kNN's row padding is the usual multiple of 1024
        const bool spec_ok = spec.f && dev_sort && Ktot <= (int)spec.f->n;
        if (spec_ok) {
            f = spec.f;
            spec.f = nullptr;
            f->n = Ktot;
            f->n_pad = std::max<int64_t>(((int64_t)Ktot + IMATCH_ROW_PAD - 1) / IMATCH_ROW_PAD * IMATCH_ROW_PAD, IMATCH_ROW_PAD);
        } else {
            rc = imatch_features_alloc(ctx, Ktot, &f);
            if (rc) return rc;
        }
        f->kp = (float*)malloc(sizeof(float) * 7 * (size_t)std::max(Ktot, 1));
        auto fill_table = [&](int b) {
            const KpRec* rec = h_kp + (size_t)b * kp_hint;
            for (int i = 0; i < seg.off[b + 1] - seg.off[b]; ++i) {
                float* p = f->kp + 7 * (size_t)(seg.off[b] + i);
                p[0] = rec[i].x; p[1] = rec[i].y; p[2] = rec[i].size; p[3] = rec[i].angle; p[4] = rec[i].response;
                p[5] = (float)rec[i].octave; p[6] = -1.f;
            }
        };

        const double t_d = now_ms();
        if (timing)
            fprintf(stderr, "[imatch] extract host phases: wait for GPU %.3f ms, sort %.3f ms, alloc + keypoint table %.3f ms\n",
                    t_b - t_a, t_c - t_b, t_d - t_c);
        if (offsets)
            for (int b = 0; b <= batch; ++b) offsets[b] = seg.off[b];
        if (P->clahe && batch == 1 && !f->image) {   // (the set queued ahead of the counts has its copy already)
            f->image = (uint8_t*)imatch_pool_alloc(ctx, (size_t)h * w);
            if (!f->image) {
                imatch_features_free(f);
                return imatch_set_err(ctx, IMATCH_ENOMEM, "image copy");
            }
            f->img_h = h; f->img_w = w;
            HIP_TRY(ctx, hipMemcpyAsync(f->image, d_clahe, (size_t)h * w, hipMemcpyDeviceToDevice, ctx->stream));
        }
        // sorted records and u8 descriptors reuse the candidate / refined lists (no longer needed)
        uint8_t* d_desc = (uint8_t*)d_kp;  // cap*2*24 bytes >= K*128 needs K <= cap*0.375: checked below
        if ((size_t)max_k * 128 > (size_t)cap * 2 * sizeof(KpRec) || (size_t)max_k * sizeof(KpRec) > (size_t)cap * sizeof(RefinedKp)) {
            imatch_features_free(f);
            cap = (int)((size_t)max_k * 3 + 1024);
            continue;
        }
        if (Ktot > 0 && !spec_ok) {
            // this copy is still reading the pinned rows when the call returns: the next writer of ctx->hs waits for it
            // (imatch_hs_reserve)
            if (!dev_sort) {
                HIP_TRY(ctx, hipMemcpy2DAsync(d_sorted, zs, h_kp, (size_t)kp_hint * sizeof(KpRec), (size_t)max_k * sizeof(KpRec),
                                              (size_t)batch, hipMemcpyHostToDevice, ctx->stream));
                rc = imatch_hs_mark_busy(ctx);
                if (rc) { imatch_features_free(f); return rc; }
            }
            // Four waves per keypoint everywhere, the 236 k keypoints of an 8192^2 map included: with one wave per keypoint
            // (four keypoints of different window sizes per workgroup) the map's launch took 3.15 ms against 2.4 ms, a
            // 64-frame batch 2.2 ms against 1.7 ms -- the waves of a workgroup finish together when they share a keypoint.
            // IMATCH_DESC_WPK = 1 / 2 keeps the other forms for A/B runs.
            static const int wpk_env = getenv("IMATCH_DESC_WPK") ? atoi(getenv("IMATCH_DESC_WPK")) : 0;
            const int wpk = wpk_env ? wpk_env : 4;
            (void)imatch_stage_begin(ctx, IMATCH_STAGE_DESCRIBE, ctx->stream);
            if (wpk == 4)
                hipLaunchKernelGGL(describe_kernel<4>, dim3(max_k, 1, batch), dim3(256), 0, ctx->stream, T, (const KpRec*)d_sorted,
                                   seg, first_octave, f->desc, f->norm, f->tneg, zs, (const int*)nullptr, (float2*)f->xy);
            else if (wpk == 2)
                hipLaunchKernelGGL(describe_kernel<2>, dim3((max_k + 1) / 2, 1, batch), dim3(256), 0, ctx->stream, T,
                                   (const KpRec*)d_sorted, seg, first_octave, f->desc, f->norm, f->tneg, zs, (const int*)nullptr, (float2*)f->xy);
            else
                hipLaunchKernelGGL(describe_kernel<1>, dim3((max_k + 3) / 4, 1, batch), dim3(256), 0, ctx->stream, T,
                                   (const KpRec*)d_sorted, seg, first_octave, f->desc, f->norm, f->tneg, zs, (const int*)nullptr, (float2*)f->xy);
            HIP_TRY(ctx, hipGetLastError());
            (void)imatch_stage_end(ctx, IMATCH_STAGE_DESCRIBE, ctx->stream);
        }
        // the set's pad rows and flag (u8 rows are integral by type)
        if (!spec_ok) {
            rc = imatch_features_pad_dev(ctx, f);   // the rows themselves came from describe_kernel
            if (rc) { imatch_features_free(f); return rc; }
        }
        // the host copy of the keypoint table is written while the GPU works on the descriptors (it reads the pinned rows
        // the upload queued above is reading too)
        {
            DBG_STEP("descriptors queued: wait for the record copy");
            const hipError_t e = rec_copy.wait();
            if (e != hipSuccess) { imatch_features_free(f); return imatch_set_err(ctx, IMATCH_EHIP, "record copy: %s", hipGetErrorString(e)); }
        }
        if (batch >= 4) imatch_parallel_for(ctx, batch, fill_table);
        else
            for (int b = 0; b < batch; ++b) fill_table(b);
        f->has_xy = 1;   // describe_kernel wrote the coordinate table beside the rows
        *out = f;
        DBG_STEP("leave");
        return IMATCH_OK;
    }
    return imatch_set_err(ctx, IMATCH_ESIZE, "keypoint lists kept overflowing");
}

int imatch_sift_extract(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, const imatch_sift_params* P,
                        imatch_features** out) {
    return sift_extract(ctx, img, 0, 1, 0, h, w, stride, P, out, nullptr);
}

int imatch_sift_extract_dev(imatch_ctx* ctx, const uint8_t* img_dev, int h, int w, int stride,
                            const imatch_sift_params* P, imatch_features** out) {
    return sift_extract(ctx, img_dev, 1, 1, 0, h, w, stride, P, out, nullptr);
}

int imatch_sift_extract_batch_dev(imatch_ctx* ctx, const uint8_t* img_dev, int n_img, size_t img_pitch, int h, int w,
                                  int stride, const imatch_sift_params* P, imatch_features** out, int64_t* offsets) {
    if (img_pitch < (size_t)stride * (size_t)(h > 0 ? h : 0) && n_img > 1)
        return imatch_set_err(ctx, IMATCH_EINVAL, "image pitch smaller than one image");
    return sift_extract(ctx, img_dev, 1, n_img, img_pitch, h, w, stride, P, out, offsets);
}

int imatch_features_keypoints_xy(const imatch_features* f, const int64_t* rows, int64_t n, float* out_xy) {
    if (!f || n < 0 || (n > 0 && (!rows || !out_xy))) return IMATCH_EINVAL;
    if (n > 0 && !f->kp) return imatch_set_err(f->ctx, IMATCH_EINVAL, "this descriptor set carries no keypoints");
    for (int64_t i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] >= f->n) return imatch_set_err(f->ctx, IMATCH_EINVAL, "keypoint row %lld of %lld", (long long)rows[i], (long long)f->n);
        out_xy[2 * i] = f->kp[7 * rows[i]];
        out_xy[2 * i + 1] = f->kp[7 * rows[i] + 1];
    }
    return IMATCH_OK;
}

// keypoint records for a set that was made from descriptor rows alone (a map database loaded from disk): the host table
// and the device-side coordinate table, as an extraction leaves them -- imatch_match_localize then takes the set
int imatch_features_set_keypoints(imatch_features* f, const float* kp) {
    if (!f) return IMATCH_EINVAL;
    imatch_ctx* ctx = f->ctx;
    if (f->n > 0 && !kp) return imatch_set_err(ctx, IMATCH_EINVAL, "null keypoint table");
    if (f->n == 0) { f->has_xy = 1; return IMATCH_OK; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* table = (float*)malloc(sizeof(float) * 7 * (size_t)f->n);
    if (!table) return imatch_set_err(ctx, IMATCH_ENOMEM, "keypoint table");
    memcpy(table, kp, sizeof(float) * 7 * (size_t)f->n);
    // coordinates up through the pinned block, a slice at a time (a 20 M-row map's are 160 MB)
    const size_t slice = (size_t)1 << 20;
    int rc = imatch_hs_reserve(ctx, std::min((size_t)f->n, slice) * 8);
    if (rc) { free(table); return rc; }
    for (size_t lo = 0; lo < (size_t)f->n; lo += slice) {
        const size_t cnt = std::min(slice, (size_t)f->n - lo);
        float* h = (float*)ctx->hs;
        for (size_t i = 0; i < cnt; ++i) { h[2 * i] = table[7 * (lo + i)]; h[2 * i + 1] = table[7 * (lo + i) + 1]; }
        hipError_t e = hipMemcpyAsync(f->xy + 2 * lo, h, cnt * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { free(table); return imatch_set_err(ctx, IMATCH_EHIP, "coordinate upload: %s", hipGetErrorString(e)); }
    }
    free(f->kp);
    f->kp = table;
    f->has_xy = 1;
    return IMATCH_OK;
}

int imatch_features_download(const imatch_features* f, float* kp, float* desc) {
    if (!f) return IMATCH_EINVAL;
    imatch_ctx* ctx = f->ctx;
    if (kp) {
        if (!f->kp && f->n > 0) return imatch_set_err(ctx, IMATCH_EINVAL, "this descriptor set carries no keypoints");
        if (f->n > 0) memcpy(kp, f->kp, sizeof(float) * 7 * (size_t)f->n);
    }
    if (desc && f->n > 0) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (!f->integral) return imatch_set_err(ctx, IMATCH_EINVAL, "descriptor set is not integer valued: its rows were not kept");
        int rc = imatch_hs_reserve(ctx, (size_t)f->n * 128);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->hs, f->desc, (size_t)f->n * 128, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const int8_t* b = (const int8_t*)ctx->hs;   // stored as value - 128
        for (size_t i = 0; i < (size_t)f->n * 128; ++i) desc[i] = (float)((int)b[i] + 128);
    }
    return IMATCH_OK;
}

int imatch_features_image(const imatch_features* f, uint8_t* out, int h, int w) {
    if (!f || !out) return IMATCH_EINVAL;
    imatch_ctx* ctx = f->ctx;
    if (!f->image) return imatch_set_err(ctx, IMATCH_EINVAL, "this descriptor set carries no image");
    if (h != f->img_h || w != f->img_w) return imatch_set_err(ctx, IMATCH_EINVAL, "image is %dx%d", f->img_w, f->img_h);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, f->image, (size_t)h * w, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IMATCH_OK;
}

int imatch_sift_debug_octaves(imatch_ctx* ctx) { return (ctx && ctx->sift) ? ctx->sift->T.n_oct : 0; }

int imatch_sift_debug_layer(imatch_ctx* ctx, int kind, int o, int i, float* out, int* rows, int* cols) {
    if (!ctx || !ctx->sift || !ctx->sift->arena) return imatch_set_err(ctx, IMATCH_EINVAL, "no extraction yet");
    const PyrTable& T = ctx->sift->T;
    if (o < 0 || o >= T.n_oct || i < 0 || i >= (kind == 0 ? T.nl + 3 : T.nl + 2))
        return imatch_set_err(ctx, IMATCH_EINVAL, "no such layer");
    if (rows) *rows = T.rows[o];
    if (cols) *cols = T.cols[o];
    if (out) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const size_t n = (size_t)T.rows[o] * T.cols[o];
        HIP_TRY(ctx, hipMemcpyAsync(out, T.gauss[o][i], n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (kind == 1) {   // DoG layers are not stored: the same f32 subtraction the kernels make
            std::vector<float> hi(n);
            HIP_TRY(ctx, hipMemcpyAsync(hi.data(), T.gauss[o][i + 1], n * 4, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (size_t q = 0; q < n; ++q) out[q] = hi[q] - out[q];
        }
    }
    return IMATCH_OK;
}

// rows [row0, row0 + nrows) of Gaussian layer i of octave o of the last extraction (a 32768^2 layer is 4.3 GB: the
// full-size tests look at it in bands)
int imatch_sift_debug_layer_rows(imatch_ctx* ctx, int o, int i, int row0, int nrows, float* out) {
    if (!ctx || !ctx->sift || !ctx->sift->arena) return imatch_set_err(ctx, IMATCH_EINVAL, "no extraction yet");
    const PyrTable& T = ctx->sift->T;
    if (o < 0 || o >= T.n_oct || i < 0 || i >= T.nl + 3) return imatch_set_err(ctx, IMATCH_EINVAL, "no such layer");
    if (!out || row0 < 0 || nrows < 0 || row0 + nrows > T.rows[o]) return imatch_set_err(ctx, IMATCH_EINVAL, "rows outside the layer");
    if (nrows == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, T.gauss[o][i] + (size_t)row0 * T.cols[o], (size_t)nrows * T.cols[o] * 4,
                                hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IMATCH_OK;
}

}  // extern "C"
