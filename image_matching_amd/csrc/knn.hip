// Brute-force kNN (k = 2) on int8 MFMA with the top-2 selection fused into the epilogue.
// Replaces cv2.BFMatcher(NORM_L2).knnMatch(k=2) and the ratio loop of
// match_finder.find_matches (reference match_finder.py:177-197).
//
// Formulation: SIFT descriptors are integers 0..255. A row is stored as x' = x - 128 (int8), which leaves every
// difference, so every distance, as it was: ssd(q,t) = |q'|^2 + |t'|^2 - 2 q'.t'. v_mfma_i32_16x16x64_i8 forms q'.t' in
// exact 32-bit integer arithmetic at twice the multiply-adds per clock of the bf16 instruction (MI355X_MICROARCH.md,
// matrix cores: "the cycles of the BF16 form of the same MxN at 2x the K"): two instructions per 16x16 block of 128-d
// rows instead of four, half the bytes per row in HBM, L2 and LDS, half the registers for the resident queries. (Rounds
// 1 and 2 ran this kernel on bf16 operands, which are also exact for these values; git history has that form.)
//
// Tiling (wave64): the TRAIN rows are the A operand (a lane sees rows 4*(lane >> 4) .. +3 of a 16-row tile in its four
// accumulator registers), the QUERIES are the B operand (query lane & 15 of a 16-query block sits on the lane), so a
// query's running top-2 is kept by four lanes in REGISTERS (four partial lists per train chunk, merged at the end). A wave
// holds the fragments of 16*QB queries for the whole kernel; train rows stream HBM -> LDS (global_load_lds, 16 B per
// lane, XOR-swizzled through the source address) and are read as A fragments by ds_read_b128.
//
// Per block the accumulator starts at C = T' >> 1 with T' = -|t'|^2 of the result row (i.e. -ceil(|t'|^2 / 2)), so the chain
// ends with acc = q'.t' + (T' >> 1) and the exact partial distance is x = |t'|^2 - 2 q'.t' = -2 acc - (T' & 1). A result can
// enter the query's top-2 only if x <= T, T = an upper bound of the query's final second-best; whatever the parity bit,
// that needs acc >= G = -((T + 1) >> 1): the hot epilogue is a 4-way max and ONE compare per block; the insertion path
// (wave-uniform, rare) recomputes x exactly with the parity bit re-read from LDS. T is shared chip-wide: waves publish
// their bounds with atomicMin into gthr once per exchange and fetch the minimum back (asynchronously, into LDS), so short
// train chunks do not keep loose bounds. The chain of block u+1 is issued before the epilogue of block u (two accumulator
// sets), so the VALU work hides under the MFMAs.
//
// Pad rows (index >= n) are zero bytes with T' = -2^24: their x = 2^24 lies above every real value (< 2^23), so a pad only
// ever sits in a slot that no real row has claimed, and the merge drops entries whose index is not below n.
#include "common.h"
#include <float.h>
#include <limits.h>
#include <math.h>
#include <mutex>

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// Asynchronous global -> LDS copy (LDS-DMA), written as inline asm so that the compiler does not
// track it: hipcc otherwise drains vmcnt(0) before every LDS access that follows a global_load_lds,
// which serialises the prefetch with the compute it is meant to overlap. Completion is handled by the
// explicit s_waitcnt vmcnt(0) + barrier at the top of every stage. base: wave-uniform global pointer;
// voff: per-lane byte offset; lds_addr: wave-uniform LDS byte address (lane i lands at lds_addr + i*size).
__device__ __forceinline__ void glds16(const void* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void glds4(const void* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void glds4_sc1(const void* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_addr) : "memory");
}
// lane id recomputed on the spot (volatile: not CSE'd into a long-lived register that would be spilled;
// spill reloads are vector-memory operations and would drain the LDS-DMA queue)
__device__ __forceinline__ unsigned lane_now() {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ const void* uniform_ptr(const void* p) {
    const unsigned long long b = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return (const void*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(LDS_AS const char*)p);
}

typedef int __attribute__((ext_vector_type(4))) i32x4;

#ifndef KNN_TILE_PAIRS
#define KNN_TILE_PAIRS 1   // 0: the one-tile main loop (kept for A/B builds)
#endif

#ifndef KNN_XM
#define KNN_XM 7           // the large form exchanges bounds every KNN_XM + 1 stages once a chunk is no longer young
#endif

#define KNN_INF 0x7f7f7f7f            // "no bound yet" / an empty list slot (what hipMemset 0x7f leaves in a word)
#define KNN_PAD_TNEG (-(1 << 24))     // T' of a pad row

// acc >= bound_g(T) is necessary for x = -2 acc - parity <= T
__device__ __forceinline__ int bound_g(int T) { return -((T + 1) >> 1); }

#ifdef KNN_TRACE   // development only (tools/dbg/knn_trace.py): per-workgroup wall-clock stamps (100 MHz)
#define KNN_TRACE_SLOTS 40
__device__ long long knn_trace[2048 * KNN_TRACE_SLOTS];
#ifndef KT_AGE0
#define KT_AGE0 0     // first stage (age in its chunk) whose stamps are kept
#endif
#define KT(slot) do { if (threadIdx.x == 0 && blockIdx.x < 2048 && (slot) >= 0 && (slot) < KNN_TRACE_SLOTS) \
        knn_trace[blockIdx.x * KNN_TRACE_SLOTS + (slot)] = (long long)wall_clock64(); } while (0)
extern "C" int imatch_dbg_knn_trace(long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(knn_trace), sizeof(long long) * 2048 * KNN_TRACE_SLOTS);
}
#else
#define KT(slot) do {} while (0)
#endif

// QB blocks of 16 queries per wave (8: 128 queries, 512 per workgroup; 4: 256 per workgroup), TN train rows per LDS stage,
// XM + 1 stages between bound exchanges once a chunk is no longer young.
// SHORT: the 512-query form for a few query tiles against chunks of a few dozen stages (a frame against a large map):
// the 256-query form's first exchange (merged publishing, waited for) without its key registers.
template <int QB, int TN, int XM, bool SHORT = false>
__global__ void __launch_bounds__(256, TN <= 128 ? 3 : 2)   // three workgroups per CU (40 KB of LDS each, <= 168 VGPRs)
knn2_kernel8(const uint4* __restrict__ Qd, const uint4* __restrict__ Td, const int* __restrict__ tneg,
             int nQt, int nChunks, int nStages, int64_t Qpad, int2* __restrict__ pb, int2* __restrict__ pi,
             int* __restrict__ gthr, int young, unsigned young_wg, int stage0, int list0, int rot) {
    static_assert(QB % 4 == 0, "bounds are fetched 64 queries at a time");
    static_assert(TN <= 128 || QB > 4, "the first-stage keys hold a 7-bit row");
    static_assert(TN % 64 == 0, "the T' words of a stage are loaded 64 at a time");
    static_assert((TN / 16) % 2 == 0 && QB % 2 == 0, "the main loop walks pairs of tiles with two accumulator sets");
    // the key form of a young chunk's first stage (below) costs 2 * QB registers and is worth them where chunks are short
    constexpr bool KEY0 = QB <= 4;
    constexpr bool MERGED = KEY0 || SHORT;   // lane-merged publishing; the second stage waits for the first exchange
    constexpr int STAGE_U4 = TN * 8 + TN / 4;    // uint4 per stage: TN rows x 128 B, then TN int32 (T' = -|t'|^2)
    constexpr int NT = TN / 16;                  // 16-row tiles per stage
    constexpr int TH_F = 3 * (QB / 4) * 64;      // fetched bound words per wave: [3][QB/4][64 queries]
    __shared__ uint4 lds[2 * STAGE_U4 + (4 * TH_F) / 4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4, grp = q >> 1;
    int* const thl = (int*)(lds + 2 * STAGE_U4) + wave * TH_F;

    // workgroups b and b+8 share an XCD (its L2): same train chunk, different query tiles
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int chunk = (j / nQt) * 8 + xcd;
    KT(0);
    if (chunk >= nChunks) return;
    // Every XCD walks the query tiles from a different start: a tile then meets its chunks one XCD after the other
    // instead of on all eight at once, so only its FIRST chunk is scanned with loose bounds -- the others inherit the
    // chip-wide words.
    const int qt = (j % nQt + (rot ? (xcd * nQt) >> 3 : 0)) % nQt;
    const int s_begin = stage0 + (int)((int64_t)nStages * chunk / nChunks);
    const int s_end = stage0 + (int)((int64_t)nStages * (chunk + 1) / nChunks);

    // resident query fragments: lane holds bytes 16 (4 t + q) .. +15 of row r of block qb (k-step t = 0, 1). The train
    // fragment below takes the SAME bytes of its row on the same lane, so the sum over k pairs equal dimensions whatever
    // order the instruction walks a lane's sixteen bytes in.
    const int64_t qrow0 = (int64_t)qt * (4 * 16 * QB) + wave * (16 * QB);
    i32x4 qf[QB][2];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            uint4 u = Qd[(qrow0 + qb * 16 + r) * 8 + 4 * t + q];
            qf[qb][t] = __builtin_bit_cast(i32x4, u);
        }

    const int* gq = gthr + qrow0;                 // wave-uniform base of this wave's bound words
    int G[QB], b1[QB], b2[QB];
    int i1[QB], i2[QB];
    // fetched bounds of block qb, word a: thl[(a * (QB/4) + (qb >> 2)) * 64 + (qb & 3) * 16 + r]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        b1[qb] = KNN_INF; b2[qb] = KNN_INF; i1[qb] = -1; i2[qb] = -1;
        const int t0 = __hip_atomic_load(gq + qb * 16 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int at = (qb >> 2) * 64 + (qb & 3) * 16 + r;
        thl[at] = t0;
        thl[(QB / 4) * 64 + at] = KNN_INF;
        thl[2 * (QB / 4) * 64 + at] = KNN_INF;
        G[qb] = bound_g(t0);
    }

    // Stage loader. The LDS image is lane-linear (a wave's piece = 1 KiB = 8 rows of 8 chunks); the swizzle
    // chunk' = chunk ^ (row & 7) is applied to the per-lane SOURCE address. slot = i*256 + wave*64 + lane ->
    // row = i*32 + wave*8 + lane/8, so every piece is (uniform base) + (one per-lane byte offset that never changes).
    const unsigned lds0 = lds_addr_of(lds);
    auto stage = [&](int s, int buf) {
        const unsigned ln = lane_now();
        const unsigned r0 = (unsigned)wave * 8u + (ln >> 3);
        const unsigned lane_off = (r0 * 8 + ((ln & 7) ^ (r0 & 7))) * 16;   // bytes
        const char* src = (const char*)(Td + (int64_t)s * TN * 8);
#pragma unroll
        for (int i = 0; i < (TN * 8) / 256; ++i)
            glds16(uniform_ptr(src + i * 4096), lane_off, lds0 + (buf * STAGE_U4 + i * 256 + wave * 64) * 16);
        if (wave < TN / 64)
            glds4(uniform_ptr(tneg + (int64_t)s * TN + wave * 64), ln * 4,
                  lds0 + (buf * STAGE_U4 + TN * 8) * 16 + wave * 256);
    };

    // Bound exchange. take: fold the fetched chip-wide words into G. publish: the wave's own bounds -- in the MERGED
    // forms (short chunks, a frame against a map) merged over the four lanes that share a query first, one atomic per
    // word, query and wave instead of four: at a chunk's first exchange every lane has something to say, and forty
    // workgroups per query tile say it to the same addresses (11.6 us -> 1.4 us for that exchange at 1500 x 59k).
    auto take_bounds = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int at = (qb >> 2) * 64 + (qb & 3) * 16 + r;
            const int tg = thl[at], ta = thl[(QB / 4) * 64 + at], tb = thl[2 * (QB / 4) * 64 + at];
            // Chip-wide bound on the query's second-best: the least second-best any list holds, or -- rows of lists
            // q = 0,1 and of lists q = 2,3 are disjoint -- the larger of the two groups' least BESTS (two rows lie below
            // it). The second form matters: the two nearest rows rarely sit in the same list.
            G[qb] = max(G[qb], bound_g(min(tg, max(ta, tb))));
        }
    };
    auto publish_bounds = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int at = (qb >> 2) * 64 + (qb & 3) * 16 + r;
            const int tg = thl[at];
            const int tgrp = thl[(1 + grp) * (QB / 4) * 64 + at];
            if (MERGED) {
                int m2 = min(b2[qb], __shfl_xor(b2[qb], 16));
                m2 = min(m2, __shfl_xor(m2, 32));                        // least second-best of the four lists
                const int m1 = min(b1[qb], __shfl_xor(b1[qb], 16));     // least best of the group's two lists
                if (q == 0 && m2 < tg) atomicMin(const_cast<int*>(gq) + qb * 16 + r, m2);
                if ((q & 1) == 0 && m1 < tgrp) atomicMin(const_cast<int*>(gq) + (1 + grp) * Qpad + qb * 16 + r, m1);
            } else {   // long chunks exchange rarely and have little to say: every lane for itself, no shuffles
                if (b2[qb] < tg) atomicMin(const_cast<int*>(gq) + qb * 16 + r, b2[qb]);
                if (b1[qb] < tgrp) atomicMin(const_cast<int*>(gq) + (1 + grp) * Qpad + qb * 16 + r, b1[qb]);
            }
        }
    };
    auto fetch_bounds = [&]() {   // sc1: from memory, not from this CU's or XCD's caches
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int jj = 0; jj < QB / 4; ++jj)
                glds4_sc1(uniform_ptr(gq + a * Qpad + jj * 64), lane_now() * 4, lds_addr_of(thl) + (a * (QB / 4) + jj) * 256);
    };

    KT(1);
    if (s_begin < s_end) stage(s_begin, 0);
    for (int s = s_begin; s < s_end; ++s) {
        const int buf = (s - s_begin) & 1;
        KT(4 + 3 * (s - s_begin - KT_AGE0));   // stamps per stage: top, own loads landed, barrier released
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        KT(5 + 3 * (s - s_begin - KT_AGE0));
        __syncthreads();
        KT(6 + 3 * (s - s_begin - KT_AGE0));
        const int age = s - s_begin;
        // every stage while the chunk is young -- only workgroups of the first rounds start from loose bounds, later ones
        // inherit the chip-wide words -- then XM + 1 stages between bound exchanges
        const bool young_now = age < young && blockIdx.x < young_wg;
        const bool exchange = young_now || (age & XM) == 0;
        // A young chunk's second stage: the words every workgroup published after its first stage are fetched and WAITED
        // for (about a microsecond), or this stage too would run on the wave's own bounds, i.e. through the insertion path
        if (MERGED && age == 1 && young_now && young > 1) {
            fetch_bounds();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (exchange) {
            take_bounds();
            publish_bounds();
        }
        // asynchronous loads: the next stage, and the chip-wide words for the next exchange
        if (s + 1 < s_end) stage(s + 1, buf ^ 1);
        if (exchange) fetch_bounds();
        const unsigned lf = lane_now();
        const unsigned frag_off = (lf & 15) * 128u + ((((lf >> 4) ^ lf) & 7) << 4);
        const uint4* img = lds + buf * STAGE_U4;
        i32x4 af[2], tin, traw;
        // chunk (4t + q) ^ (r & 7) = ((q ^ r) & 7) ^ 4t: one per-lane address XOR an immediate per k-step
        auto load_tile = [&](int tile) {
            const unsigned a0 = (unsigned)tile * 2048u + frag_off;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint4 u = *(const uint4*)((const char*)img + (a0 ^ (unsigned)(t << 6)));
                af[t] = __builtin_bit_cast(i32x4, u);
            }
            const uint4 u = img[TN * 8 + tile * 4 + (int)(lf >> 4)];   // T' of rows 4q .. 4q+3 of the tile
            traw = __builtin_bit_cast(i32x4, u);
            tin = traw >> 1;
        };
        auto chain = [&](int qb) {
            i32x4 acc = tin;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(af[t], qf[qb][t], acc, 0, 0, 0);
            return acc;
        };
        // Insertion path (rare, wave-uniform entry). Rows are visited in ascending train index with strict <: OpenCV's
        // lowest-index tie-break.
        auto insert_block = [&](const i32x4& a, int qb, int tile) {
            {
                int rowbase = s * TN + tile * 16 + 4 * q;
                asm volatile("" : "+v"(rowbase));
                const uint4 tw = img[TN * 8 + tile * 4 + q];   // the rows' parity bits (the tile's T' words again)
                const int par[4] = {(int)(tw.x & 1u), (int)(tw.y & 1u), (int)(tw.z & 1u), (int)(tw.w & 1u)};
                const int g = G[qb];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int v = a[reg];
                    const bool pass = v >= g;
                    if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;
                    const int x = -2 * v - par[reg];
                    const int idx = rowbase + reg;
                    const bool lt1 = pass && x < b1[qb];
                    const bool lt2 = pass && !lt1 && x < b2[qb];
                    b2[qb] = lt1 ? b1[qb] : (lt2 ? x : b2[qb]);
                    i2[qb] = lt1 ? i1[qb] : (lt2 ? idx : i2[qb]);
                    b1[qb] = lt1 ? x : b1[qb];
                    i1[qb] = lt1 ? idx : i1[qb];
                }
                G[qb] = max(g, bound_g(b2[qb]));
            }
        };
        auto epilogue = [&](const i32x4& a, int qb, int tile) {
            const int m = max(max(a[0], a[1]), max(a[2], a[3]));
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= G[qb]) != 0, 0))   // rare: keep the hot path the fall-through
                insert_block(a, qb, tile);
        };

#if KNN_TILE_PAIRS
        // (not in the short-chunk form: with the merged publishing's registers on top it would spill 44 of them)
        if (!SHORT && !(KEY0 && age == 0 && young_now)) {
            // Main loop on PAIRS OF TILES: a query block meets tiles 2p and 2p+1 back to back, and their eight results per
            // lane are examined by ONE compare and one branch (same queries, same bound) -- 2.5 vector instructions per
            // block instead of 3 and half the branches, which is where this kernel's time goes (profiles/r2e_pmc_knn_mfma.md);
            // the four MFMAs of a step alternate between the two tiles, so no MFMA follows the one it depends on.
            i32x4 afA[2], afB[2], tinA, tinB;
            auto load_pair = [&](int tp) {
                const unsigned a0 = (unsigned)tp * 4096u + frag_off;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    uint4 u = *(const uint4*)((const char*)img + (a0 ^ (unsigned)(t << 6)));
                    afA[t] = __builtin_bit_cast(i32x4, u);
                    uint4 w = *(const uint4*)((const char*)img + ((a0 + 2048u) ^ (unsigned)(t << 6)));
                    afB[t] = __builtin_bit_cast(i32x4, w);
                }
                const uint4 ua = img[TN * 8 + tp * 8 + (int)(lf >> 4)], ub = img[TN * 8 + tp * 8 + 4 + (int)(lf >> 4)];
                tinA = __builtin_bit_cast(i32x4, ua) >> 1;
                tinB = __builtin_bit_cast(i32x4, ub) >> 1;
            };
            auto chain2 = [&](int qb, i32x4 (&x)[2]) {
                x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afA[0], qf[qb][0], tinA, 0, 0, 0);
                x[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afB[0], qf[qb][0], tinB, 0, 0, 0);
                x[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afA[1], qf[qb][1], x[0], 0, 0, 0);
                x[1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(afB[1], qf[qb][1], x[1], 0, 0, 0);
            };
            auto epilogue2 = [&](const i32x4 (&x)[2], int qb, int tp) {
                int m = max(max(x[0][0], x[0][1]), max(x[0][2], x[0][3]));
                m = max(max(m, x[1][0]), max(x[1][1], max(x[1][2], x[1][3])));
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(m >= G[qb]) != 0, 0)) {
                    insert_block(x[0], qb, 2 * tp);        // ascending rows: the lower tile first
                    insert_block(x[1], qb, 2 * tp + 1);
                }
            };
            i32x4 P[2][2];
            load_pair(0);
            chain2(0, P[0]);
#pragma unroll 1
            for (int tp = 0; tp < NT / 2; ++tp) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const int cs = qb & 1, ns = cs ^ 1;      // QB is even: a tile pair's first block is always in set 0
                    if (qb + 1 < QB) chain2(qb + 1, P[ns]);
                    else if (tp + 1 < NT / 2) {               // the pair's last chains are out: the next pair's operands
                        load_pair(tp + 1);
                        chain2(0, P[ns]);
                    }
                    epilogue2(P[cs], qb, tp);
                }
            }
            if (MERGED && age == 0 && young_now && young > 1) publish_bounds();
            continue;
        }
#endif
        load_tile(0);
        i32x4 acc0 = chain(0), acc1;
        if (KEY0 && age == 0 && young_now) {
            // A young chunk's FIRST stage has no bound to test against: nearly every block would take the insertion path
            // (60 vector instructions). Its results go through a branch-free top-2 of 32-bit KEYS instead -- x + 2^23
            // (below 2^24 for a real row) above the row's 7-bit place in the stage, so key order is (value, row) order,
            // lowest row first on ties -- a handful of instructions per result; the lists are empty before this stage, so
            // its two best keys simply become them. A pad row's x = 2^24 gives a key of 2^31 or more: never taken.
            unsigned k1[QB], k2[QB];
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) { k1[qb] = 0xffffffffu; k2[qb] = 0xffffffffu; }
            i32x4 pc;   // 2^23 - parity of the CURRENT tile's rows (load_tile(tile + 1) runs before the tile's last epilogues)
            auto epilogue_key = [&](const i32x4& a, int qb, int tile) {
                const unsigned rowl = (unsigned)tile * 16u + 4u * (unsigned)q;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const unsigned u = (unsigned)(pc[reg] - 2 * a[reg]);
                    const unsigned key = (u << 7) | (rowl + reg);
                    const unsigned t = max(k1[qb], key);
                    k1[qb] = min(k1[qb], key);
                    k2[qb] = min(k2[qb], t);
                }
            };
#pragma unroll 1
            for (int tile = 0; tile < NT; ++tile) {
                pc = 0x800000 - (traw & 1);
#pragma unroll
                for (int qb = 0; qb < QB; qb += 2) {
                    acc1 = chain(qb + 1);
                    if (qb == QB - 2 && tile + 1 < NT) load_tile(tile + 1);
                    epilogue_key(acc0, qb, tile);
                    if (qb + 2 < QB) acc0 = chain(qb + 2);
                    else if (tile + 1 < NT) acc0 = chain(0);
                    epilogue_key(acc1, qb + 1, tile);
                }
            }
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (k1[qb] < 0x80000000u) { b1[qb] = (int)(k1[qb] >> 7) - 0x800000; i1[qb] = s * TN + (int)(k1[qb] & 127u); }
                if (k2[qb] < 0x80000000u) { b2[qb] = (int)(k2[qb] >> 7) - 0x800000; i2[qb] = s * TN + (int)(k2[qb] & 127u); }
                G[qb] = max(G[qb], bound_g(b2[qb]));
            }
        } else {
#pragma unroll 1
            for (int tile = 0; tile < NT; ++tile) {
#pragma unroll
                for (int qb = 0; qb < QB; qb += 2) {
                    // block qb+1 of this tile is issued, then block qb is examined
                    acc1 = chain(qb + 1);
                    if (qb == QB - 2 && tile + 1 < NT) load_tile(tile + 1);  // the tile's last chain is out
                    epilogue(acc0, qb, tile);
                    if (qb + 2 < QB) acc0 = chain(qb + 2);
                    else if (tile + 1 < NT) acc0 = chain(0);                  // first block of the next tile
                    epilogue(acc1, qb + 1, tile);
                }
            }
        }
        if (MERGED && age == 0 && young_now && young > 1) publish_bounds();   // read back, and waited for, at the top of the next stage
    }

    KT(2);
    // The four lanes that hold a query's lists (row quarters, disjoint rows) merge them by two exchange rounds, in
    // (value, train index) order -- equal values are true ties and the lower index ranks first, as the ascending scan
    // would have it; one partial list per chunk is written.
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const int ob1 = __shfl_xor(b1[qb], off), ob2 = __shfl_xor(b2[qb], off);
            const int oi1 = __shfl_xor(i1[qb], off), oi2 = __shfl_xor(i2[qb], off);
            const bool o_first = ob1 < b1[qb] || (ob1 == b1[qb] && (unsigned)oi1 < (unsigned)i1[qb]);
            // the runner-up: the loser of the first comparison against the winner's own second
            const int w1 = o_first ? ob1 : b1[qb], l1 = o_first ? b1[qb] : ob1, w2 = o_first ? ob2 : b2[qb];
            const int wi1 = o_first ? oi1 : i1[qb], li1 = o_first ? i1[qb] : oi1, wi2 = o_first ? oi2 : i2[qb];
            const bool l_second = l1 < w2 || (l1 == w2 && (unsigned)li1 < (unsigned)wi2);
            b1[qb] = w1; i1[qb] = wi1;
            b2[qb] = l_second ? l1 : w2; i2[qb] = l_second ? li1 : wi2;
        }
    }
    // (the lane's place recomputed here: kept in a register across the stage loop it was the one value the allocator spilled)
    const unsigned lz = lane_now();
    if ((lz >> 4) == 0) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int64_t o = (int64_t)(list0 + chunk) * Qpad + qrow0 + qb * 16 + (int)(lz & 15u);
            pb[o] = make_int2(b1[qb], b2[qb]);
            pi[o] = make_int2(i1[qb], i2[qb]);
        }
    }
    KT(3);
}


__device__ inline void top2_insert(int& s1, int& j1, int& s2, int& j2, int s, int j) {
    if (s < s1 || (s == s1 && j < j1)) {
        s2 = s1; j2 = j1; s1 = s; j1 = j;
    } else if (s < s2 || (s == s2 && j < j2)) {
        s2 = s; j2 = j;
    }
}

// merges the partial lists of one query (entries of pad rows, index >= nt, dropped), adds |q'|^2, writes (ssd1, idx1, ssd2, idx2).
// 256 threads = 16 queries x 16 list slices; the slices are combined through LDS.
__global__ void __launch_bounds__(256)
knn2_finish_kernel(const int2* __restrict__ pb, const int2* __restrict__ pi, int nLists, int64_t Qpad,
                   const int* __restrict__ qnorm, int64_t nq, int64_t nt, int idx_offset, int4* __restrict__ out,
                   int* __restrict__ rearm, int64_t n_rearm) {
    __shared__ int4 part[16][16];
    // the launch before this one is done with its bound words: they go back to "no bound" for the context's next launch
    // (the context's own block, see knn2_run: no clear in front of the next kNN)
    if (rearm)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_rearm; i += (int64_t)gridDim.x * 256) rearm[i] = 0x7f7f7f7f;
    const int ql = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int64_t q = (int64_t)blockIdx.x * 16 + ql;
    int s1 = INT_MAX, s2 = INT_MAX, j1 = INT_MAX, j2 = INT_MAX;
    if (q < nq) {
        for (int l0 = slice; l0 < nLists; l0 += 64) {   // four lists' loads in flight (a frame's launch has 40 to 112 lists)
            int2 b[4];
            int2 i[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int l = min(l0 + 16 * u, nLists - 1);
                b[u] = pb[(int64_t)l * Qpad + q];
                i[u] = pi[(int64_t)l * Qpad + q];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (l0 + 16 * u >= nLists) break;
                if (i[u].x >= 0 && i[u].x < nt) top2_insert(s1, j1, s2, j2, b[u].x, i[u].x);
                if (i[u].y >= 0 && i[u].y < nt) top2_insert(s1, j1, s2, j2, b[u].y, i[u].y);
            }
        }
    }
    part[slice][ql] = make_int4(s1, j1, s2, j2);
    __syncthreads();
    if (slice != 0 || q >= nq) return;
    for (int k = 1; k < 16; ++k) {
        const int4 e = part[k][ql];
        if (e.y != INT_MAX) top2_insert(s1, j1, s2, j2, e.x, e.y);
        if (e.w != INT_MAX) top2_insert(s1, j1, s2, j2, e.z, e.w);
    }
    const int qn = qnorm[q];
    int4 o;
    o.x = (j1 == INT_MAX) ? INT_MAX : s1 + qn;
    o.y = (j1 == INT_MAX) ? -1 : j1 + idx_offset;
    o.z = (j2 == INT_MAX) ? INT_MAX : s2 + qn;
    o.w = (j2 == INT_MAX) ? -1 : j2 + idx_offset;
    out[q] = o;
}

__global__ void merge_top2_kernel(const int4* __restrict__ in, int nLists, int64_t nq, int4* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    int s1 = INT_MAX, s2 = INT_MAX, j1 = INT_MAX, j2 = INT_MAX;
    for (int l = 0; l < nLists; ++l) {
        const int4 e = in[(int64_t)l * nq + q];
        if (e.y >= 0) top2_insert(s1, j1, s2, j2, e.x, e.y);
        if (e.w >= 0) top2_insert(s1, j1, s2, j2, e.z, e.w);
    }
    int4 o;
    o.x = (j1 == INT_MAX) ? INT_MAX : s1;
    o.y = (j1 == INT_MAX) ? -1 : j1;
    o.z = (j2 == INT_MAX) ? INT_MAX : s2;
    o.w = (j2 == INT_MAX) ? -1 : j2;
    out[q] = o;
}

// Lowe ratio test with the reference's float semantics (match_finder.py:193): the f32-rounded
// distances are widened to double, the product is a double. Survivors are compacted in query
// order by one workgroup (a block-wide exclusive scan per 1024 queries).
// One workgroup per segment (= per image of a batch; a single call is one segment): survivors of
// queries [off[b], off[b+1]) are written from position off[b] on, with query indices local to the segment.
__global__ void __launch_bounds__(1024)
ratio_kernel(const int4* __restrict__ in, imatch_segtab seg, double ratio, int* __restrict__ gq, int* __restrict__ gt,
             float* __restrict__ gd, unsigned long long* __restrict__ n_good) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t q_lo = seg.off[blockIdx.x], q_hi = seg.off[blockIdx.x + 1];
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int64_t q0 = q_lo; q0 < q_hi; q0 += 1024) {
        const int64_t q = q0 + threadIdx.x;
        bool good = false;
        int4 e = make_int4(0, -1, 0, -1);
        float d1 = 0.f;
        if (q < q_hi) {
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
        int64_t off = q_lo + base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (good) {
            gq[off + within] = (int)(q - q_lo);
            gt[off + within] = e.y;
            gd[off + within] = d1;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base_s += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_good[blockIdx.x] = (unsigned long long)base_s;
}

// ---- descriptor upload: u8 / f32 -> int8 rows (x - 128) + squared norms of the shifted rows -----------
// one wave per row; lane handles 2 values
template <typename T, bool SEG = false>
__global__ void pack_rows_kernel(const T* __restrict__ src, int64_t n, int64_t n_pad, int8_t* __restrict__ dst,
                                 int* __restrict__ norm, int* __restrict__ tneg, int* __restrict__ flag,
                                 imatch_segtab seg = imatch_segtab(), size_t zs = 0) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_pad) return;
    float a = 128.f, b = 128.f;   // a pad row: zero bytes
    bool bad = false;
    if (row < n) {
        const T* s = src + row * 128;
        if (SEG) {  // the row's image (wave-uniform scan of at most IMATCH_MAX_BATCH offsets)
            int im = 0;
            while (im + 1 < seg.n && row >= seg.off[im + 1]) ++im;
            s = (const T*)((const char*)src + (size_t)im * zs) + (row - seg.off[im]) * 128;
        }
        a = (float)s[2 * lane];
        b = (float)s[2 * lane + 1];
        // integer in 0..255? decided on the bit patterns: this file is built with -fno-honor-nans, under which a float
        // compare may not be relied on to reject a NaN
        auto is_u8 = [](float v) {
            const int i = (int)fminf(fmaxf(v, -1.f), 256.f);   // NaN -> one of the bounds or 0, never a match below
            return i >= 0 && i <= 255 && (__float_as_uint((float)i) == __float_as_uint(v) || __float_as_uint(v) == 0x80000000u);
        };
        bad = !is_u8(a) || !is_u8(b);
    }
    // (a set with other values is flagged and refused by the kNN entry points; its rows hold the clamped values)
    const int ia = (int)fminf(fmaxf(a, 0.f), 255.f) - 128, ib = (int)fminf(fmaxf(b, 0.f), 255.f) - 128;
    ((uint16_t*)dst)[row * 64 + lane] = (uint16_t)((uint32_t)(ia & 0xff) | ((uint32_t)(ib & 0xff) << 8));
    int s = ia * ia + ib * ib;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        norm[row] = (row < n) ? s : 0;
        tneg[row] = (row < n) ? -s : KNN_PAD_TNEG;
    }
    if (__builtin_amdgcn_ballot_w64(bad) && lane == 0) atomicOr(flag, 1);
}

int imatch_features_alloc(imatch_ctx* ctx, int64_t n, imatch_features** out) {
    imatch_features* f = (imatch_features*)calloc(1, sizeof(imatch_features));
    if (!f) return imatch_set_err(ctx, IMATCH_ENOMEM, "host alloc");
    f->ctx = ctx;
    f->n = n;
    f->n_pad = ((n + IMATCH_ROW_PAD - 1) / IMATCH_ROW_PAD) * IMATCH_ROW_PAD;
    if (f->n_pad == 0) f->n_pad = IMATCH_ROW_PAD;
    // one block: int8 rows, then norm + tneg (int32 each), the integrality flag, and room for the keypoints' (x, y) of
    // sets made by an extraction (has_xy; imatch_match_localize gathers matched coordinates on the device)
    f->desc = (int8_t*)imatch_pool_alloc(ctx, (size_t)f->n_pad * (128 + 8 + 8) + 256);
    if (!f->desc) {
        free(f);
        return imatch_set_err(ctx, IMATCH_ENOMEM, "device alloc for %lld descriptors", (long long)n);
    }
    f->norm = (int*)((char*)f->desc + (size_t)f->n_pad * 128);
    f->flag = (int*)((char*)f->desc + (size_t)f->n_pad * (128 + 8));
    f->tneg = f->norm + f->n_pad;
    f->xy = (float*)((char*)f->desc + (size_t)f->n_pad * (128 + 8) + 256);
    *out = f;
    return IMATCH_OK;
}

int imatch_features_pack_u8_seg_dev(imatch_ctx* ctx, imatch_features* f, const uint8_t* src_dev, const imatch_segtab* seg,
                                    size_t zs) {
    HIP_TRY(ctx, hipMemsetAsync(f->flag, 0, 256, ctx->stream));
    const unsigned grid = (unsigned)((f->n_pad + 3) / 4);
    hipLaunchKernelGGL((pack_rows_kernel<uint8_t, true>), dim3(grid), dim3(256), 0, ctx->stream, src_dev, f->n, f->n_pad,
                       f->desc, f->norm, f->tneg, f->flag, *seg, zs);
    HIP_TRY(ctx, hipGetLastError());
    f->integral = 1;
    return IMATCH_OK;
}

template <typename T>
static int features_pack(imatch_ctx* ctx, imatch_features* f, const T* src_dev) {
    HIP_TRY(ctx, hipMemsetAsync(f->flag, 0, 256, ctx->stream));
    const int rows_per_block = 4;
    const unsigned grid = (unsigned)((f->n_pad + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(pack_rows_kernel<T>, dim3(grid), dim3(256), 0, ctx->stream, src_dev, f->n, f->n_pad, f->desc,
                       f->norm, f->tneg, f->flag);
    HIP_TRY(ctx, hipGetLastError());
    int flag = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&flag, f->flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    f->integral = !flag;
    return IMATCH_OK;
}

// pad rows of a set whose real rows (int8 values, norm, T') were written by their producer (describe_kernel)
__global__ void pad_rows_kernel(int64_t n, int64_t n_pad, int8_t* __restrict__ dst, int* __restrict__ norm,
                                int* __restrict__ tneg, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = n + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;
    if (row >= n_pad) return;
    ((uint16_t*)dst)[row * 64 + lane] = 0;
    if (lane == 0) {
        norm[row] = 0;
        tneg[row] = KNN_PAD_TNEG;
    }
}

int imatch_features_pad_dev(imatch_ctx* ctx, imatch_features* f) {
    const unsigned grid = (unsigned)((f->n_pad - f->n + 3) / 4 + 1);
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, ctx->stream, f->n, f->n_pad, f->desc, f->norm, f->tneg, f->flag);
    HIP_TRY(ctx, hipGetLastError());
    f->integral = 1;
    return IMATCH_OK;
}

// The same for a set whose row count is still on the device when the launch is queued (a single frame's descriptors,
// launched behind the sort before the host has seen its count): every row from *n_dev (a negative count reads as 0) up
// to the set's allocated capacity becomes a pad row; the host sets f->n and f->n_pad once it knows the count.
__global__ void pad_rows_counted_kernel(const int* __restrict__ n_dev, int64_t n_alloc, int8_t* __restrict__ dst,
                                        int* __restrict__ norm, int* __restrict__ tneg, int* __restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) *flag = 0;
    if (row >= n_alloc || row < (int64_t)max(*n_dev, 0)) return;
    ((uint16_t*)dst)[row * 64 + lane] = 0;
    if (lane == 0) {
        norm[row] = 0;
        tneg[row] = KNN_PAD_TNEG;
    }
}

int imatch_features_pad_counted_dev(imatch_ctx* ctx, imatch_features* f, const int* n_dev) {
    const unsigned grid = (unsigned)((f->n_pad + 3) / 4);
    hipLaunchKernelGGL(pad_rows_counted_kernel, dim3(grid), dim3(256), 0, ctx->stream, n_dev, f->n_pad, f->desc, f->norm,
                       f->tneg, f->flag);
    HIP_TRY(ctx, hipGetLastError());
    f->integral = 1;
    return IMATCH_OK;
}

// u8 rows are integral by type: no flag to read back, so this only queues work on the stream
int imatch_features_pack_u8_dev(imatch_ctx* ctx, imatch_features* f, const uint8_t* src_dev) {
    HIP_TRY(ctx, hipMemsetAsync(f->flag, 0, 256, ctx->stream));
    const unsigned grid = (unsigned)((f->n_pad + 3) / 4);
    hipLaunchKernelGGL(pack_rows_kernel<uint8_t>, dim3(grid), dim3(256), 0, ctx->stream, src_dev, f->n, f->n_pad, f->desc,
                       f->norm, f->tneg, f->flag);
    HIP_TRY(ctx, hipGetLastError());
    f->integral = 1;
    return IMATCH_OK;
}

template <typename T>
static int features_from_host(imatch_ctx* ctx, const T* desc, int64_t n, int d, imatch_features** out) {
    if (!ctx || !out || n < 0 || (n > 0 && !desc)) return imatch_set_err(ctx, IMATCH_EINVAL, "null or negative argument");
    if (d != IMATCH_DESC_DIM) return imatch_set_err(ctx, IMATCH_EINVAL, "descriptor dimension %d, need 128", d);
    if (n > (int64_t)INT_MAX - 2 * IMATCH_ROW_PAD) return imatch_set_err(ctx, IMATCH_ESIZE, "too many descriptors");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    imatch_features* f = nullptr;
    int rc = imatch_features_alloc(ctx, n, &f);
    if (rc) return rc;
    const size_t bytes = (size_t)n * 128 * sizeof(T);
    rc = imatch_ws_reserve(ctx, bytes ? bytes : 16);
    if (rc) { imatch_features_free(f); return rc; }
    if (bytes) {
        hipError_t e = hipMemcpyAsync(ctx->ws, desc, bytes, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { imatch_features_free(f); return imatch_set_err(ctx, IMATCH_EHIP, "upload: %s", hipGetErrorString(e)); }
    }
    rc = features_pack<T>(ctx, f, (const T*)ctx->ws);
    if (rc) { imatch_features_free(f); return rc; }
    *out = f;
    return IMATCH_OK;
}

// int8 rows -> u8 rows (integral sets only): the exchange format of the multi-GPU path; four values per thread
__global__ void unpack_rows_u8_kernel(const uint32_t* __restrict__ src, int64_t n, uint32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 32) return;
    dst[i] = src[i] ^ 0x80808080u;   // x' + 128 on every byte
}

// ---- launch -------------------------------------------------------------------------------
// ---- scratch of the entry point that allocates nothing (imatch_knn2_noalloc_dev) -----------------------------------
// The launcher's scratch -- one partial top-2 list per (train chunk, query) and three bound words per query -- normally
// lives in the context's growable workspace (hipMalloc on growth). Callers that cannot allocate get it from this
// file-scope device array instead: 64 MiB per GPU, held until the process ends. In host code the array's NAME is only a
// stand-in, not a device address; the per-device address is looked up once per device, with that device current
// (imatch_init calls imatch_knn2_scratch_setup), and kept here.
#define KNN_SCRATCH_BYTES ((size_t)64 << 20)
#define KNN_SCRATCH_MAX_DEV 64
static __device__ __align__(256) unsigned char knn_scratch[KNN_SCRATCH_BYTES];
static void* knn_scratch_ptr[KNN_SCRATCH_MAX_DEV];   // device addresses of knn_scratch, one per GPU
static std::mutex knn_scratch_mu;

int imatch_knn2_scratch_setup(imatch_ctx* ctx) {
    if (ctx->device < 0 || ctx->device >= KNN_SCRATCH_MAX_DEV) return IMATCH_OK;   // the entry point reports it
    std::lock_guard<std::mutex> lk(knn_scratch_mu);
    if (knn_scratch_ptr[ctx->device]) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    HIP_TRY(ctx, hipGetSymbolAddress(&p, HIP_SYMBOL(knn_scratch)));
    if (!p) return imatch_set_err(ctx, IMATCH_EHIP, "hipGetSymbolAddress gave no address for the kNN scratch array");
    knn_scratch_ptr[ctx->device] = p;
    return IMATCH_OK;
}

// ws == nullptr: the scratch comes from the context's workspace (grown on demand); else from the ws_cap bytes at ws.
static int knn2_run(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, int32_t idx_offset,
                    int32_t* out4_dev, void* ws = nullptr, size_t ws_cap = 0) {
    if (!ctx || !q || !t || !out4_dev) return imatch_set_err(ctx, IMATCH_EINVAL, "null argument");
    // the int8 rows hold integers 0..255 (SIFT's descriptors) and nothing else; any other set would get wrong neighbours
    // with rc 0
    if (!q->integral || !t->integral)
        return imatch_set_err(ctx, IMATCH_EINVAL, "%s descriptor set is not integer valued in 0..255: the int8 MFMA kernel "
                              "takes only such sets; use imatch_knn2_f32", !q->integral ? "query" : "train");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (q->n == 0) return IMATCH_OK;
    static const int tn_env = getenv("IMATCH_KNN_TN") ? atoi(getenv("IMATCH_KNN_TN")) : 128;   // 256: A/B runs of the large form
    static const int young = getenv("IMATCH_KNN_YOUNG") ? atoi(getenv("IMATCH_KNN_YOUNG")) : 8;
    static const unsigned young_wg = getenv("IMATCH_KNN_YOUNG_WG") ? (unsigned)atoi(getenv("IMATCH_KNN_YOUNG_WG")) : 0xffffffffu;
    // Three forms of one kernel. A single frame's few thousand queries against a map of up to a million rows: 256-query
    // workgroups (QB = 4), twice the workgroups per chunk, so the chunks are twice as long and the resident-query prologue
    // and the partial lists weigh half as much; from 6144 queries or 1 M rows on the 512-query form (QB = 8) ...
    static const int64_t q_small = getenv("IMATCH_KNN_Q16_SMALL") ? atoll(getenv("IMATCH_KNN_Q16_SMALL")) : 6144;
    static const int form_env = getenv("IMATCH_KNN_FORM") ? atoi(getenv("IMATCH_KNN_FORM")) : 0;   // 4 / 8: force a form (A/B runs)
    static const double short_pairs = getenv("IMATCH_KNN_SHORT_PAIRS") ? atof(getenv("IMATCH_KNN_SHORT_PAIRS")) : 8e8;
    bool small = q->n <= q_small && t->n_pad <= (1 << 20);
    bool short8 = false;
    // ... and from about 8e8 (query, row) pairs on -- a frame against a map of a quarter of a million rows or more -- the
    // 512-query form with the 256-query form's first exchange: its stages are twice as long, so the per-tile and per-stage
    // overheads weigh half as much (profiles/r2_knn_shapes.txt)
    if (small && q->n > 256 && (form_env == 8 || (form_env == 0 && (double)q->n * (double)t->n_pad >= short_pairs))) {
        small = false;
        short8 = true;
    }
    if (form_env == 8 && q->n <= q_small) { small = false; short8 = true; }
    if (form_env == 4 && q->n <= q_small) { small = true; short8 = false; }
    const int QB = small ? 4 : 8;
    const int TNs = (!small && !short8 && tn_env == 256 && t->n_pad % 256 == 0) ? 256 : 128;
    const int qtile = 4 * 16 * QB;
    const int64_t Qpad = q->n_pad;  // multiple of 1024, so of every qtile
    const int nQt = (int)((q->n + qtile - 1) / qtile);
    const int nStagesAll = (int)(t->n_pad / TNs);
    const int slots = 512;
    // chunks of a range of stages: a multiple of 8 (one group per XCD); two rounds of 512 workgroup slots (large form) or one of 768 (a frame's forms); >= 4 stages each
    auto pick_chunks = [&](int nStages) {
        // (rounded DOWN to a multiple of 8, so that the grid stays within two rounds of slots)
        // -- ONE round for the frame forms: their launches are short (a frame against a map of well under a million
        // rows), a second, partly filled round would be a large part of them (1500 x 59k: 54 -> 44 us, 2951 x 59k: 76 -> 67)
        const bool one_round = small || short8;
        // (three workgroups fit a CU since the int8 form: the single round is 768 slots; the two-round rule of the large
        // form keeps its 2 x 512 -- more, shorter chunks were measured and lose there)
        int c = ((one_round ? slots + slots / 2 : 2 * slots + nQt - 1) / nQt / 8) * 8;
        // ... and chunks of at most 16k rows (2 MB, half an XCD's L2): the query tiles that share a chunk then
        // re-read it from L2 however far apart they run, and the many short workgroups balance the tail
        c = std::max(c, (int)(((int64_t)nStages * TNs + 16383) / 16384 + 7) / 8 * 8);
        c = std::min(c, 256);
        if (c > nStages / 4) c = (nStages / 4 / 8) * 8;   // (six stages at least were tried: 700 x 20 k 0.023 -> 0.033 ms)
        if (c < 8) c = 8;
        if (const char* e = getenv("IMATCH_KNN_CHUNKS")) c = std::max(8, atoi(e) / 8 * 8);  // tuning override
        return c;
    };
    static const int rot_on = getenv("IMATCH_KNN_ROT") ? atoi(getenv("IMATCH_KNN_ROT")) : 1;
    const int nStages = nStagesAll;
    int nChunks = pick_chunks(nStages);
    const size_t gbytes = ((size_t)Qpad * 12 + 255) & ~(size_t)255;   // three bounds per query
    if (ws) {   // a fixed scratch block: fewer, longer chunks until the partial lists fit
        const size_t per_chunk = (size_t)Qpad * 16;
        const size_t room = ws_cap > gbytes ? (ws_cap - gbytes) / per_chunk : 0;
        if (room < 8)
            return imatch_set_err(ctx, IMATCH_ESIZE, "%lld queries need %zu bytes of kNN scratch, the fixed block holds %zu "
                                  "(at most 262144 queries per call)", (long long)q->n, 8 * per_chunk + gbytes, ws_cap);
        if ((size_t)nChunks > room) nChunks = (int)(room / 8) * 8;
    }
    const int64_t nLists = nChunks;   // the kernel merges its four row quarters itself: one list per chunk
    ctx->knn_last_chunks = nChunks;
    const size_t pbytes = (size_t)nLists * Qpad * 8;
    const bool own_ws = !ws;
    if (own_ws) {
        int rc = imatch_ws_reserve(ctx, 2 * pbytes + gbytes);
        if (rc) return rc;
        ws = ctx->ws;
    }
    int2* pb = (int2*)ws;
    int2* pi = (int2*)((char*)ws + pbytes);
    int* gthr = (int*)((char*)ws + 2 * pbytes);
    // The bound words start at KNN_INF. With the context's workspace they live in a block of the context's own that is
    // cleared when it is made or grown and put back in order by the merge kernel of every launch (4 us of clear plus a
    // launch gap in front of a frame's 0.1 ms kNN); a launch that failed half-way leaves the block marked and the next
    // one clears it. The caller-supplied scratch block is cleared per call.
    int* rearm = nullptr;
    if (own_ws) {
        const size_t words = gbytes / 4;
        if (ctx->knn_gthr_words < words) {
            imatch_pool_free(ctx, ctx->knn_gthr);
            ctx->knn_gthr_words = 0;
            const size_t cap_words = std::max(words, (size_t)3 * 16384);
            ctx->knn_gthr = (int*)imatch_pool_alloc(ctx, cap_words * 4);
            if (!ctx->knn_gthr) return imatch_set_err(ctx, IMATCH_ENOMEM, "kNN bound words");
            ctx->knn_gthr_words = cap_words;
            ctx->knn_gthr_armed = 0;
        }
        gthr = ctx->knn_gthr;
        if (!ctx->knn_gthr_armed) HIP_TRY(ctx, hipMemsetAsync(gthr, 0x7f, ctx->knn_gthr_words * 4, ctx->stream));
        ctx->knn_gthr_armed = 0;
        rearm = gthr;
    } else {
        HIP_TRY(ctx, hipMemsetAsync(gthr, 0x7f, gbytes, ctx->stream));  // KNN_INF in every word
    }
    { const int rcs = imatch_stage_begin(ctx, IMATCH_STAGE_KNN, ctx->stream); if (rcs) return rcs; }
    {
        const unsigned grid = (unsigned)(nQt * nChunks);
        const int rot = rot_on && grid > 2u * slots;
#define KARGS (const uint4*)q->desc, (const uint4*)t->desc, t->tneg, nQt, nChunks, nStages, Qpad, pb, pi, gthr, young, young_wg, 0, 0, rot
        // (development: extra dynamic LDS per workgroup, i.e. fewer workgroups per CU: 1 / 2 / 3 per CU give 28 / 51 / 59 %
        // of the int8 peak at the bench shape)
        static const unsigned dyn = getenv("IMATCH_KNN_DYNLDS") ? (unsigned)atoi(getenv("IMATCH_KNN_DYNLDS")) : 0;
        if (small)
            hipLaunchKernelGGL((knn2_kernel8<4, 128, 7>), dim3(grid), dim3(256), dyn, ctx->stream, KARGS);
        else if (short8)
            hipLaunchKernelGGL((knn2_kernel8<8, 128, 7, true>), dim3(grid), dim3(256), dyn, ctx->stream, KARGS);
        else if (TNs == 256)
            hipLaunchKernelGGL((knn2_kernel8<8, 256, 7>), dim3(grid), dim3(256), dyn, ctx->stream, KARGS);
        else
            hipLaunchKernelGGL((knn2_kernel8<8, 128, KNN_XM>), dim3(grid), dim3(256), dyn, ctx->stream, KARGS);
#undef KARGS
    }
    HIP_TRY(ctx, hipGetLastError());
    { const int rcs = imatch_stage_end(ctx, IMATCH_STAGE_KNN, ctx->stream); if (rcs) return rcs; }
    { const int rcs = imatch_stage_begin(ctx, IMATCH_STAGE_KNN_MERGE, ctx->stream); if (rcs) return rcs; }
    hipLaunchKernelGGL(knn2_finish_kernel, dim3((unsigned)((q->n + 15) / 16)), dim3(256), 0, ctx->stream, pb, pi,
                       (int)nLists, Qpad, q->norm, q->n, t->n, idx_offset, (int4*)out4_dev, rearm, (int64_t)(gbytes / 4));
    HIP_TRY(ctx, hipGetLastError());
    if (own_ws) ctx->knn_gthr_armed = 1;
    { const int rcs = imatch_stage_end(ctx, IMATCH_STAGE_KNN_MERGE, ctx->stream); if (rcs) return rcs; }
    return IMATCH_OK;
}

extern "C" {

int imatch_features_from_host_f32(imatch_ctx* ctx, const float* desc, int64_t n, int d, imatch_features** out) {
    return features_from_host<float>(ctx, desc, n, d, out);
}

int imatch_features_from_host_u8(imatch_ctx* ctx, const uint8_t* desc, int64_t n, int d, imatch_features** out) {
    return features_from_host<uint8_t>(ctx, desc, n, d, out);
}

int imatch_features_from_device_u8(imatch_ctx* ctx, const uint8_t* desc_dev, int64_t n, int d, imatch_features** out) {
    if (!ctx || !out || n < 0 || (n > 0 && !desc_dev)) return imatch_set_err(ctx, IMATCH_EINVAL, "null or negative argument");
    if (d != IMATCH_DESC_DIM) return imatch_set_err(ctx, IMATCH_EINVAL, "descriptor dimension %d, need 128", d);
    if (n > (int64_t)INT_MAX - 2 * IMATCH_ROW_PAD) return imatch_set_err(ctx, IMATCH_ESIZE, "too many descriptors");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    imatch_features* f = nullptr;
    int rc = imatch_features_alloc(ctx, n, &f);
    if (rc) return rc;
    rc = features_pack<uint8_t>(ctx, f, desc_dev);
    if (rc) { imatch_features_free(f); return rc; }
    *out = f;
    return IMATCH_OK;
}

int imatch_features_from_device_u8_seg(imatch_ctx* ctx, const uint8_t* desc_dev, int n_seg, const int64_t* counts,
                                       size_t seg_pitch, int d, imatch_features** out) {
    if (!ctx || !out || !counts || n_seg < 1 || n_seg > IMATCH_MAX_BATCH)
        return imatch_set_err(ctx, IMATCH_EINVAL, "null argument or more than %d segments", IMATCH_MAX_BATCH);
    if (d != IMATCH_DESC_DIM) return imatch_set_err(ctx, IMATCH_EINVAL, "descriptor dimension %d, need 128", d);
    imatch_segtab seg;
    memset(&seg, 0, sizeof(seg));
    seg.n = n_seg;
    int64_t total = 0;
    for (int b = 0; b < n_seg; ++b) {
        if (counts[b] < 0 || (size_t)counts[b] * 128 > seg_pitch)
            return imatch_set_err(ctx, IMATCH_EINVAL, "segment %d: %lld rows do not fit its pitch of %zu bytes", b,
                                  (long long)counts[b], seg_pitch);
        total += counts[b];
        if (total > (int64_t)INT_MAX - 2 * IMATCH_ROW_PAD) return imatch_set_err(ctx, IMATCH_ESIZE, "too many descriptors");
        seg.off[b + 1] = (int)total;
    }
    if (total > 0 && !desc_dev) return imatch_set_err(ctx, IMATCH_EINVAL, "null descriptor pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out = nullptr;
    imatch_features* f = nullptr;
    int rc = imatch_features_alloc(ctx, total, &f);
    if (rc) return rc;
    rc = imatch_features_pack_u8_seg_dev(ctx, f, desc_dev, &seg, seg_pitch);
    if (rc) { imatch_features_free(f); return rc; }
    *out = f;
    return IMATCH_OK;
}

int imatch_features_desc_u8_dev(const imatch_features* f, uint8_t* out_dev) {
    if (!f) return IMATCH_EINVAL;
    imatch_ctx* ctx = f->ctx;
    if (!out_dev && f->n > 0) return imatch_set_err(ctx, IMATCH_EINVAL, "null output");
    if (!f->integral) return imatch_set_err(ctx, IMATCH_EINVAL, "descriptor set is not integer valued");
    if (f->n == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if ((uintptr_t)out_dev & 3) return imatch_set_err(ctx, IMATCH_EINVAL, "output rows must be 4-byte aligned");
    hipLaunchKernelGGL(unpack_rows_u8_kernel, dim3((unsigned)((f->n * 32 + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint32_t*)f->desc, f->n, (uint32_t*)out_dev);
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

int64_t imatch_features_count(const imatch_features* f) { return f ? f->n : 0; }

int imatch_features_is_integral(const imatch_features* f) { return f ? f->integral : 0; }

void imatch_features_free(imatch_features* f) {
    if (!f) return;
    if (f->ctx) {
        imatch_pool_free(f->ctx, f->desc);
        imatch_pool_free(f->ctx, f->image);
    }
    free(f->kp);
    free(f);
}

int imatch_knn2_dev(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, int32_t idx_offset,
                    int32_t* out4_dev) {
    return knn2_run(ctx, q, t, idx_offset, out4_dev);
}

int imatch_knn2_debug_last_chunks(imatch_ctx* ctx) { return ctx ? ctx->knn_last_chunks : 0; }

int imatch_knn2_noalloc_dev(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, int32_t idx_offset,
                            int32_t* out4_dev) {
    if (!ctx || !q || !t || !out4_dev) return imatch_set_err(ctx, IMATCH_EINVAL, "null argument");
    if (ctx->device < 0 || ctx->device >= KNN_SCRATCH_MAX_DEV || !knn_scratch_ptr[ctx->device])
        return imatch_set_err(ctx, IMATCH_EINVAL, "no kNN scratch block on device %d (set up by imatch_init; devices 0..%d)",
                              ctx->device, KNN_SCRATCH_MAX_DEV - 1);
    if (q->n > 262144)
        return imatch_set_err(ctx, IMATCH_ESIZE, "%lld queries: the entry point that allocates nothing takes at most 262144 per call",
                              (long long)q->n);
    // every kernel argument and the memset below take the looked-up device address, never the array's name
    return knn2_run(ctx, q, t, idx_offset, out4_dev, knn_scratch_ptr[ctx->device], KNN_SCRATCH_BYTES);
}

int imatch_knn2(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, int32_t* idx, float* dist) {
    if (!ctx || !q || !t || (q->n > 0 && (!idx || !dist))) return imatch_set_err(ctx, IMATCH_EINVAL, "null argument");
    if (q->n == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int4* out_dev = (int4*)imatch_pool_alloc(ctx, (size_t)q->n * 16);
    if (!out_dev) return imatch_set_err(ctx, IMATCH_ENOMEM, "result list for %lld queries", (long long)q->n);
    int rc = knn2_run(ctx, q, t, 0, (int32_t*)out_dev);
    if (rc == IMATCH_OK) rc = imatch_hs_reserve(ctx, (size_t)q->n * 16);
    if (rc == IMATCH_OK) {
        hipError_t e = hipMemcpyAsync(ctx->hs, out_dev, (size_t)q->n * 16, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = imatch_set_err(ctx, IMATCH_EHIP, "knn2 download: %s", hipGetErrorString(e));
    }
    imatch_pool_free(ctx, out_dev);
    if (rc) return rc;
    const int32_t* o = (const int32_t*)ctx->hs;
    for (int64_t i = 0; i < q->n; ++i) {
        idx[2 * i] = o[4 * i + 1];
        idx[2 * i + 1] = o[4 * i + 3];
        dist[2 * i] = o[4 * i + 1] >= 0 ? sqrtf((float)o[4 * i]) : INFINITY;
        dist[2 * i + 1] = o[4 * i + 3] >= 0 ? sqrtf((float)o[4 * i + 2]) : INFINITY;
    }
    return IMATCH_OK;
}

int imatch_merge_top2_dev(imatch_ctx* ctx, const int32_t* in4_dev, int n_lists, int64_t nq, int32_t* out4_dev) {
    if (!ctx || !in4_dev || !out4_dev || n_lists < 1 || nq < 0) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    if (nq == 0) return IMATCH_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(merge_top2_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const int4*)in4_dev, n_lists, nq, (int4*)out4_dev);
    HIP_TRY(ctx, hipGetLastError());
    return IMATCH_OK;
}

static int ratio_run(imatch_ctx* ctx, const int32_t* out4_dev, const imatch_segtab& seg, double ratio, int32_t* good_q,
                     int32_t* good_t, float* good_d, int64_t* good_off) {
    const int64_t nq = seg.off[seg.n];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // gq, gt, gd (nq each) + one counter per segment
    const size_t arr = ((size_t)nq * 4 + 255) & ~(size_t)255;
    const size_t cnt = ((size_t)seg.n * 8 + 255) & ~(size_t)255;
    char* dev = (char*)imatch_pool_alloc(ctx, 3 * arr + cnt);
    if (!dev) return imatch_set_err(ctx, IMATCH_ENOMEM, "match lists for %lld queries", (long long)nq);
    int rc = imatch_hs_reserve(ctx, 3 * arr + cnt);
    if (rc) { imatch_pool_free(ctx, dev); return rc; }
    (void)imatch_stage_begin(ctx, IMATCH_STAGE_RATIO, ctx->stream);
    hipLaunchKernelGGL(ratio_kernel, dim3(seg.n), dim3(1024), 0, ctx->stream, (const int4*)out4_dev, seg, ratio, (int*)dev,
                       (int*)(dev + arr), (float*)(dev + 2 * arr), (unsigned long long*)(dev + 3 * arr));
    hipError_t e = hipGetLastError();
    (void)imatch_stage_end(ctx, IMATCH_STAGE_RATIO, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->hs, dev, 3 * arr + cnt, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    imatch_pool_free(ctx, dev);
    if (e != hipSuccess) return imatch_set_err(ctx, IMATCH_EHIP, "ratio: %s", hipGetErrorString(e));
    const char* hsrc = (const char*)ctx->hs;
    const unsigned long long* ng = (const unsigned long long*)(hsrc + 3 * arr);
    int64_t at = 0;
    for (int b = 0; b < seg.n; ++b) {
        const int64_t g = (int64_t)ng[b], lo = seg.off[b];
        good_off[b] = at;
        memcpy(good_q + at, hsrc + lo * 4, (size_t)g * 4);
        memcpy(good_t + at, hsrc + arr + lo * 4, (size_t)g * 4);
        memcpy(good_d + at, hsrc + 2 * arr + lo * 4, (size_t)g * 4);
        at += g;
    }
    good_off[seg.n] = at;
    return IMATCH_OK;
}

int imatch_ratio_dev(imatch_ctx* ctx, const int32_t* out4_dev, int64_t nq, double ratio, int32_t* good_q,
                     int32_t* good_t, float* good_d, int64_t* n_good) {
    if (!ctx || !n_good || nq < 0 || nq > INT_MAX || (nq > 0 && (!out4_dev || !good_q || !good_t || !good_d)))
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    *n_good = 0;
    if (nq == 0) return IMATCH_OK;
    imatch_segtab seg;
    memset(&seg, 0, sizeof(seg));
    seg.n = 1;
    seg.off[1] = (int)nq;
    int64_t off[2] = {0, 0};
    int rc = ratio_run(ctx, out4_dev, seg, ratio, good_q, good_t, good_d, off);
    if (rc) return rc;
    *n_good = off[1];
    return IMATCH_OK;
}

int imatch_ratio_batch_dev(imatch_ctx* ctx, const int32_t* out4_dev, const int64_t* offsets, int n_seg, double ratio,
                           int32_t* good_q, int32_t* good_t, float* good_d, int64_t* good_offsets) {
    if (!ctx || !offsets || !good_offsets || n_seg < 1 || n_seg > IMATCH_MAX_BATCH)
        return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    imatch_segtab seg;
    memset(&seg, 0, sizeof(seg));
    seg.n = n_seg;
    for (int b = 0; b <= n_seg; ++b) {
        if (offsets[b] < 0 || offsets[b] > INT_MAX || (b > 0 && offsets[b] < offsets[b - 1]))
            return imatch_set_err(ctx, IMATCH_EINVAL, "segment offsets must ascend");
        seg.off[b] = (int)offsets[b];
    }
    for (int b = 0; b <= n_seg; ++b) good_offsets[b] = 0;
    if (seg.off[n_seg] == 0) return IMATCH_OK;
    if (offsets[0] != 0 || !out4_dev || !good_q || !good_t || !good_d) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    return ratio_run(ctx, out4_dev, seg, ratio, good_q, good_t, good_d, good_offsets);
}

}  // extern "C"
