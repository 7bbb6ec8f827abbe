// Internal declarations shared by the HIP translation units of libimatch.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/imatch.h"

struct imatch_ctx {
    int device;
    hipStream_t stream;
    char err[512];
    // growable device workspace, reused across calls so the hot path does not hipMalloc
    void* ws;
    size_t ws_bytes;
    // pinned host staging
    void* hs;
    size_t hs_bytes;
    // an asynchronous host-to-device copy may still be reading hs after the call that queued it has returned (the sorted
    // keypoint records of imatch_sift_extract): it records hs_ev, and imatch_hs_reserve -- which every writer of hs calls
    // first -- waits for it
    hipEvent_t hs_ev;
    int hs_pending;
    // pinned staging of host images on their way to / from the device (see imatch_stage_upload)
    void* is;
    size_t is_bytes;
    // optional event brackets around the stages of the path (imatch_set_profiling, imatch_stage_times): [stage][begin, end]
    int prof;
    hipEvent_t st_ev[IMATCH_STAGE_COUNT][2];
    unsigned st_seen;      // bit s: stage s has been bracketed at least once since profiling was switched on
    int knn_last_chunks;   // train chunks of the last kNN launch (imatch_knn2_debug_last_chunks)
    // bound words of the int8 kNN (three per query), a pool block of the context: all KNN_INF between launches when
    // knn_gthr_armed (the merge kernel of a launch puts back what the launch used)
    int* knn_gthr;
    size_t knn_gthr_words;
    int knn_gthr_armed;
    int tail_mismatches;   // imatch_match_localize: calls whose device-side Hartley transforms were not the host's (expected: 0)
    int knnf_last_path;    // the float kNN's last call: 0 exhaustive kernel, 1 filter + refinement, 2 filter tried, then exhaustive (imatch_knn2_f32_debug_last_path)
    // SIFT pyramid arena of the last extraction (kept for reuse by the next frame of the same size)
    struct imatch_sift_state* sift;
    struct imatch_workers* workers;   // host worker pool, created on first use
    struct imatch_akaze_state* akaze;     // AKAZE arena and last result (akaze.hip)
    struct imatch_camera_state* camera;   // rectification maps of the last calibration (camera.hip)
    // device blocks handed out by imatch_pool_alloc; freed blocks are kept for the next call
    struct imatch_pool_block* pool;
    int pool_n, pool_cap;
};

struct imatch_pool_block {
    void* ptr;
    size_t bytes;
    int in_use;
};

struct imatch_features {
    imatch_ctx* ctx;
    int64_t n;        // valid rows
    int64_t n_pad;    // rows allocated, multiple of IMATCH_ROW_PAD; pad rows are zero bytes with tneg = -2^24
    int8_t* desc;     // int8 [n_pad][128] = value - 128 (values are integers 0..255), device
    int* norm;        // int32 [n_pad] squared L2 norms of the SHIFTED rows (the kNN works on differences only), device
    int* tneg;        // int32 [n_pad] = -norm: the MFMA accumulator starts at tneg >> 1; norm + n_pad
    int* flag;        // device int: 1 if some value was not an integer in 0..255
    float* xy;        // device float [n_pad][2]: the keypoints' coordinates, written by the descriptor kernel (has_xy), else unset
    int has_xy;
    int integral;     // host copy of !flag, valid after creation
    // sets produced by imatch_sift_extract also carry:
    float* kp;        // host [n][7] keypoint records (final order), or null
    uint8_t* image;   // device [img_h][img_w] CLAHE output, or null
    int img_h, img_w;
};

#define IMATCH_ROW_PAD 1024

// Batched extraction: rows [off[b], off[b+1]) of a concatenated set belong to image b.
#define IMATCH_MAX_BATCH 256
struct imatch_segtab { int n; int off[IMATCH_MAX_BATCH + 1]; };

int imatch_features_alloc(imatch_ctx* ctx, int64_t n, imatch_features** out);
int imatch_features_pack_u8_dev(imatch_ctx* ctx, imatch_features* f, const uint8_t* src_dev);
int imatch_features_pad_dev(imatch_ctx* ctx, imatch_features* f);   // rows n .. n_pad and the flag of a set whose rows a kernel wrote itself
// ... when the row count is still on the device: rows *n_dev .. n_pad (the allocated capacity)
int imatch_features_pad_counted_dev(imatch_ctx* ctx, imatch_features* f, const int* n_dev);
// rows of image b are read from src_dev + b*zs (bytes) + (row - off[b])*128
int imatch_features_pack_u8_seg_dev(imatch_ctx* ctx, imatch_features* f, const uint8_t* src_dev, const imatch_segtab* seg,
                                    size_t zs);
// looks up the device address of the kNN scratch array for ctx->device (knn.hip; called by imatch_init)
int imatch_knn2_scratch_setup(imatch_ctx* ctx);
void imatch_sift_state_free(imatch_ctx* ctx);
void imatch_camera_state_free(imatch_ctx* ctx);
void imatch_akaze_state_free(imatch_ctx* ctx);
// the descriptors [n][64] of the AKAZE extraction `serial`, where the extraction left them on the device; IMATCH_EINVAL
// (no error text) when the context's last extraction is another one
extern "C" int imatch_akaze_last_dev(imatch_ctx* ctx, uint64_t serial, const float** desc_dev, int64_t* n);
int imatch_set_err(imatch_ctx* ctx, int code, const char* fmt, ...);
// Stage brackets (core.hip): with profiling on, an event on `st` and a roctx range (when libroctx64 is loadable) at both
// ends of a stage; nothing otherwise. Return a hipError_t-free int: IMATCH_OK or IMATCH_EHIP with the message set.
int imatch_stage_begin(imatch_ctx* ctx, int stage, hipStream_t st);
int imatch_stage_end(imatch_ctx* ctx, int stage, hipStream_t st);
int imatch_ws_reserve(imatch_ctx* ctx, size_t bytes);
// Per-context device block pool: a frame's descriptor set, image copy and result lists come from here,
// so the per-frame path calls neither hipMalloc nor hipFree (hipFree drains the device). Blocks are
// only ever used on ctx->stream, so a freed block may be handed out again without a synchronisation.
void* imatch_pool_alloc(imatch_ctx* ctx, size_t bytes);
void imatch_pool_free(imatch_ctx* ctx, void* p);
void imatch_pool_destroy(imatch_ctx* ctx);
int imatch_hs_reserve(imatch_ctx* ctx, size_t bytes);
// call after queueing a copy that reads ctx->hs and is not followed by a synchronisation before the call returns
int imatch_hs_mark_busy(imatch_ctx* ctx);
// Host worker pool (core.hip): runs fn(i) for i in [0, n) on up to 16 persistent threads of the context and returns
// when all are done. The per-image sorts of a batch and the per-frame homography refits use it: spawning 16
// std::threads per call cost about as much as the work itself.
#ifdef __cplusplus
#include <functional>
void imatch_parallel_for(imatch_ctx* ctx, int n, const std::function<void(int)>& fn);
void imatch_workers_destroy(imatch_ctx* ctx);
#endif
// Host images are pageable numpy memory: an asynchronous 2-D copy from it degenerates into one small
// transfer per row (6 ms for a 1280x720 frame). The rows are packed into a pinned block first and go up
// in one transfer; a download lands in the block's second part and is copied out after the caller's
// synchronisation. Every user synchronises ctx->stream before it returns, so the block is free again by
// the next call.
int imatch_stage_upload(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, uint8_t* dev);
int imatch_stage_download_begin(imatch_ctx* ctx, const uint8_t* dev, size_t bytes);   // queued on ctx->stream
void imatch_stage_download_end(imatch_ctx* ctx, uint8_t* out, size_t bytes);          // after the synchronisation

#define HIP_TRY(ctx, call)                                                              \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return imatch_set_err((ctx), IMATCH_EHIP, "%s: %s (%s:%d)", #call,          \
                                  hipGetErrorString(e_), __FILE__, __LINE__);           \
    } while (0)
