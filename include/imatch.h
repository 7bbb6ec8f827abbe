/*
 * imatch.h -- C-ABI of libimatch.so, the MI355X (gfx950) implementation of the
 * SIFT extract -> kNN(k=2)+ratio -> homography hot path of Fricodelco/image_matching.
 *
 * The reference has no FFI: its seam is the Python class `match_finder`
 * (image_processing/src/image_processing/match_finder.py:26-352, "MF" below).
 * Each entry point here is what a ctypes binding inside that class calls in
 * place of one cv2 call; the citation names the cv2 call site it replaces.
 *
 * Conventions: plain pointers and sizes only; every function returns
 * IMATCH_OK (0) or a negative IMATCH_E* code and never aborts;
 * imatch_last_error(ctx) gives the message. A context owns one HIP device and
 * one stream; calls on one context must be serialised by the caller (the Python
 * wrapper holds a lock per context). Host pointers unless a name says `_dev`.
 */
#ifndef IMATCH_H
#define IMATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMATCH_OK       0
#define IMATCH_EINVAL  (-1)  /* bad argument (null pointer, negative size, wrong dimension) */
#define IMATCH_EHIP    (-2)  /* a HIP runtime call failed; message holds hipGetErrorString */
#define IMATCH_ENOMEM  (-3)  /* device or host allocation failed */
#define IMATCH_ESIZE   (-4)  /* size beyond what the kernels support */

#define IMATCH_DESC_DIM 128  /* SIFT descriptor length; the only dimension the MFMA kernel takes */

typedef struct imatch_ctx imatch_ctx;
typedef struct imatch_features imatch_features;

/* Library / context ------------------------------------------------------- */

/* Replaces the capability probe match_finder.is_cuda_cv (MF:341-352): number of HIP devices, or <0. */
int imatch_device_count(void);
/* Replaces the per-call cv2.cuda_Stream()/matcher construction (MF:185-186): one device + one stream. */
int imatch_init(int device, imatch_ctx** out);
void imatch_destroy(imatch_ctx* ctx);
const char* imatch_last_error(const imatch_ctx* ctx);
/* The HIP stream of the context as a void* (hipStream_t), for callers that time with events. */
void* imatch_stream(imatch_ctx* ctx);
int imatch_synchronize(imatch_ctx* ctx);

/* Measurement hook (no reference counterpart; bench.py's roofline block): with profiling on, the
 * kNN launcher brackets its MFMA kernel with HIP events on the context's stream;
 * imatch_knn2_kernel_ms waits for the last launch and returns its duration. */
int imatch_set_profiling(imatch_ctx* ctx, int on);
int imatch_knn2_kernel_ms(imatch_ctx* ctx, float* ms);
/* Per-stage timers and trace ranges (tracing; no reference counterpart -- the reference logs one wall
 * time per frame, position_finder.py:164,219). With profiling on, every stage of the path is bracketed by HIP events on
 * the stream it runs on and by a roctx range of the same name around its enqueue (visible to rocprofv3 --marker-trace;
 * skipped when libroctx64.so cannot be loaded). imatch_stage_times waits for the brackets recorded so far and writes
 * the duration in ms of the LAST bracket of each stage into ms[0 .. n) (n <= IMATCH_STAGE_COUNT), -1 where a stage has not
 * run since profiling was switched on. */
#define IMATCH_STAGE_CLAHE        0   /* CLAHE inside imatch_sift_extract* (MF:200-203) */
#define IMATCH_STAGE_SCALE_SPACE  1   /* base blur .. last extremum scan (MF:205-209,220) */
#define IMATCH_STAGE_KEYPOINTS    2   /* refinement, orientation, sort + de-duplication */
#define IMATCH_STAGE_DESCRIBE     3   /* descriptors into the matcher's row format */
#define IMATCH_STAGE_KNN          4   /* the MFMA kNN kernel (MF:182-183) */
#define IMATCH_STAGE_KNN_MERGE    5   /* merge of the per-chunk lists */
#define IMATCH_STAGE_RATIO        6   /* ratio test + compaction (MF:191-194) */
#define IMATCH_STAGE_HOMOGRAPHY   7   /* hypothesis scoring launch of imatch_homography* (MF:233) */
#define IMATCH_STAGE_COUNT        8
int imatch_stage_times(imatch_ctx* ctx, float* ms, int n);
const char* imatch_stage_name(int stage);
/* Same bracket around the scale-space stage of the last imatch_sift_extract (upsample, the Gaussian
 * / DoG kernels, decimation and the extrema scans): the stage SURVEY 8(d)'s 475.7 B/pixel covers. */
int imatch_sift_pyramid_ms(imatch_ctx* ctx, float* ms);

/* Descriptor sets resident in HBM ----------------------------------------- */
/* Replaces cv2.cuda_GpuMat(descriptors) (MF:187-188), but the set outlives the call:
 * rows are stored as int8 [n_pad][128] (value - 128, for values that are integers 0..255) plus int32 squared norms of the
 * shifted rows: the matcher runs on the int8 MFMA in exact integer arithmetic. */
int imatch_features_from_host_f32(imatch_ctx* ctx, const float* desc, int64_t n, int d, imatch_features** out);
int imatch_features_from_host_u8(imatch_ctx* ctx, const uint8_t* desc, int64_t n, int d, imatch_features** out);
/* Same, from a u8 [n][128] matrix already in device memory (a torch tensor's data_ptr). */
int imatch_features_from_device_u8(imatch_ctx* ctx, const uint8_t* desc_dev, int64_t n, int d, imatch_features** out);
/* Same, from n_seg (<= 256) row blocks in device memory: block b holds counts[b] rows of 128 bytes at
 * desc_dev + b * seg_pitch bytes (the layout an all-gather of per-rank query blocks of a fixed capacity leaves: each
 * rank's rows at the start of its slot). The set is the blocks' rows back to back, without the unused tail of any
 * slot. Only queues work on the context's stream (u8 rows are integral by type: nothing to read back). No reference
 * counterpart (multi-GPU path). */
int imatch_features_from_device_u8_seg(imatch_ctx* ctx, const uint8_t* desc_dev, int n_seg, const int64_t* counts,
                                       size_t seg_pitch, int d, imatch_features** out);
/* Writes the set as u8 [n][128] into device memory (asynchronously, on the context's stream): the
 * exchange format of the multi-GPU path (queries are all-gathered as bytes). Integral sets only; out_dev must be
 * 4-byte aligned (the rows go out a word at a time). */
int imatch_features_desc_u8_dev(const imatch_features* f, uint8_t* out_dev);
int64_t imatch_features_count(const imatch_features* f);
/* 1 when every descriptor value of the set is an integer in 0..255 (kNN is then bit-exact), else 0. */
int imatch_features_is_integral(const imatch_features* f);
void imatch_features_free(imatch_features* f);

/* Pre-processing ----------------------------------------------------------- */
/* Replaces cv2.createCLAHE(clipLimit, tileGridSize=(tiles_x,tiles_y)).apply(img) (MF:200,203; also
 * normalize_images MF:324-326). 8-bit gray, `stride` bytes per row; out is [h][w] contiguous. */
int imatch_clahe(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, double clip_limit,
                 int tiles_x, int tiles_y, uint8_t* out);
/* Replaces cv2.resize(img, (dw, dh), interpolation=cv2.INTER_AREA) as utils.resize_img calls it
 * (utils.py:17-22; callers MF:165-175,308-311). An enlarged axis takes OpenCV's bilinear kernel with INTER_AREA's
 * coefficients, as cv2 does. */
int imatch_resize_area(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride,
                       uint8_t* out, int dh, int dw);

/* Camera pre-step ---------------------------------------- */
/* Replaces cv2.cvtColor(image, cv2.COLOR_BGR2GRAY) (position_finder.py:171, mezhcadr.py:85,
 * photo_publisher.py:112) and, with swap_rb = 1, COLOR_RGB2GRAY (image_processing.py:54). img is 8-bit with 3
 * interleaved channels, `stride` bytes per row; out is [h][w]. */
int imatch_cvt_gray(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, int swap_rb, uint8_t* out);
/* Replaces cv2.fisheye.estimateNewCameraMatrixForUndistortRectify(K, D, (w, h), np.eye(3), balance=balance)
 * (csi_camera.py:159, photo_publisher.py:164). K, newK: row-major 3x3; D: k1..k4. Host arithmetic only. */
int imatch_fisheye_new_camera_matrix(const double* K, const double* D, int w, int h, double balance, double* newK);
/* Replaces the rest of csi_camera.undistort (csi_camera.py:160-166; photo_publisher.py:165-171):
 * cv2.fisheye.initUndistortRectifyMap(K, D, eye(3), newK, (w, h), CV_16SC2) -> cv2.remap(img, map1, map2,
 * INTER_AREA (= linear inside remap), BORDER_CONSTANT) -> the slice [y0:y0+ch, x0:x0+cw] of the result. The maps are
 * built once per calibration and window and stay in HBM (the reference rebuilds them every frame); only the kept
 * window is computed. channels = 1 or 3 (interleaved); to_gray = 1 (3-channel input) returns
 * cvtColor(BGR2GRAY) of the slice, [ch][cw], the form position_finder consumes; else out is [ch][cw][channels]. */
int imatch_fisheye_undistort(imatch_ctx* ctx, const uint8_t* img, int h, int w, int channels, int stride, const double* K,
                             const double* D, const double* newK, int x0, int y0, int cw, int ch, int to_gray,
                             uint8_t* out);

/* Replaces cv2.drawMatchesKnn(cadr.img, keypoints_1, roi.img, roi.kp, good, None,
 * flags=cv2.DrawMatchesFlags_NOT_DRAW_SINGLE_POINTS) (MF:196): out is the [max(h1,h2)][w1+w2][3] canvas, img1 left,
 * img2 right (8-bit gray, replicated to three channels), and for each of the n good matches xy1[i] <-> xy2[i]
 * (f32 [n][2], the matched keypoints' .pt) a circle around both points joined by a line. Composed and drawn on the
 * GPU. Layout and dtype are cv2's; the pixels of the primitives follow "Spec D" (oracle/draw_oracle.c), since cv2's
 * anti-aliased drawing in RNG colours cannot be reproduced by any caller. */
int imatch_draw_matches(imatch_ctx* ctx, const uint8_t* img1, int h1, int w1, int stride1, const uint8_t* img2, int h2,
                        int w2, int stride2, const float* xy1, const float* xy2, int n, uint8_t* out);

/* Replaces the drawing calls that mark a result on an image: the pose overlay of find_keypoints_transform,
 * cv2.circle(img_for_pub, centre, 10, 255, 5) + cv2.polylines(img_for_pub, [np.int32(dst)], True, 255, 3, LINE_AA)
 * (MF:249-250), and the map marker of utils.draw_circle_on_map_by_coord_and_angles, cv2.line(img, p, p2, color, 2) +
 * cv2.circle(img, p, 4, color, 2) (utils.py:75-76; callers position_finder.py:333-346,382-388). img is 8-bit with 1 or 3
 * interleaved channels, `stride` bytes per row; out is [h][w][channels]; the n primitives are drawn in order (a later one
 * covers an earlier one), on the GPU. Integer coordinates, as the reference passes them. Pixels follow "Spec D"
 * (oracle/draw_oracle.c): a line of thickness t is the integer DDA line stamped with the disc 4 (u^2 + v^2) <= t^2, a
 * circle of radius r is the ring (2r - t)^2 <= 4 d^2 < (2r + t)^2 (thickness < 0: the filled disc); no anti-aliasing. */
#define IMATCH_DRAW_LINE   0   /* (x0, y0) -> (x1, y1) */
#define IMATCH_DRAW_CIRCLE 1   /* centre (x0, y0), radius x1 (y1 unused) */
typedef struct imatch_draw_prim {
    int32_t kind, x0, y0, x1, y1;
    int32_t thickness;         /* 1..255; circles: < 0 = filled */
    uint8_t color[4];          /* channel values in the image's own channel order (cv2's Scalar); 1-channel: color[0] */
    int32_t reserved;
} imatch_draw_prim;
int imatch_draw_overlay(imatch_ctx* ctx, const uint8_t* img, int h, int w, int channels, int stride,
                        const imatch_draw_prim* prims, int n, uint8_t* out);

/* SIFT extraction ---------------------------------------------------------- */
typedef struct imatch_sift_params {
    int n_octave_layers;       /* nOctaveLayers_sift   (MF:31), OpenCV default 3 */
    double contrast_threshold; /* contrastThreshold_sift (MF:32), 0.04 */
    double edge_threshold;     /* edgeThreshold_sift   (MF:33), 10 */
    double sigma;              /* sigma_sift           (MF:34), 1.6 */
    int clahe;                 /* 1: apply CLAHE first, as find_kp_dp does (MF:200-203) */
    double clahe_clip;         /* 30.0 */
    int clahe_tiles_x, clahe_tiles_y; /* 8, 8 */
} imatch_sift_params;

/* Replaces find_kp_dp's body (MF:199-221 with the SIFT call of MF:205-209):
 * [CLAHE ->] SIFT_create(0, nOctaveLayers, contrastThreshold, edgeThreshold, sigma)
 * .detectAndCompute(img, None). The result is a descriptor set resident in HBM (usable directly by
 * imatch_knn2) that also carries the keypoints and, with clahe=1, the equalised image. */
int imatch_sift_extract(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride,
                        const imatch_sift_params* params, imatch_features** out);
/* Same, for an image already in device memory (the frame hand-off from a torch tensor / a capture
 * pipeline that decodes on the GPU; the reference has no device images). */
int imatch_sift_extract_dev(imatch_ctx* ctx, const uint8_t* img_dev, int h, int w, int stride,
                            const imatch_sift_params* params, imatch_features** out);
/* Batched form (no reference counterpart: the reference handles one /photo message at a time,
 * PF:159-224; this is BASELINE configs[4], many frames per launch). n_img equally sized u8 images in
 * device memory, image b at img_dev + b * img_pitch bytes (n_img <= 256). Every kernel runs once for
 * the whole batch. The result is ONE descriptor set with the images' rows back to back:
 * offsets[b] .. offsets[b+1] (n_img + 1 entries) are image b's rows and keypoint records; each
 * image's rows are exactly what imatch_sift_extract_dev returns for it. No image copy is kept. */
int imatch_sift_extract_batch_dev(imatch_ctx* ctx, const uint8_t* img_dev, int n_img, size_t img_pitch, int h, int w,
                                  int stride, const imatch_sift_params* params, imatch_features** out,
                                  int64_t* offsets);
/* kp: [K][7] = x, y, size, angle, response, octave, class_id (cv2.KeyPoint fields, OpenCV's final
 * order: sorted and de-duplicated as KeyPointsFilter::removeDuplicatedSorted leaves them);
 * desc: [K][128] f32, integer valued 0..255. Either pointer may be null. */
int imatch_features_download(const imatch_features* f, float* kp, float* desc);
/* Gives a set made from descriptor rows alone (imatch_features_from_*: a map database read from disk, whose keypoints
 * the reference keeps beside the descriptors, match_finder.py:220-221) its keypoint records [n][7], as
 * imatch_features_download returns them: the host table and the device-side coordinate table an extraction would have
 * left, so that imatch_match_localize accepts the set. Replaces records given before. */
int imatch_features_set_keypoints(imatch_features* f, const float* kp);
/* (x, y) of the keypoints rows[0..n): what the reference gathers for findHomography with
 * `kp[m.queryIdx].pt` (MF:231-232), without copying the whole keypoint table. out_xy: [n][2] f32. */
int imatch_features_keypoints_xy(const imatch_features* f, const int64_t* rows, int64_t n, float* out_xy);
/* The image find_kp_dp returns (MF:221): the CLAHE output, [h][w] u8. */
int imatch_features_image(const imatch_features* f, uint8_t* out, int h, int w);
/* Test hook (no reference counterpart): copies one pyramid layer of the context's LAST extraction
 * to the host. kind 0 = Gaussian layer i of octave o, 1 = DoG layer (gauss[i+1] - gauss[i], formed on the host: DoG layers are not stored). rows/cols receive its size;
 * out may be null to query the size only. */
int imatch_sift_debug_layer(imatch_ctx* ctx, int kind, int o, int i, float* out, int* rows, int* cols);
int imatch_sift_debug_octaves(imatch_ctx* ctx);
/* Same hook for a band of a Gaussian layer: rows [row0, row0 + nrows) of layer i of octave o, nrows x cols floats (a
 * 32768^2 layer of a 16384^2 map is 4.3 GB; the full-size tests read it in bands). */
int imatch_sift_debug_layer_rows(imatch_ctx* ctx, int o, int i, int row0, int nrows, float* out);

/* AKAZE extraction --------------------------------------- */
typedef struct imatch_akaze_params {
    double threshold;          /* 0.0005 (MF:210) */
    int n_octaves;             /* 4 */
    int n_octave_layers;       /* 5 */
    int clahe;                 /* 1: apply CLAHE first, as find_kp_dp does (MF:200-203) */
    double clahe_clip;         /* 30.0 */
    int clahe_tiles_x, clahe_tiles_y; /* 8, 8 */
} imatch_akaze_params;

/* Replaces find_kp_dp's body as checked in (MF:199-221 with the AKAZE call of MF:210-211):
 * [CLAHE ->] cv2.AKAZE_create(descriptor_type=AKAZE_DESCRIPTOR_KAZE, threshold, nOctaves, nOctaveLayers,
 * diffusivity=KAZE_DIFF_PM_G2).detectAndCompute(img, None). The result stays with the context until the next
 * extraction: *n_out keypoints, fetched with imatch_akaze_download. */
int imatch_akaze_extract(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, const imatch_akaze_params* params,
                         int64_t* n_out);
int imatch_akaze_extract_dev(imatch_ctx* ctx, const uint8_t* img_dev, int h, int w, int stride,
                             const imatch_akaze_params* params, int64_t* n_out);
/* kp: [n][7] = x, y, size, angle, response, octave, class_id (cv2.KeyPoint fields; class_id is the evolution level, as
 * OpenCV leaves it); desc: [n][64] f32, unit length (M-SURF). Either pointer may be null. */
int imatch_akaze_download(imatch_ctx* ctx, float* kp, float* desc);
/* The image find_kp_dp returns (MF:221): the CLAHE output of the last AKAZE extraction, [h][w] u8. */
int imatch_akaze_image(imatch_ctx* ctx, uint8_t* out, int h, int w);
/* Test hooks (no reference counterpart): number of evolution levels of the last extraction, and one of its planes
 * (kind 0 Lt, 1 Lsmooth, 2 Lx, 3 Ly, 4 Ldet; 5: the contrast factor of the level's octave as a 1 x 1 plane). */
int imatch_akaze_debug_levels(imatch_ctx* ctx);
int imatch_akaze_debug_layer(imatch_ctx* ctx, int kind, int level, float* out, int* rows, int* cols);

/* Brute-force kNN, k = 2 --------------------------------------------------- */
/* Replaces cv2.BFMatcher(cv2.NORM_L2).knnMatch(q, t, k=2) (MF:182-183) and the cv2.cuda
 * matcher (MF:185-189). idx/dist are [Q][2]; a missing neighbour (N < 2) has idx -1, dist +inf.
 * dist = sqrtf((float)ssd) with ssd the exact integer sum of squared differences; ties go to
 * the lower train index, as OpenCV's ascending strict-< scan. */
int imatch_knn2(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t,
                int32_t* idx, float* dist);
/* Device-side form used by the bench and the multi-GPU path: writes int32 [Q][4] =
 * (ssd1, idx1 + idx_offset, ssd2, idx2 + idx_offset) to device memory, asynchronously on the
 * context's stream; a missing neighbour is (INT32_MAX, -1). */
int imatch_knn2_dev(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t,
                    int32_t idx_offset, int32_t* out4_dev);
/* imatch_knn2_dev for callers that cannot allocate. No counterpart in the reference: it is for code that must not call
 * hipMalloc / hipFree -- a region under HIP graph capture, a tight serving loop, a harness that forbids allocation.
 * Same arguments and the same result, bit for bit, as imatch_knn2_dev; the launcher's scratch (per-chunk partial top-2
 * lists and three bound words per query) comes from a fixed 64 MiB device array inside the library, one per GPU, held
 * until the process ends, instead of the context's growable workspace. Limit: at most 262144 queries per call (any
 * number of train rows; with many queries the map is walked in fewer, longer chunks so that the lists fit), beyond that
 * IMATCH_ESIZE. The call makes no hipMalloc, hipFree or other allocating call. All calls on one GPU share that GPU's
 * array: calls that overlap in time on one GPU (two contexts, two streams) race; serialise them. */
int imatch_knn2_noalloc_dev(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t,
                            int32_t idx_offset, int32_t* out4_dev);
/* Test hook (no reference counterpart): the number of train chunks the last imatch_knn2* launch of this context walked
 * the map in (one partial top-2 list per chunk; the fixed scratch block of imatch_knn2_noalloc_dev may force fewer). */
int imatch_knn2_debug_last_chunks(imatch_ctx* ctx);
/* The same query for float descriptors of any length d <= 128 (the 64-float M-SURF descriptors of the AKAZE path,
 * MF:182-183 with MF:210-211): dist = sqrtf of the f32 sum of squared differences accumulated in index order, ties
 * to the lower train index. Host pointers; q is [nq][d], t is [nt][d]. */
int imatch_knn2_f32(imatch_ctx* ctx, const float* q, int64_t nq, const float* t, int64_t nt, int d, int32_t* idx,
                    float* dist);
/* A float descriptor set kept in HBM (what cv2.cuda_GpuMat(roi.dp) is for one call, MF:188, kept across calls: the
 * reference matches every frame against the same ROI descriptors for up to 11 frames, PF:232-240), and the same query
 * against it: only the frame's descriptors travel. A set belongs to the context it was made on. */
typedef struct imatch_fset imatch_fset;
int imatch_fset_from_host(imatch_ctx* ctx, const float* desc, int64_t n, int d, imatch_fset** out);
int64_t imatch_fset_count(const imatch_fset* f);
void imatch_fset_free(imatch_fset* f);
int imatch_knn2_fset(imatch_ctx* ctx, const float* q, int64_t nq, const imatch_fset* t, int32_t* idx, float* dist);
/* The frame's side of the same call without the round trip: the reference hands find_matches the descriptor matrix
 * find_kp_dp has just returned (PF:202-203, 232-240 -> MF:182-183), and after imatch_akaze_extract that matrix is still
 * on the device. imatch_akaze_last names the context's last AKAZE extraction (*serial, 0 if there is none or it found no
 * keypoint; *n its keypoints, may be null); imatch_knn2_fset_akaze matches THAT extraction's descriptors, where they
 * lie, against a resident set: idx/dist [n][2] as imatch_knn2_fset gives them for the downloaded matrix, bit for bit.
 * IMATCH_EINVAL when another extraction has run on the context since (the caller then uploads its host copy). */
/* Resident sets of 4096 rows and more are answered by a filter on the f16 matrix cores (rows that can be among a query's
 * two nearest, with a proven error bound) followed by the exact f32 chain on those rows only -- the same bits as the
 * exhaustive kernel, which remains the path of small sets, of sets with values beyond +-1024 or non-finite ones, of more
 * than 16384 queries per call and of calls whose candidate lists overflow. Test hook: the path the context's last float
 * kNN call took (0 exhaustive, 1 filter + refinement, 2 filter tried, then exhaustive). */
int imatch_knn2_f32_debug_last_path(imatch_ctx* ctx);
int imatch_akaze_last(imatch_ctx* ctx, uint64_t* serial, int64_t* n);
int imatch_knn2_fset_akaze(imatch_ctx* ctx, uint64_t serial, const imatch_fset* t, int32_t* idx, float* dist);
/* Merge step after the all-gather of per-shard top-2 (no reference counterpart: the reference is
 * single-process). in4_dev is int32 [n_lists][Q][4]; picks the two smallest by
 * (ssd, idx) lexicographic order into out4_dev [Q][4]. */
int imatch_merge_top2_dev(imatch_ctx* ctx, const int32_t* in4_dev, int n_lists, int64_t nq,
                          int32_t* out4_dev);
/* Replaces the Python ratio loop `m.distance < points_quality * n.distance` (MF:191-194):
 * evaluates double(f32 d1) < ratio * double(f32 d2) per query on out4 (device), compacts the
 * survivors in query order. good_q/good_t/good_d are host arrays of capacity nq. */
int imatch_ratio_dev(imatch_ctx* ctx, const int32_t* out4_dev, int64_t nq, double ratio,
                     int32_t* good_q, int32_t* good_t, float* good_d, int64_t* n_good);
/* The same test on the rows of a batch (no reference counterpart; pairs with
 * imatch_sift_extract_batch_dev): segment b = queries offsets[b] .. offsets[b+1] (offsets[0] = 0,
 * n_seg <= 64). Segment b's survivors are good_*[good_offsets[b] .. good_offsets[b+1]) with
 * good_q LOCAL to the segment, i.e. what imatch_ratio_dev returns for that image alone.
 * good_* have capacity offsets[n_seg]; good_offsets has n_seg + 1 entries. */
int imatch_ratio_batch_dev(imatch_ctx* ctx, const int32_t* out4_dev, const int64_t* offsets, int n_seg,
                           double ratio, int32_t* good_q, int32_t* good_t, float* good_d,
                           int64_t* good_offsets);

/* Robust homography -------------------------------------------------------- */
/* Replaces cv2.findHomography(src_pts, dst_pts, cv2.USAC_MAGSAC, thr) (MF:233). src/dst are f32
 * [n][2] (the reference passes kp[].pt gathered as float32, MF:231-232). On success *found = 1,
 * H is row-major 3x3 with H[8] = 1 and mask[i] = 1 where the forward reprojection error is <= thr.
 * *found = 0 (H untouched, mask zero) when there is no model: n < 4 or only degenerate samples --
 * the case in which cv2 returns (None, None). All hypotheses are scored on the GPU at once;
 * `seed` makes the sampling reproducible. mask may be null. */
int imatch_homography(imatch_ctx* ctx, const float* src, const float* dst, int n, double thr,
                      uint64_t seed, double* H, uint8_t* mask, int* found);
/* n_seg independent problems in one launch (no reference counterpart; pairs with the batch entry
 * points above): problem b is src/dst rows offsets[b] .. offsets[b+1] with sampler seed seeds[b];
 * H is [n_seg][9], found [n_seg], mask [offsets[n_seg]] (may be null). Each problem's result equals
 * imatch_homography on its rows with its seed. */
int imatch_homography_batch(imatch_ctx* ctx, const float* src, const float* dst, const int64_t* offsets,
                            int n_seg, double thr, const uint64_t* seeds, double* H, uint8_t* mask,
                            int* found);
/* One frame's match and localisation in one call: what position_finder.py:294-318 does with a frame's descriptors
 * (find_matches, MF:175-197; `if len(good) > 10`; find_keypoints_transform, MF:224-236) = imatch_knn2_dev +
 * imatch_ratio_dev + the lookup of the matched keypoints' coordinates + imatch_homography, without the host round trip
 * between the ratio test and the hypothesis scoring (one workgroup tests, compacts, gathers and normalises where the
 * kNN's result lies; the scoring kernel reads the count on the device). Both sets must come from imatch_sift_extract*
 * on this context (they carry their keypoints' coordinates, on the host and on the device). good_q / good_t / good_dist
 * hold imatch_features_count(q) entries; *n_good survivors are written. A homography is looked for when
 * *n_good > min_good (and >= 4); H, mask (may be null, *n_good bytes) and *found as imatch_homography's. Every output
 * equals the separate calls' bit for bit (more than 1024 survivors, or device-side transforms that are not the host's,
 * take those calls internally). */
int imatch_match_localize(imatch_ctx* ctx, const imatch_features* q, const imatch_features* t, double ratio,
                          int min_good, double thr, uint64_t seed, int32_t* good_q, int32_t* good_t,
                          float* good_dist, int64_t* n_good, double* H, uint8_t* mask, int* found);
/* test hook: calls of imatch_match_localize on this context whose device-side Hartley transforms differed from the
 * host's (they then went through imatch_homography; expected to stay 0) */
int imatch_match_localize_debug_mismatches(imatch_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* IMATCH_H */
