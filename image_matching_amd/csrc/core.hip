// Context, error handling and workspace management of libimatch.so.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <stdarg.h>
#include <stdlib.h>

int imatch_set_err(imatch_ctx* ctx, int code, const char* fmt, ...) {
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

int imatch_ws_reserve(imatch_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return IMATCH_OK;
    // the old block may still be in use by queued kernels
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) HIP_TRY(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    size_t want = bytes + bytes / 4;
    if (hipMalloc(&ctx->ws, want) != hipSuccess)
        return imatch_set_err(ctx, IMATCH_ENOMEM, "workspace of %zu bytes", want);
    ctx->ws_bytes = want;
    return IMATCH_OK;
}

static int is_reserve(imatch_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->is_bytes) return IMATCH_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->is) HIP_TRY(ctx, hipHostFree(ctx->is));
    ctx->is = nullptr;
    ctx->is_bytes = 0;
    if (hipHostMalloc(&ctx->is, bytes, hipHostMallocDefault) != hipSuccess)
        return imatch_set_err(ctx, IMATCH_ENOMEM, "pinned image staging of %zu bytes", bytes);
    ctx->is_bytes = bytes;
    return IMATCH_OK;
}

// the block is split in halves: [0, half) uploads, [half, 2*half) downloads
static size_t is_half(const imatch_ctx* ctx) { return ctx->is_bytes / 2; }

int imatch_stage_upload(imatch_ctx* ctx, const uint8_t* img, int h, int w, int stride, uint8_t* dev) {
    const size_t bytes = (size_t)h * w;
    if (bytes > is_half(ctx)) {
        int rc = is_reserve(ctx, 2 * (bytes + bytes / 4 + 4096));
        if (rc) return rc;
    }
    uint8_t* st = (uint8_t*)ctx->is;
    if (stride == w) memcpy(st, img, bytes);
    else
        for (int y = 0; y < h; ++y) memcpy(st + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
    HIP_TRY(ctx, hipMemcpyAsync(dev, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    return IMATCH_OK;
}

int imatch_stage_download_begin(imatch_ctx* ctx, const uint8_t* dev, size_t bytes) {
    if (bytes > is_half(ctx)) {
        int rc = is_reserve(ctx, 2 * (bytes + bytes / 4 + 4096));   // synchronises: nothing of this call is lost, the
        if (rc) return rc;                                          // upload half has been consumed by then
    }
    HIP_TRY(ctx, hipMemcpyAsync((uint8_t*)ctx->is + is_half(ctx), dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return IMATCH_OK;
}

void imatch_stage_download_end(imatch_ctx* ctx, uint8_t* out, size_t bytes) {
    memcpy(out, (const uint8_t*)ctx->is + is_half(ctx), bytes);
}

// ---- host worker pool ------------------------------------------------------------------------------
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

struct imatch_workers {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)>* fn = nullptr;
    int n = 0;                 // items of the current job
    std::atomic<int> next{0};  // next item to take
    int running = 0;           // workers that have not finished the current job
    unsigned long job = 0;     // job counter: a worker runs each job once
    bool stop = false;
};

static void worker_main(imatch_workers* w) {
    unsigned long seen = 0;
    for (;;) {
        std::unique_lock<std::mutex> lk(w->m);
        w->cv_work.wait(lk, [&] { return w->stop || w->job != seen; });
        if (w->stop) return;
        seen = w->job;
        const std::function<void(int)>* fn = w->fn;
        const int n = w->n;
        lk.unlock();
        for (int i = w->next.fetch_add(1); i < n; i = w->next.fetch_add(1)) (*fn)(i);
        lk.lock();
        if (--w->running == 0) w->cv_done.notify_one();
    }
}

void imatch_parallel_for(imatch_ctx* ctx, int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || !ctx) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    if (!ctx->workers) {
        ctx->workers = new imatch_workers();
        const unsigned hw = std::thread::hardware_concurrency();
        const int nt = (int)std::min<unsigned>(16u, hw ? hw : 4u) - 1;   // the caller works too
        for (int t = 0; t < nt; ++t) ctx->workers->threads.emplace_back(worker_main, ctx->workers);
    }
    imatch_workers* w = ctx->workers;
    {
        std::lock_guard<std::mutex> lk(w->m);
        w->fn = &fn;
        w->n = n;
        w->next.store(0);
        w->running = (int)w->threads.size();
        ++w->job;
    }
    w->cv_work.notify_all();
    for (int i = w->next.fetch_add(1); i < n; i = w->next.fetch_add(1)) fn(i);
    std::unique_lock<std::mutex> lk(w->m);
    w->cv_done.wait(lk, [&] { return w->running == 0; });
    w->fn = nullptr;
}

void imatch_workers_destroy(imatch_ctx* ctx) {
    if (!ctx || !ctx->workers) return;
    {
        std::lock_guard<std::mutex> lk(ctx->workers->m);
        ctx->workers->stop = true;
    }
    ctx->workers->cv_work.notify_all();
    for (auto& t : ctx->workers->threads) t.join();
    delete ctx->workers;
    ctx->workers = nullptr;
}

int imatch_hs_mark_busy(imatch_ctx* ctx) {
    if (!ctx->hs_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->hs_ev, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->hs_ev, ctx->stream));
    ctx->hs_pending = 1;
    return IMATCH_OK;
}

int imatch_hs_reserve(imatch_ctx* ctx, size_t bytes) {
    if (ctx->hs_pending) {   // a queued copy still reads the block: the caller is about to write into it
        HIP_TRY(ctx, hipEventSynchronize(ctx->hs_ev));
        ctx->hs_pending = 0;
    }
    if (bytes <= ctx->hs_bytes) return IMATCH_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->hs) HIP_TRY(ctx, hipHostFree(ctx->hs));
    ctx->hs = nullptr;
    ctx->hs_bytes = 0;
    size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&ctx->hs, want, hipHostMallocDefault) != hipSuccess)
        return imatch_set_err(ctx, IMATCH_ENOMEM, "pinned staging of %zu bytes", want);
    ctx->hs_bytes = want;
    return IMATCH_OK;
}

// ---- device block pool -------------------------------------------------------------------------
static const size_t POOL_KEEP_BYTES = (size_t)2 << 30;  // free blocks kept at most, in total

void* imatch_pool_alloc(imatch_ctx* ctx, size_t bytes) {
    if (bytes == 0) bytes = 256;
    int best = -1;
    for (int i = 0; i < ctx->pool_n; ++i) {
        const imatch_pool_block& b = ctx->pool[i];
        if (!b.in_use && b.bytes >= bytes && b.bytes <= 2 * bytes + 65536 && (best < 0 || b.bytes < ctx->pool[best].bytes))
            best = i;
    }
    if (best >= 0) {
        ctx->pool[best].in_use = 1;
        return ctx->pool[best].ptr;
    }
    if (ctx->pool_n == ctx->pool_cap) {
        const int cap = ctx->pool_cap ? 2 * ctx->pool_cap : 32;
        imatch_pool_block* nb = (imatch_pool_block*)realloc(ctx->pool, sizeof(imatch_pool_block) * cap);
        if (!nb) return nullptr;
        ctx->pool = nb;
        ctx->pool_cap = cap;
    }
    // round up so that frames with slightly different keypoint counts share blocks
    size_t want = bytes < 65536 ? 65536 : bytes + bytes / 8;
    want = (want + 65535) & ~(size_t)65535;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        // give the cached free blocks back and try once more
        (void)hipStreamSynchronize(ctx->stream);
        int j = 0;
        for (int i = 0; i < ctx->pool_n; ++i) {
            if (ctx->pool[i].in_use) ctx->pool[j++] = ctx->pool[i];
            else (void)hipFree(ctx->pool[i].ptr);
        }
        ctx->pool_n = j;
        if (hipMalloc(&p, want) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    ctx->pool[ctx->pool_n++] = imatch_pool_block{p, want, 1};
    return p;
}

void imatch_pool_free(imatch_ctx* ctx, void* p) {
    if (!p) return;
    size_t free_bytes = 0;
    int at = -1;
    for (int i = 0; i < ctx->pool_n; ++i) {
        if (ctx->pool[i].ptr == p) at = i;
        else if (!ctx->pool[i].in_use) free_bytes += ctx->pool[i].bytes;
    }
    if (at < 0) return;  // not ours
    if (free_bytes + ctx->pool[at].bytes > POOL_KEEP_BYTES) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        ctx->pool[at] = ctx->pool[--ctx->pool_n];
    } else {
        ctx->pool[at].in_use = 0;
    }
}

void imatch_pool_destroy(imatch_ctx* ctx) {
    for (int i = 0; i < ctx->pool_n; ++i) (void)hipFree(ctx->pool[i].ptr);
    free(ctx->pool);
    ctx->pool = nullptr;
    ctx->pool_n = ctx->pool_cap = 0;
}

extern "C" {

int imatch_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return IMATCH_EHIP;
    return n;
}

int imatch_init(int device, imatch_ctx** out) {
    if (!out) return IMATCH_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return IMATCH_EHIP;
    imatch_ctx* c = (imatch_ctx*)calloc(1, sizeof(imatch_ctx));
    if (!c) return IMATCH_ENOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        free(c);
        return IMATCH_EHIP;
    }
    if (imatch_knn2_scratch_setup(c) != IMATCH_OK) {   // per-device address of the fixed kNN scratch block
        (void)hipStreamDestroy(c->stream);
        free(c);
        return IMATCH_EHIP;
    }
    *out = c;
    return IMATCH_OK;
}

void imatch_destroy(imatch_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    imatch_sift_state_free(ctx);
    imatch_camera_state_free(ctx);
    imatch_akaze_state_free(ctx);
    imatch_pool_destroy(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    imatch_workers_destroy(ctx);
    if (ctx->hs) (void)hipHostFree(ctx->hs);
    if (ctx->is) (void)hipHostFree(ctx->is);
    for (int s = 0; s < IMATCH_STAGE_COUNT; ++s)
        for (int e = 0; e < 2; ++e)
            if (ctx->st_ev[s][e]) (void)hipEventDestroy(ctx->st_ev[s][e]);
    if (ctx->hs_ev) (void)hipEventDestroy(ctx->hs_ev);
    (void)hipStreamDestroy(ctx->stream);
    free(ctx);
}

const char* imatch_last_error(const imatch_ctx* ctx) { return ctx ? ctx->err : "null context"; }

void* imatch_stream(imatch_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// roctx ranges, when the tracer's marker library is there (it is part of ROCm but not a link dependency of this library)
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
static roctx_push_fn roctx_push = nullptr;
static roctx_pop_fn roctx_pop = nullptr;
static std::once_flag roctx_once;
static void roctx_load() {
    std::call_once(roctx_once, [] {
        void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
        roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
        if (!roctx_push || !roctx_pop) roctx_push = nullptr, roctx_pop = nullptr;
    });
}

static const char* const stage_names[IMATCH_STAGE_COUNT] = {"imatch:clahe", "imatch:scale_space", "imatch:keypoints",
                                                            "imatch:describe", "imatch:knn", "imatch:knn_merge",
                                                            "imatch:ratio", "imatch:homography_score"};

const char* imatch_stage_name(int stage) {
    return stage >= 0 && stage < IMATCH_STAGE_COUNT ? stage_names[stage] + 7 : "";
}

int imatch_set_profiling(imatch_ctx* ctx, int on) {
    if (!ctx) return IMATCH_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->st_ev[0][0]) {
        for (int s = 0; s < IMATCH_STAGE_COUNT; ++s)
            for (int e = 0; e < 2; ++e) HIP_TRY(ctx, hipEventCreate(&ctx->st_ev[s][e]));
        roctx_load();
    }
    if (on && !ctx->prof) ctx->st_seen = 0;
    ctx->prof = on ? 1 : 0;
    return IMATCH_OK;
}

extern "C++" int imatch_stage_begin(imatch_ctx* ctx, int stage, hipStream_t st) {
    if (!ctx->prof) return IMATCH_OK;
    if (roctx_push) (void)roctx_push(stage_names[stage]);
    HIP_TRY(ctx, hipEventRecord(ctx->st_ev[stage][0], st));
    return IMATCH_OK;
}

extern "C++" int imatch_stage_end(imatch_ctx* ctx, int stage, hipStream_t st) {
    if (!ctx->prof) return IMATCH_OK;
    if (roctx_pop) (void)roctx_pop();
    HIP_TRY(ctx, hipEventRecord(ctx->st_ev[stage][1], st));
    ctx->st_seen |= 1u << stage;
    return IMATCH_OK;
}

static int stage_ms(imatch_ctx* ctx, int stage, float* ms) {
    if (!ctx->st_ev[0][0]) return imatch_set_err(ctx, IMATCH_EINVAL, "profiling was never enabled");
    if (!(ctx->st_seen & (1u << stage))) { *ms = -1.f; return IMATCH_OK; }
    HIP_TRY(ctx, hipEventSynchronize(ctx->st_ev[stage][1]));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->st_ev[stage][0], ctx->st_ev[stage][1]));
    return IMATCH_OK;
}

int imatch_stage_times(imatch_ctx* ctx, float* ms, int n) {
    if (!ctx || !ms || n < 0 || n > IMATCH_STAGE_COUNT) return imatch_set_err(ctx, IMATCH_EINVAL, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int s = 0; s < n; ++s) {
        const int rc = stage_ms(ctx, s, ms + s);
        if (rc) return rc;
    }
    return IMATCH_OK;
}

int imatch_knn2_kernel_ms(imatch_ctx* ctx, float* ms) {
    if (!ctx || !ms) return IMATCH_EINVAL;
    const int rc = stage_ms(ctx, IMATCH_STAGE_KNN, ms);
    if (rc == IMATCH_OK && *ms < 0) return imatch_set_err(ctx, IMATCH_EINVAL, "no kNN launch since profiling was switched on");
    return rc;
}

int imatch_sift_pyramid_ms(imatch_ctx* ctx, float* ms) {
    if (!ctx || !ms) return IMATCH_EINVAL;
    const int rc = stage_ms(ctx, IMATCH_STAGE_SCALE_SPACE, ms);
    if (rc == IMATCH_OK && *ms < 0) return imatch_set_err(ctx, IMATCH_EINVAL, "no extraction since profiling was switched on");
    return rc;
}

int imatch_synchronize(imatch_ctx* ctx) {
    if (!ctx) return IMATCH_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IMATCH_OK;
}

}  // extern "C"
