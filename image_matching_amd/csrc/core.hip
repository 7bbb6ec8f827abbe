// Context, error handling and workspace management of libimatch.so.
#include "common.h"
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

int imatch_hs_reserve(imatch_ctx* ctx, size_t bytes) {
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
    *out = c;
    return IMATCH_OK;
}

void imatch_destroy(imatch_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    imatch_sift_state_free(ctx);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->hs) (void)hipHostFree(ctx->hs);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
    if (ctx->ev3) (void)hipEventDestroy(ctx->ev3);
    (void)hipStreamDestroy(ctx->stream);
    free(ctx);
}

const char* imatch_last_error(const imatch_ctx* ctx) { return ctx ? ctx->err : "null context"; }

void* imatch_stream(imatch_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int imatch_set_profiling(imatch_ctx* ctx, int on) {
    if (!ctx) return IMATCH_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->ev0) {
        HIP_TRY(ctx, hipEventCreate(&ctx->ev0));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev1));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev2));
        HIP_TRY(ctx, hipEventCreate(&ctx->ev3));
    }
    ctx->prof = on ? 1 : 0;
    return IMATCH_OK;
}

int imatch_knn2_kernel_ms(imatch_ctx* ctx, float* ms) {
    if (!ctx || !ms) return IMATCH_EINVAL;
    if (!ctx->ev0) return imatch_set_err(ctx, IMATCH_EINVAL, "profiling was never enabled");
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return IMATCH_OK;
}

int imatch_sift_pyramid_ms(imatch_ctx* ctx, float* ms) {
    if (!ctx || !ms) return IMATCH_EINVAL;
    if (!ctx->ev2) return imatch_set_err(ctx, IMATCH_EINVAL, "profiling was never enabled");
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev3));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev2, ctx->ev3));
    return IMATCH_OK;
}

int imatch_synchronize(imatch_ctx* ctx) {
    if (!ctx) return IMATCH_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return IMATCH_OK;
}

}  // extern "C"
