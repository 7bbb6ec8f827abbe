// Device-pointer forms of the pre-processing stages, shared with the SIFT pipeline.
#pragma once
#include "common.h"

// scratch: at least tiles_x*tiles_y*1280 bytes of device memory. batch > 1: image b reads src_dev + b*src_zs and uses
// dst_dev + b*zs, scratch + b*zs (one launch per stage for the whole batch).
// clear64 (optional): 64 ints per image (zs bytes apart) that the caller wants zeroed before its next kernel -- the
// one-workgroup-per-tile form does it on the way (*cleared = true) and saves the caller a clear of its own.
int imatch_clahe_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, double clip_limit, int tiles_x,
                     int tiles_y, uint8_t* dst_dev, void* scratch, int batch = 1, size_t src_zs = 0, size_t zs = 0,
                     int* clear64 = nullptr, bool* cleared = nullptr);
int imatch_resize_area_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, uint8_t* dst_dev, int dh,
                           int dw);
