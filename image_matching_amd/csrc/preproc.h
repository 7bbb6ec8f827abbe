// Device-pointer forms of the pre-processing stages, shared with the SIFT pipeline.
#pragma once
#include "common.h"

// scratch: at least tiles_x*tiles_y*1280 bytes of device memory
int imatch_clahe_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, double clip_limit, int tiles_x,
                     int tiles_y, uint8_t* dst_dev, void* scratch);
int imatch_resize_area_dev(imatch_ctx* ctx, const uint8_t* src_dev, int h, int w, int stride, uint8_t* dst_dev, int dh,
                           int dw);
