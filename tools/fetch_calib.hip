// Calibration of rocprofv3's FETCH_SIZE against a known byte count for the access widths of the extremum scan:
// each lane reads V floats (V = 1, 2, 4: 4 / 8 / 16 bytes) of a row segment per load, rows walked downwards, as
// detect_stream_kernel / detect_streamv_kernel<V> read their layers. MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half
// the bytes of 16-byte-per-lane streaming reads and "other access widths are uncalibrated: calibrate on a known byte
// count in your own access pattern". Run under `rocprofv3 --pmc FETCH_SIZE`; each kernel reads BYTES exactly once.
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o tools/bin/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int V>
__global__ void __launch_bounds__(256) read_rows_kernel(const float* __restrict__ src, int rows, int cols, int rpw,
                                                        float* __restrict__ sink) {
    typedef float vf __attribute__((ext_vector_type(V)));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 64 + lane) * V;
    const int r0 = (blockIdx.y * 4 + wv) * rpw;
    if (c0 >= cols) return;
    float acc = 0.f;
    for (int r = r0; r < min(r0 + rpw, rows); ++r) {
        const vf v = *(const vf*)(src + (size_t)r * cols + c0);
#pragma unroll
        for (int e = 0; e < V; ++e) acc += v[e];
    }
    if (acc == 123.456f) sink[0] = acc;   // never true: keeps the loads
}

template <int V>
static void run(const float* d, int rows, int cols, float* sink) {
    const int rpw = 64;
    dim3 grid((cols / V + 63) / 64, (rows + 4 * rpw - 1) / (4 * rpw));
    hipLaunchKernelGGL(read_rows_kernel<V>, grid, dim3(256), 0, 0, d, rows, cols, rpw, sink);
    (void)hipDeviceSynchronize();
}

int main() {
    const int rows = 16384, cols = 32768;   // 2 GiB: eight times the Infinity Cache
    float *d, *sink;
    if (hipMalloc(&d, (size_t)rows * cols * 4) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("no memory\n"); return 1; }
    (void)hipMemset(d, 0, (size_t)rows * cols * 4);
    (void)hipDeviceSynchronize();
    run<1>(d, rows, cols, sink);
    run<2>(d, rows, cols, sink);
    run<4>(d, rows, cols, sink);
    printf("each read_rows_kernel<V> launch reads %zu KB\n", (size_t)rows * cols * 4 / 1024);
    return 0;
}
