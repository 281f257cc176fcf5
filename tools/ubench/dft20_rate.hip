// Microbenchmark: how close does the compiled 20-point DFT (fft_core.h) come to the VALU issue peak?
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../speech_to_text_amd/csrc/fft_core.h"
using namespace sttfe;
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    cf v[20], y[20];
    for (int i = 0; i < 20; ++i) v[i] = cf{threadIdx.x * 1e-3f + i, 0.5f * i - threadIdx.x * 1e-4f};
    for (int it = 0; it < iters; ++it) {
        dft20(v, y);
#pragma unroll
        for (int i = 0; i < 20; ++i) v[i] = cf{y[i].x * 0.05f, y[i].y * 0.05f};      // 40 more VALU ops, keeps values bounded
    }
    float s = 0;
    for (int i = 0; i < 20; ++i) s += v[i].x + v[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; ++wps) {
        dim3 grid(256 * wps), block(256);
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, grid, block, 0, 0, out, iters);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD=%d  %.3f ms -> %.0f ns per (dft20 + 40 mul) per SIMD-wave slot, i.e. %.2f ns per wave per SIMD\n",
               wps, ms, ms * 1e6 / iters, ms * 1e6 / iters / wps);
    }
    return 0;
}
