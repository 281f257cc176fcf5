// Microbenchmark: what a 4-byte global load costs the address path as a function of the ACTIVE lanes of the wave.
// Each wave issues 64 x 16 dword loads from a small L1 / L2-resident buffer, all 64 lanes active or only lanes >= first;
// 8 waves per CU keep the address unit saturated.  hipcc --offload-arch=gfx950 -O3 -o ta_exec ta_exec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* __restrict__ y, int first, int reps) {
    const int lane = threadIdx.x & 63;
    const float* p = x + (threadIdx.x >> 6) * 4096 + lane;
    float acc = 0.f;
    if (lane >= first) {
        for (int r = 0; r < reps; ++r) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = p[20 * i + ((r & 3) << 8)];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i];
            asm volatile("" : "+v"(acc));
        }
    }
    if (acc == 1234.5f) y[threadIdx.x] = acc;
}
int main() {
    float *x, *y;
    hipMalloc(&x, 1 << 20); hipMalloc(&y, 1 << 16);
    hipMemset(x, 0, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 2000;
    for (int first : {0, 0, 16, 32, 40, 48, 60}) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, x, y, first, reps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, x, y, first, reps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double loads_per_cu = 8.0 * reps * 16;
        printf("active lanes %2d..63 (%2d): %.3f ms = %.1f ns per wave load instruction and CU = %.1f cycles at 2.4 GHz\n", first, 64 - first, ms, ms * 1e6 / loads_per_cu, ms * 1e6 / loads_per_cu * 2.4);
    }
    return 0;
}
