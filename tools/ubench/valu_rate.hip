// Microbenchmark: chip-wide issue rate of plain vs packed f32 VALU on gfx950 (wall clock, HIP events).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8]; f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    const float c = 1.0001f, d = 0.5f;
    const f2 c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, d);
                else if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], c2, d2);
                else if (MODE == 2) a[i] = a[i] + c;
                else if (MODE == 3) p[i] = p[i] + c2;
                else if (MODE == 4) a[0] = __builtin_fmaf(a[0], c, d);                 // one dependent chain
                else if (MODE == 5) a[i & 1] = __builtin_fmaf(a[i & 1], c, d);         // two chains
                else a[i & 3] = __builtin_fmaf(a[i & 3], c, d);                        // four chains
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}
int main() {
    float* out; (void)hipMalloc(&out, 4096);
    const int iters = 4000;
    const char* names[7] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "fma 1 chain", "fma 2 chains", "fma 4 chains"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 7; ++mode)
        for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD: grid = 256 CUs * wps blocks of 4 waves
            dim3 grid(256 * wps), block(256);
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, out, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, grid, block, 0, 0, out, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, grid, block, 0, 0, out, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, grid, block, 0, 0, out, iters);
                if (mode == 4) hipLaunchKernelGGL(k<4>, grid, block, 0, 0, out, iters);
                if (mode == 5) hipLaunchKernelGGL(k<5>, grid, block, 0, 0, out, iters);
                if (mode == 6) hipLaunchKernelGGL(k<6>, grid, block, 0, 0, out, iters);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = (double)iters * 64.0 * wps;          // each SIMD hosts wps waves
            printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n",
                   names[mode], wps, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}
