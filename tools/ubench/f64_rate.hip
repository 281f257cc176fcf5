// Microbenchmark: issue cost of the float64 VALU forms the HF-parity Kaldi kernel uses (gfx950), per wave instruction
// and SIMD at 1 / 2 / 3 waves per SIMD: add, mul, fma, f32 <-> f64 conversions, a DPP row move of a 64-bit value.
// hipcc --offload-arch=gfx950 -O3 -o f64_rate f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    double a[8]; float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
    const double c = 1.0001, d = 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = a[i] + c;
                else if (MODE == 1) a[i] = a[i] * c;
                else if (MODE == 2) a[i] = __builtin_fma(a[i], c, d);
                else if (MODE == 3) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i])); }
                else if (MODE == 4) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i])); }
                else if (MODE == 5) { asm volatile("v_max_f64 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c)); }
                else { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(f[i]) : "v"(f[(i + 1) & 7])); }
            }
        }
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i];
    if (s == 12345.678) out[threadIdx.x] = s;
}
int main() {
    double* out; (void)hipMalloc(&out, 4096);
    const int iters = 4000;
    const char* names[7] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_max_f64", "v_mov_b32_dpp"};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 7; ++mode)
        for (int wps = 1; wps <= 3; ++wps) {
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
            double instr_per_simd = (double)iters * 64.0 * wps;
            printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave instruction and SIMD @2.4GHz\n",
                   names[mode], wps, ms, ms * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}
