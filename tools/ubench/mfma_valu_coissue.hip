// Microbenchmark: do exact-f32 MFMAs (v_mfma_f32_16x16x4_f32) of one wave overlap packed-f32 VALU work of ANOTHER wave
// on the same SIMD?  Workgroups of 8 waves (two per SIMD): waves 0-3 run role A, waves 4-7 role B.
//   mode 0: A = MFMA, B = MFMA      mode 1: A = VALU, B = VALU      mode 2: A = MFMA, B = VALU
//   mode 3: A = MFMA, B idle        mode 4: A = VALU, B idle        mode 5: one wave alternating 4 MFMA + 16 VALU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_block(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, float x, float y) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
}
typedef short bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma_bf16_block(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, bf8 x, bf8 y) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a3, 0, 0, 0);
}
__device__ __forceinline__ void valu_block(f2 (&p)[8], f2 c, f2 d) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], c, d);
}
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    const bool roleA = wave < 4;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f2 p[8];
    for (int i = 0; i < 8; ++i) p[i] = f2{threadIdx.x * 1e-3f + i, 1.f};
    const float x = threadIdx.x * 1e-4f, y = 1.0f - x;
    const f2 c = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
    bf8 bx, by;
    for (int i = 0; i < 8; ++i) { bx[i] = (short)(0x3f80 + threadIdx.x + i); by[i] = (short)(0x3f00 + i); }
    const bool do_bf16 = ((MODE == 6 || MODE == 7) && roleA) || MODE == 8;
    const bool do_mfma = (MODE == 0) || ((MODE == 2 || MODE == 3) && roleA);
    const bool do_valu = (MODE == 1) || (MODE == 2 && !roleA) || (MODE == 4 && roleA) || (MODE == 6 && !roleA);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 5) {
        if (roleA)
            for (int it = 0; it < iters; ++it) { mfma_block(a0, a1, a2, a3, x, y); valu_block(p, c, d); }
    } else if (MODE == 9) {
        if (roleA)
            for (int it = 0; it < iters; ++it) { mfma_bf16_block(a0, a1, a2, a3, bx, by); valu_block(p, c, d); }
    } else if (do_bf16) {
        for (int it = 0; it < iters; ++it) mfma_bf16_block(a0, a1, a2, a3, bx, by);
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) mfma_block(a0, a1, a2, a3, x, y);
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) valu_block(p, c, d);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* out; (void)hipMalloc(&out, 8192);
    unsigned long long* cyc; (void)hipMalloc(&cyc, 64);
    const int iters = 20000;
    const char* names[10] = {"MFMA | MFMA", "VALU | VALU", "MFMA | VALU", "MFMA | idle", "VALU | idle", "one wave: 4 MFMA + 16 pk_fma",
                             "bf16 MFMA | VALU", "bf16 MFMA | idle", "bf16 MFMA | bf16 MFMA", "one wave: 4 bf16 MFMA + 16 pk_fma"};
    for (int mode = 0; mode < 10; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(256), dim3(512), 0, 0, out, iters, cyc);
            (void)hipDeviceSynchronize();
        }
        unsigned long long h[8];
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%-30s  role A wave: %.1f cycles per block (4 MFMA or 16 pk_fma)   role B wave: %.1f\n", names[mode],
               (double)h[0] / iters, (double)h[4] / iters);
    }
    return 0;
}
