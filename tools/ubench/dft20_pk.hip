// Microbenchmark: packed-f32 (v_pk_* with op_sel swizzles) 20-point DFT vs the scalar form of dft20_rate.hip;
// prints the maximum error of one transform against a double-precision DFT first.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mul_mi(f2 a) { return f2{a.y, -a.x}; }
__device__ __forceinline__ f2 mul_pi(f2 a) { return f2{-a.y, a.x}; }
__device__ __forceinline__ f2 cmul(f2 a, f2 b) {
    // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
    f2 t = a.yx * f2{-b.y, b.x};         // (-a.y b.y, a.x b.x)?? careful
    return __builtin_elementwise_fma(f2{a.x, a.x}, b, f2{a.y, a.y} * f2{-b.y, b.x});
}
// a + (-i) b = (a.x + b.y, a.y - b.x);  a + (+i) b = (a.x - b.y, a.y + b.x): one packed add with operand swizzle
__device__ __forceinline__ f2 add_mi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 add_pi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void dft4(f2& a0, f2& a1, f2& a2, f2& a3) {
    f2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13; a2 = s02 - s13; a1 = add_mi(d02, d13); a3 = add_pi(d02, d13);
}
__device__ __forceinline__ void dft5(f2& a0, f2& a1, f2& a2, f2& a3, f2& a4) {
    constexpr float C1 = 0.30901699437494742f, C2 = -0.80901699437494742f, S1 = 0.95105651629515357f, S2 = 0.58778525229247313f;
    f2 t1 = a1 + a4, t2 = a2 + a3, t3 = a1 - a4, t4 = a2 - a3;
    f2 m1 = __builtin_elementwise_fma(f2{C2, C2}, t2, __builtin_elementwise_fma(f2{C1, C1}, t1, a0));
    f2 m2 = __builtin_elementwise_fma(f2{C1, C1}, t2, __builtin_elementwise_fma(f2{C2, C2}, t1, a0));
    f2 s1 = __builtin_elementwise_fma(f2{S2, S2}, t4, f2{S1, S1} * t3);
    f2 s2 = __builtin_elementwise_fma(f2{-S1, -S1}, t4, f2{S2, S2} * t3);
    a0 = a0 + t1 + t2;
    a1 = add_mi(m1, s1); a4 = add_pi(m1, s1); a2 = add_mi(m2, s2); a3 = add_pi(m2, s2);
}
__device__ __forceinline__ void dft20(const f2 (&in)[20], f2 (&out)[20]) {
    f2 t[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) t[i] = in[i];
#pragma unroll
    for (int b = 0; b < 5; ++b) dft4(t[(4 * b) % 20], t[(5 + 4 * b) % 20], t[(10 + 4 * b) % 20], t[(15 + 4 * b) % 20]);
#pragma unroll
    for (int c = 0; c < 4; ++c) dft5(t[(5 * c) % 20], t[(5 * c + 4) % 20], t[(5 * c + 8) % 20], t[(5 * c + 12) % 20], t[(5 * c + 16) % 20]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 5; ++d) out[(5 * c + 16 * d) % 20] = t[(5 * c + 4 * d) % 20];
}
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 v[20], y[20];
    for (int i = 0; i < 20; ++i) v[i] = f2{threadIdx.x * 1e-3f + i, 0.5f * i - threadIdx.x * 1e-4f};
    for (int it = 0; it < iters; ++it) {
        dft20(v, y);
#pragma unroll
        for (int i = 0; i < 20; ++i) v[i] = y[i] * 0.05f;
    }
    float s = 0;
    for (int i = 0; i < 20; ++i) s += v[i].x + v[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}
__global__ void chk(float* out) {
    f2 v[20], y[20];
    for (int i = 0; i < 20; ++i) v[i] = f2{(float)((i * 7 + 3) % 11) - 5.f, (float)((i * 5 + 1) % 13) - 6.f};
    dft20(v, y);
    for (int i = 0; i < 20; ++i) { out[2 * i] = y[i].x; out[2 * i + 1] = y[i].y; }
}
int main() {
    {
        float* d; (void)hipMalloc(&d, 160); hipLaunchKernelGGL(chk, dim3(1), dim3(1), 0, 0, d);
        float h[40]; (void)hipMemcpy(h, d, 160, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int k = 0; k < 20; ++k) {
            double re = 0, im = 0;
            for (int n = 0; n < 20; ++n) {
                double xr = (double)((n * 7 + 3) % 11) - 5., xi = (double)((n * 5 + 1) % 13) - 6.;
                double a = -2 * 3.14159265358979323846 * n * k / 20.; double c = cos(a), sn = sin(a);
                re += xr * c - xi * sn; im += xr * sn + xi * c;
            }
            maxerr = fmax(maxerr, fmax(fabs(re - h[2 * k]), fabs(im - h[2 * k + 1])));
        }
        printf("dft20 packed max err %.3g\n", maxerr);
    }
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
        printf("pk: waves/SIMD=%d  %.3f ms -> %.0f ns per (dft20 + 40 mul), %.2f ns per wave per SIMD\n", wps, ms, ms * 1e6 / iters, ms * 1e6 / iters / wps);
    }
    return 0;
}
