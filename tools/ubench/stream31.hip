// Microbenchmark: streaming floor of the 3 : 1 decimator's traffic (read 3 n floats, write n floats, n = 41 M as on
// config 3) by access shape, no filter arithmetic, no LDS:
//   A  one 1024-output tile per workgroup, three contiguous 16-byte loads per lane, one 16-byte store
//   B  the same with twelve 4-byte loads per lane (256 contiguous bytes per wave instruction, as the decimator's staging)
//   C  as B, eight consecutive tiles per workgroup, the next tile's loads issued before the current tile's store
//   D  as A, eight consecutive tiles per workgroup (same prefetch)
// hipcc --offload-arch=gfx950 -O3 -o stream31 stream31.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_a(const f32x4* __restrict__ x, f32x4* __restrict__ y, size_t n4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const size_t wbase = (t & ~(size_t)63) * 3, lane = t & 63;
    const f32x4 a = x[wbase + lane], b = x[wbase + 64 + lane], c = x[wbase + 128 + lane];
    y[t] = a + b + c;
}

__global__ __launch_bounds__(256) void k_b(const float* __restrict__ x, f32x4* __restrict__ y, size_t n4) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    const float* src = x + (size_t)blockIdx.x * 256 * 12 + threadIdx.x;
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = src[256 * i];
    y[t] = f32x4{v[0] + v[4] + v[8], v[1] + v[5] + v[9], v[2] + v[6] + v[10], v[3] + v[7] + v[11]};
}

template <int TILES>
__global__ __launch_bounds__(256) void k_c(const float* __restrict__ x, f32x4* __restrict__ y, size_t n4) {
    size_t tile = (size_t)blockIdx.x * TILES;
    float v[12], w[12];
    const size_t n_tiles = n4 / 256;
    if (tile >= n_tiles) return;
    {
        const float* src = x + tile * 256 * 12 + threadIdx.x;
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = src[256 * i];
    }
    for (int k = 0; k < TILES && tile < n_tiles; ++k, ++tile) {
        const bool more = k + 1 < TILES && tile + 1 < n_tiles;
        if (more) {
            const float* src = x + (tile + 1) * 256 * 12 + threadIdx.x;
#pragma unroll
            for (int i = 0; i < 12; ++i) w[i] = src[256 * i];
        }
        y[tile * 256 + threadIdx.x] = f32x4{v[0] + v[4] + v[8], v[1] + v[5] + v[9], v[2] + v[6] + v[10], v[3] + v[7] + v[11]};
        if (more) {
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] = w[i];
        }
    }
}

template <int TILES>
__global__ __launch_bounds__(256) void k_d(const f32x4* __restrict__ x, f32x4* __restrict__ y, size_t n4) {
    size_t tile = (size_t)blockIdx.x * TILES;
    const size_t n_tiles = n4 / 256;
    if (tile >= n_tiles) return;
    const size_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x4 a, b, c, a2, b2, c2;
    {
        const f32x4* src = x + (tile * 256 + wv * 64) * 3 + lane;
        a = src[0]; b = src[64]; c = src[128];
    }
    for (int k = 0; k < TILES && tile < n_tiles; ++k, ++tile) {
        const bool more = k + 1 < TILES && tile + 1 < n_tiles;
        if (more) {
            const f32x4* src = x + ((tile + 1) * 256 + wv * 64) * 3 + lane;
            a2 = src[0]; b2 = src[64]; c2 = src[128];
        }
        y[tile * 256 + threadIdx.x] = a + b + c;
        if (more) { a = a2; b = b2; c = c2; }
    }
}

int main() {
    const size_t n4 = (size_t)40960 * 256;                    // 41.9 M outputs
    float *x, *y, *dirty;
    const size_t dirty_bytes = (size_t)512 * 80 * 3000 * 4;
    hipMalloc(&x, n4 * 48); hipMalloc(&y, n4 * 16); hipMalloc(&dirty, dirty_bytes);
    hipMemset(x, 0, n4 * 48);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)n4 * 64;
    auto time = [&](const char* name, auto&& launch) {
        for (int ctx = 0; ctx < 2; ++ctx) {
            float total = 0;
            for (int i = 0; i < 25; ++i) {
                if (ctx) hipMemsetAsync(dirty, 1, dirty_bytes, 0); else launch();
                hipEventRecord(e0);
                launch();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (i >= 5) total += ms;
            }
            printf("%-62s %-28s %.4f ms  %.2f TB/s\n", name, ctx ? "behind a 491 MB memset" : "behind itself", total / 20, bytes / (total / 20 * 1e-3) / 1e12);
        }
    };
    const int g = (int)(n4 / 256);
    time("A 3 x 16-byte loads, 1 tile per workgroup", [&] { hipLaunchKernelGGL(k_a, dim3(g), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, n4); });
    time("B 12 x 4-byte loads, 1 tile per workgroup", [&] { hipLaunchKernelGGL(k_b, dim3(g), dim3(256), 0, 0, x, (f32x4*)y, n4); });
    time("C 12 x 4-byte loads, 8 tiles per workgroup, prefetch", [&] { hipLaunchKernelGGL(k_c<8>, dim3(g / 8), dim3(256), 0, 0, x, (f32x4*)y, n4); });
    time("C 12 x 4-byte loads, 2 tiles per workgroup, prefetch", [&] { hipLaunchKernelGGL(k_c<2>, dim3(g / 2), dim3(256), 0, 0, x, (f32x4*)y, n4); });
    time("D 3 x 16-byte loads, 8 tiles per workgroup, prefetch", [&] { hipLaunchKernelGGL(k_d<8>, dim3(g / 8), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, n4); });
    time("D 3 x 16-byte loads, 2 tiles per workgroup, prefetch", [&] { hipLaunchKernelGGL(k_d<2>, dim3(g / 2), dim3(256), 0, 0, (const f32x4*)x, (f32x4*)y, n4); });
    return 0;
}
