// Microbenchmark: how fast can the padding region of a (B, M, 3000) float32 tensor be written?
// Patterns: A = rows x quads per workgroup as whisper_clamp_fill_kernel does (grid (split, B)), with / without a
// dependent-load prologue; B = persistent grid striding over (clip, row) pairs.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill_a(float* out, const int* pad_frame, const unsigned* keys, int M, int T, int prologue) {
    const int clip = blockIdx.y, row0 = blockIdx.x, step = gridDim.x;
    int pf = pad_frame[clip];
    float c = -0.6f;
    if (prologue) {
        __shared__ unsigned long long s;
        unsigned k = keys[clip * 64 + (threadIdx.x & 63)];
        unsigned long long b = __ballot(k > 5u);
        if (threadIdx.x == 0) s = b;
        __syncthreads();
        c += (float)(s & 1ull);
    }
    const int quads = (T - pf) >> 2, rows = (M - row0 + step - 1) / step, n = rows * quads;
    int r = threadIdx.x / quads, q = threadIdx.x - r * quads;
    const int dr = blockDim.x / quads, dq = blockDim.x - dr * quads;
    float* o = out + (size_t)clip * M * T;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int mel = row0 + r * step;
        reinterpret_cast<f32x4*>(o + (size_t)mel * T + pf)[q] = f32x4{c, c, c, c};
        r += dr; q += dq;
        if (q >= quads) { q -= quads; ++r; }
    }
}

// persistent: workgroup w takes (clip, row) pairs w, w + G, ...; a row segment is written by the whole workgroup
__global__ __launch_bounds__(256) void fill_b(float* out, const int* pad_frame, int B, int M, int T, int rows_per_item) {
    const int items = B * ((M + rows_per_item - 1) / rows_per_item), per = (M + rows_per_item - 1) / rows_per_item;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        const int clip = it / per, r0 = (it - clip * per) * rows_per_item;
        const int pf = pad_frame[clip], quads = (T - pf) >> 2;
        float* o = out + (size_t)clip * M * T;
        const int n = min(rows_per_item, M - r0) * quads;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int r = i / quads, q = i - r * quads;
            reinterpret_cast<f32x4*>(o + (size_t)(r0 + r) * T + pf)[q] = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
        }
    }
}

// one workgroup per (clip, row); wave stores aligned to 1 KB (masked at the segment ends)
__global__ __launch_bounds__(256) void fill_d(float* out, const int* pad_frame, int M, int T) {
    const int clip = blockIdx.y, mel = blockIdx.x;
    const int pf = pad_frame[clip];
    const size_t a0 = ((size_t)clip * M + mel) * T + pf, a1 = ((size_t)clip * M + mel) * T + T;   // float indices
    const size_t q0 = a0 >> 2, q1 = a1 >> 2;                  // float4 indices (a0, a1 multiples of 4)
    f32x4* o = reinterpret_cast<f32x4*>(out);
    for (size_t q = (q0 & ~(size_t)63) + threadIdx.x; q < q1; q += 256)
        if (q >= q0) o[q] = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
}

// R consecutive rows per workgroup, TPB threads, rows written one after the other
template <int R>
__global__ void fill_e(float* out, const int* pad_frame, int M, int T) {
    const int clip = blockIdx.y, mel0 = blockIdx.x * R;
    const int pf = pad_frame[clip];
    const int quads = (T - pf) >> 2;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (mel0 + r >= M) break;
        f32x4* o = reinterpret_cast<f32x4*>(out + ((size_t)clip * M + mel0 + r) * T + pf);
        for (int q = threadIdx.x; q < quads; q += blockDim.x) o[q] = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
    }
}

// contiguous fills: U float4 per thread per trip, wave-contiguous within each of the U sub-blocks; NT = nontemporal
template <int U, bool NT>
__global__ __launch_bounds__(256) void fill_c(f32x4* out, size_t n4) {
    const f32x4 v = {-0.6f, -0.6f, -0.6f, -0.6f};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = base + (size_t)u * 256;
            if (i < n4) { if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v; }
        }
    }
}

int main(int argc, char** argv) {
    const int B = 1024, M = 128, T = 3000;
    const int fixed_pf = argc > 1 ? atoi(argv[1]) : -1;
    float* out; int* pf; unsigned* keys;
    hipMalloc(&out, (size_t)B * M * T * 4); hipMalloc(&pf, B * 4); hipMalloc(&keys, B * 64 * 4);
    hipMemset(keys, 0, B * 64 * 4);
    std::vector<int> h(B); size_t bytes = 0;
    for (int i = 0; i < B; ++i) { h[i] = fixed_pf >= 0 ? fixed_pf : 48 * (6 + (i * 7) % 12); bytes += (size_t)M * (T - h[i]) * 4; }
    hipMemcpy(pf, h.data(), B * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto&& launch) {
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.4f ms  %.2f TB/s\n", name, ms / 20, bytes / (ms / 20 * 1e-3) / 1e12);
    };
    for (int split : {32, 128}) for (int pro : {0, 1}) {
        char nm[64]; snprintf(nm, 64, "A split=%d prologue=%d", split, pro);
        time(nm, [&] { hipLaunchKernelGGL(fill_a, dim3(split, B), dim3(256), 0, 0, out, pf, keys, M, T, pro); });
    }
    for (int g : {512, 1024, 2048, 4096}) for (int rpi : {1, 4}) {
        char nm[64]; snprintf(nm, 64, "B grid=%d rows/item=%d", g, rpi);
        time(nm, [&] { hipLaunchKernelGGL(fill_b, dim3(g), dim3(256), 0, 0, out, pf, B, M, T, rpi); });
    }
    time("D one WG per row, 1 KB-aligned stores", [&] { hipLaunchKernelGGL(fill_d, dim3(M, B), dim3(256), 0, 0, out, pf, M, T); });
    for (int tpb : {128, 256, 512, 1024}) {
        char nm[64];
        snprintf(nm, 64, "E R=1 tpb=%d", tpb); time(nm, [&] { hipLaunchKernelGGL(fill_e<1>, dim3(M, B), dim3(tpb), 0, 0, out, pf, M, T); });
        snprintf(nm, 64, "E R=2 tpb=%d", tpb); time(nm, [&] { hipLaunchKernelGGL(fill_e<2>, dim3(M / 2, B), dim3(tpb), 0, 0, out, pf, M, T); });
        snprintf(nm, 64, "E R=4 tpb=%d", tpb); time(nm, [&] { hipLaunchKernelGGL(fill_e<4>, dim3(M / 4, B), dim3(tpb), 0, 0, out, pf, M, T); });
        snprintf(nm, 64, "E R=8 tpb=%d", tpb); time(nm, [&] { hipLaunchKernelGGL(fill_e<8>, dim3(M / 8, B), dim3(tpb), 0, 0, out, pf, M, T); });
    }
    const size_t n4 = bytes / 16;
    for (int g : {1024, 4096, 16384, 65536}) {
        char nm[64];
        snprintf(nm, 64, "C U=1 grid=%d", g); time(nm, [&] { hipLaunchKernelGGL((fill_c<1, false>), dim3(g), dim3(256), 0, 0, (f32x4*)out, n4); });
        snprintf(nm, 64, "C U=4 grid=%d", g); time(nm, [&] { hipLaunchKernelGGL((fill_c<4, false>), dim3(g), dim3(256), 0, 0, (f32x4*)out, n4); });
        snprintf(nm, 64, "C U=4 NT grid=%d", g); time(nm, [&] { hipLaunchKernelGGL((fill_c<4, true>), dim3(g), dim3(256), 0, 0, (f32x4*)out, n4); });
    }
    time("hipMemsetAsync contiguous same bytes", [&] { hipMemsetAsync(out, 0, bytes, 0); });
    return 0;
}
