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

// as fill_e<4> but only the whole 128-byte lines of each row segment are written (head / tail partial lines skipped):
// tells whether the partial lines at the segment ends (row pitch 12000 B is not a multiple of 128) cost the rate
__global__ void fill_g(float* out, const int* pad_frame, int M, int T, int align_bytes) {
    const int clip = blockIdx.y, mel0 = blockIdx.x * 4;
    const int pf = pad_frame[clip];
    for (int r = 0; r < 4; ++r) {
        if (mel0 + r >= M) break;
        const size_t b0 = (((size_t)clip * M + mel0 + r) * T + pf) * 4, b1 = (((size_t)clip * M + mel0 + r) * T + T) * 4;
        const size_t a0 = (b0 + align_bytes - 1) / align_bytes * align_bytes, a1 = b1 / align_bytes * align_bytes;
        f32x4* o = reinterpret_cast<f32x4*>(reinterpret_cast<char*>(out) + a0);
        const int quads = (int)((a1 - a0) >> 4);
        for (int q = threadIdx.x; q < quads; q += blockDim.x) o[q] = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
    }
}

// the whole segment of every row, lanes mapped to ABSOLUTE 16-byte slots so that every 8 lanes write one 128-byte
// line (head / tail lines masked): rows x virtual quads as one index space, as whisper_clamp_fill_kernel walks it
__global__ __launch_bounds__(256) void fill_h(float* out, const int* pad_frame, int M, int T) {
    const int clip = blockIdx.y, row0 = blockIdx.x, step = gridDim.x;
    const int pf = pad_frame[clip];
    const int rows = (M - row0 + step - 1) / step;
    const int vq = (((T - pf) * 4 + 127) / 128 + 1) * 8;
    const int n = rows * vq;
    int r = threadIdx.x / vq, v = threadIdx.x - r * vq;
    const int dr = blockDim.x / vq, dv = blockDim.x - dr * vq;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out + (size_t)clip * M * T);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int mel = row0 + r * step;
        const uintptr_t start = base + ((size_t)mel * T + pf) * 4, end = base + ((size_t)mel * T + T) * 4;
        const uintptr_t a = (start & ~(uintptr_t)127) + 16 * (uintptr_t)v;
        if (a >= start && a < end) *reinterpret_cast<f32x4*>(a) = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
        r += dr; v += dv;
        if (v >= vq) { v -= vq; ++r; }
    }
}

// as fill_h, but the lanes of a head / tail line that lie OUTSIDE the segment read what is there and write it back,
// so that every line reaches memory as one whole-line write (no partial-line read-modify-write at the DRAM side)
__global__ __launch_bounds__(256) void fill_i(float* out, const int* pad_frame, int M, int T, size_t total_bytes) {
    const int clip = blockIdx.y, row0 = blockIdx.x, step = gridDim.x;
    const int pf = pad_frame[clip];
    const int rows = (M - row0 + step - 1) / step;
    const int vq = (((T - pf) * 4 + 127) / 128 + 1) * 8;
    const int n = rows * vq;
    int r = threadIdx.x / vq, v = threadIdx.x - r * vq;
    const int dr = blockDim.x / vq, dv = blockDim.x - dr * vq;
    const uintptr_t buf0 = reinterpret_cast<uintptr_t>(out), buf1 = buf0 + total_bytes;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out + (size_t)clip * M * T);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int mel = row0 + r * step;
        const uintptr_t start = base + ((size_t)mel * T + pf) * 4, end = base + ((size_t)mel * T + T) * 4;
        const uintptr_t a = (start & ~(uintptr_t)127) + 16 * (uintptr_t)v;
        const uintptr_t line_end = (end + 127) & ~(uintptr_t)127;
        if (a < line_end && a >= buf0 && a + 16 <= buf1) {
            f32x4 val = {-0.6f, -0.6f, -0.6f, -0.6f};
            if (a < start || a >= end) val = *reinterpret_cast<const f32x4*>(a);
            *reinterpret_cast<f32x4*>(a) = val;
        }
        r += dr; v += dv;
        if (v >= vq) { v -= vq; ++r; }
    }
}

// as fill_i in two passes: (1) the head and tail lines of all of the workgroup's rows (16 lanes per row: read what lies
// outside the segment, write the whole line), (2) the whole lines in between, no loads
__global__ __launch_bounds__(256) void fill_j(float* out, const int* pad_frame, int M, int T, size_t total_bytes) {
    const int clip = blockIdx.y, row0 = blockIdx.x, step = gridDim.x;
    const int pf = pad_frame[clip];
    const int rows = (M - row0 + step - 1) / step;
    const uintptr_t buf0 = reinterpret_cast<uintptr_t>(out), buf1 = buf0 + total_bytes;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out + (size_t)clip * M * T);
    for (int i = threadIdx.x; i < rows * 16; i += blockDim.x) {
        const int r = i >> 4, l = i & 15, mel = row0 + r * step;
        const uintptr_t start = base + ((size_t)mel * T + pf) * 4, end = base + ((size_t)mel * T + T) * 4;
        const uintptr_t line = l < 8 ? (start & ~(uintptr_t)127) : (end & ~(uintptr_t)127);
        const uintptr_t a = line + 16 * (uintptr_t)(l & 7);
        const bool aligned_end = (l >= 8) && ((end & 127) == 0);           // no tail line
        const bool same = (l >= 8) && ((start & ~(uintptr_t)127) == (end & ~(uintptr_t)127));
        if (!aligned_end && !same && a >= buf0 && a + 16 <= buf1) {
            f32x4 val = {-0.6f, -0.6f, -0.6f, -0.6f};
            if (a < start || a >= end) val = *reinterpret_cast<const f32x4*>(a);
            *reinterpret_cast<f32x4*>(a) = val;
        }
    }
    const int vq = (((T - pf) * 4) / 128) * 8;                           // upper bound of the whole lines of a row, in quads
    const int n = rows * vq;
    int r = threadIdx.x / vq, v = threadIdx.x - r * vq;
    const int dr = blockDim.x / vq, dv = blockDim.x - dr * vq;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int mel = row0 + r * step;
        const uintptr_t start = base + ((size_t)mel * T + pf) * 4, end = base + ((size_t)mel * T + T) * 4;
        const uintptr_t a = ((start + 127) & ~(uintptr_t)127) + 16 * (uintptr_t)v;
        if (a + 16 <= (end & ~(uintptr_t)127)) *reinterpret_cast<f32x4*>(a) = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
        r += dr; v += dv;
        if (v >= vq) { v -= vq; ++r; }
    }
}

// Flat line runs: the clip's fill lines -- per row the 128-byte lines that cover its padding
// segment, rows in order -- are one sequence; every WAVE takes RUN consecutive lines of it, 8 lines (1 KB) per store
// instruction, across row boundaries.  keep = 0: lanes outside the segment are masked (partial head / tail lines);
// keep = 1: they read what is there and write it back (whole-line writes); keep = 2: head / tail lines left out.
template <int RUN>
__global__ __launch_bounds__(256) void fill_k(float* out, const int* pad_frame, int M, int T, int keep, size_t total_bytes) {
    const int clip = blockIdx.y;
    const int pf = pad_frame[clip];
    const int L = ((T - pf) * 4 + 127) / 128 + 1;                          // lines per row (upper bound of the cover)
    const int n_lines = M * L;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uintptr_t buf0 = reinterpret_cast<uintptr_t>(out), buf1 = buf0 + total_bytes;
    const uintptr_t base = reinterpret_cast<uintptr_t>(out + (size_t)clip * M * T);
    const int l0 = wave * RUN;
#pragma unroll 4
    for (int l = l0 + (lane >> 3); l < min(l0 + RUN, n_lines); l += 8) {
        const int mel = l / L, k = l - mel * L;
        const uintptr_t start = base + ((size_t)mel * T + pf) * 4, end = base + ((size_t)mel * T + T) * 4;
        const uintptr_t a = (start & ~(uintptr_t)127) + 128 * (uintptr_t)k + 16 * (uintptr_t)(lane & 7);
        const uintptr_t line = a & ~(uintptr_t)127;
        if (line >= end) continue;
        const bool inside = a >= start && a < end;
        if (keep == 2) {
            if (line >= start && line + 128 <= end) *reinterpret_cast<f32x4*>(a) = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
        } else if (inside) {
            *reinterpret_cast<f32x4*>(a) = f32x4{-0.6f, -0.6f, -0.6f, -0.6f};
        } else if (keep == 1 && a >= buf0 && a + 16 <= buf1) {
            const f32x4 val = *reinterpret_cast<const f32x4*>(a);
            *reinterpret_cast<f32x4*>(a) = val;
        }
    }
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
    for (int al : {16, 64, 128, 256, 1024}) {
        char nm[64]; snprintf(nm, 64, "G R=4 tpb=256 whole %d-byte blocks only", al);
        time(nm, [&] { hipLaunchKernelGGL(fill_g, dim3(M / 4, B), dim3(256), 0, 0, out, pf, M, T, al); });
    }
    for (int split : {16, 32, 64}) {
        char nm[64]; snprintf(nm, 64, "H split=%d line-aligned lanes, whole segment", split);
        time(nm, [&] { hipLaunchKernelGGL(fill_h, dim3(split, B), dim3(256), 0, 0, out, pf, M, T); });
    }
    for (int split : {16, 32}) {
        char nm[64]; snprintf(nm, 64, "I split=%d whole-line writes (read + rewrite)", split);
        time(nm, [&] { hipLaunchKernelGGL(fill_i, dim3(split, B), dim3(256), 0, 0, out, pf, M, T, (size_t)B * M * T * 4); });
    }
    for (int split : {16, 32}) {
        char nm[64]; snprintf(nm, 64, "J split=%d boundary lines first, then whole lines", split);
        time(nm, [&] { hipLaunchKernelGGL(fill_j, dim3(split, B), dim3(256), 0, 0, out, pf, M, T, (size_t)B * M * T * 4); });
    }
    {
        int min_pf = 3000;
        for (int v : h) min_pf = v < min_pf ? v : min_pf;
        const int max_lines = M * (((T - min_pf) * 4 + 127) / 128 + 1);
        for (int keep : {0, 1, 2}) {
            char nm[80];
            snprintf(nm, 80, "K run=32 lines per wave keep=%d", keep);
            time(nm, [&] { hipLaunchKernelGGL(fill_k<32>, dim3((max_lines + 127) / 128, B), dim3(256), 0, 0, out, pf, M, T, keep, (size_t)B * M * T * 4); });
            snprintf(nm, 80, "K run=80 lines per wave keep=%d", keep);
            time(nm, [&] { hipLaunchKernelGGL(fill_k<80>, dim3((max_lines + 319) / 320, B), dim3(256), 0, 0, out, pf, M, T, keep, (size_t)B * M * T * 4); });
            snprintf(nm, 80, "K run=320 lines per wave keep=%d", keep);
            time(nm, [&] { hipLaunchKernelGGL(fill_k<320>, dim3((max_lines + 1279) / 1280, B), dim3(256), 0, 0, out, pf, M, T, keep, (size_t)B * M * T * 4); });
        }
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
