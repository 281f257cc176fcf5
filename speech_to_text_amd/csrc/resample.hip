// Waveform preprocessing for gfx950 (MI355X): mono mix, polyphase windowed-sinc resampling, per-clip
// peak, peak normalisation and silence trim.
//
// Replaces AudioPreprocessor.preprocess (reference src/data.py:54-108; same steps in
// inference.py:100-132): torch.mean over channels (:66-67), torchaudio.transforms.Resample (:70-72),
// x / (max|x| + 1e-8) (:78-81), trim to the first..last sample above 0.01 * max (:84-91).
//
// Kernel 1a (decimate_peak<ORIG>): integer decimation ORIG:1 (48 -> 16 kHz: ORIG = 3, 41 taps;
//   32 -> 16 kHz: ORIG = 2, 28 taps).  A 256-thread workgroup takes one (clip, run of 8 tiles) item; with
//   STTFE_DECIMATE_PERSIST=1 resident workgroups walk the items instead, clip-major over an LDS prefix of the
//   clips' run counts (up to 2048 clips).  Per 1024-output tile the input span
//   is read from HBM once with lane-contiguous loads and de-interleaved into ORIG phase rows in LDS
//   (row r holds padded[ORIG m + r]), so that out[i] = sum_r sum_j h[ORIG j + r] * row_r[i + j] and a
//   thread that owns 4 consecutive outputs reads each row with five conflict-free ds_read_b128
//   (17 of 20 values used) instead of 41 scalar reads per output.  Taps are wave-uniform (SGPRs).
//   Output: one global_store_dwordx4 per thread; per-clip max|out| with one atomicMax per wave.
// Kernel 1b (resample_generic_quad / resample_generic_lds): any ORIG:NEW ratio (44.1 / 22.05 / 8 kHz sources, speed
//   perturbation): compact per-phase taps and the source span of a tile in LDS, the next tile's samples requested a
//   tile ahead; four consecutive outputs per thread off one sample window when NEW % 4 == 0.  resample_generic
//   (one thread per output, everything from global memory) remains for ratios whose taps do not fit 64 KB.
// Kernel 1c (copy_peak): same-rate input: mono mix / copy / peak only.
// Kernel 2 (normalize_trim): y = x / (peak + 1e-8) in place (IEEE division, bit-identical to the
//   float32 NumPy / torch result) and first / last index with |y| > thr via atomicMin / atomicMax.
// Kernel 3 (trim_finalize): offsets[b] += first, lengths[b] = last + 1 - first.  Trimming moves no
//   samples: every later kernel addresses a clip through the device offsets / lengths arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sttfe.h"
#include "tables.h"
#include "timing.h"

namespace sttfe {

namespace rs {
constexpr int kThreads = 256;
constexpr int kOutPerThread = 4;
constexpr int kTile = kThreads * kOutPerThread;       // 1024 outputs per tile
constexpr int kTapsPerPhase = 14;                     // ceil((2 width + ORIG) / ORIG) for every ORIG:1 ratio
constexpr int kRowLen = kTile + 16;                   // m range a tile reads: 4 * 255 + 20
constexpr int kRowStride = 1068;                      // words; % 4 == 0 (b128 alignment), % 32 == 12
constexpr int kMaxPlanClips = 2048;                   // clips whose run-count prefix a persistent workgroup keeps in LDS
static_assert(kRowStride >= kRowLen && kRowStride % 4 == 0, "phase row stride");
}  // namespace rs
constexpr int kTrimSlots = 64;                        // per-clip candidate slots of the trim bounds (normalize_trim_kernel)

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef STTFE_STAMPS
// Diagnostic build only (tools/stamps_decimate.py): per-wave cycle sums of the phases of the decimator's tile loop.
__device__ unsigned long long* g_rs_stamp_buf = nullptr;
#define RS_STAMP(i)                                                              \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        unsigned long long t_;                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                       \
        stamp_sum[i] += t_ - stamp_prev;                                         \
        stamp_prev = t_;                                                         \
    } while (0)
#else
#define RS_STAMP(i)
#endif

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// Sample i of the mono mix of a (channels, n) planar clip (src/data.py:66-67: torch.mean(dim=0)).
// Input samples are float32, or int16 as decoders deliver them: x / 32768 (what torchaudio.load's normalisation
// does, reference src/data.py:313; exact in float32), so that the int16 path halves the bytes crossing PCIe and
// HBM without a conversion pass.
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(int16_t v) { return (float)v * (1.0f / 32768.0f); }

template <typename TIn>
__device__ __forceinline__ float mono_at(const TIn* __restrict__ x, int n, int channels, int i) {
    if (channels == 1) return to_f32(x[i]);
    if (channels == 2)      // stereo: two independent loads (the loop below waits for each load before the next)
        return (0.f + to_f32(x[i]) + to_f32(x[(size_t)n + i])) / 2.0f;
    float s = 0.f;
    for (int c = 0; c < channels; ++c) s += to_f32(x[(size_t)c * n + i]);
    return s / (float)channels;
}

template <int ORIG, typename TIn>
__global__ __launch_bounds__(rs::kThreads) void decimate_peak_kernel(
    const TIn* __restrict__ pcm_in, const int64_t* __restrict__ in_off, const int32_t* __restrict__ in_len,
    int channels, const ResamplePlanHeader* __restrict__ plan, float* __restrict__ pcm_out,
    const int64_t* __restrict__ out_off, int32_t* __restrict__ out_len, unsigned* __restrict__ peak_keys,
    int tiles_per_block, int batch, int persistent) {
    using namespace rs;
    __shared__ __attribute__((aligned(16))) float rows[ORIG * kRowStride];
    __shared__ int s_pref[kMaxPlanClips + 1];
    __shared__ int s_wave_sum[kThreads / 64];
    __shared__ float s_peak[kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef STTFE_STAMPS
    unsigned long long stamp_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
    const unsigned long long stamp_entry = stamp_prev;
#endif
    const float* __restrict__ h = reinterpret_cast<const float*>(plan + 1) + 2;      // taps of phase 0 (new_rate == 1 here)
    const int width = plan->width;
    auto tiles_of = [&](int n) { return ((n + ORIG - 1) / ORIG + kTile - 1) / kTile; };
    auto runs_of = [&](int n) { return (tiles_of(n) + tiles_per_block - 1) / tiles_per_block; };

    // Items are (clip, run of tiles_per_block tiles).  persistent: the grid is the resident workgroups, every one of
    // them builds the exclusive prefix of the clips' run counts in LDS (items are clip-major over it, none is empty)
    // and takes items blockIdx.x, blockIdx.x + gridDim.x, ...: lengths, offsets, the plan header, the taps and the
    // first tile's round trip are paid once per workgroup instead of once per four tiles (with one item per workgroup
    // 1 / 2 / 4 tiles per item took 0.47 / 0.25 / 0.185 ms on config 3: the start of a workgroup cost more than its
    // tiles).  Otherwise (more than kMaxPlanClips clips): one item per workgroup, clip = blockIdx.y, run = blockIdx.x.
    int n_items = 0, item = 0, item_step = 0;
    if (persistent) {
        constexpr int kPer = kMaxPlanClips / kThreads;
        const int per = (batch + kThreads - 1) / kThreads;
        int before[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int c = tid * per + k;
            before[k] = sum;
            if (k < per && c < batch) {
                const int n = in_len[c];
                if (blockIdx.x == 0) out_len[c] = (n + ORIG - 1) / ORIG;        // ceil(1 * n / ORIG)
                sum += runs_of(n);
            }
        }
        int incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wave_sum[wave] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wave; ++w) base += s_wave_sum[w];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int c = tid * per + k;
            if (k < per && c < batch) s_pref[c] = base + before[k];
        }
        if (tid == kThreads - 1) s_pref[batch] = base + sum;
        __syncthreads();
        n_items = s_pref[batch];
        item = blockIdx.x;
        item_step = gridDim.x;
        if (item >= n_items) return;
    }
    // item -> (clip, run): largest c with s_pref[c] <= it, two 64-way ballot steps (clips without samples have no run:
    // the prefix repeats and the LAST clip of a repeat is the one that owns the item)
    auto decode = [&](int it, int& clip_out, int& run_out) {
        const int stride = (batch + 63) >> 6;
        const int c1 = lane * stride;
        const bool le1 = c1 < batch && s_pref[min(c1, batch)] <= it;
        const int block = __popcll(__ballot(le1)) - 1;
        const int c2 = block * stride + lane;
        const bool le2 = lane < stride && c2 < batch && s_pref[min(c2, batch)] <= it;
        const int c = block * stride + __popcll(__ballot(le2)) - 1;
        clip_out = __builtin_amdgcn_readfirstlane(c);
        run_out = __builtin_amdgcn_readfirstlane(it - s_pref[c]);
    };
    int clip, run;
    if (persistent) {
        decode(item, clip, run);
    } else {
        clip = blockIdx.y;
        run = blockIdx.x;
        if (run == 0 && tid == 0) out_len[clip] = (in_len[clip] + ORIG - 1) / ORIG;
    }
    int n = in_len[clip];
    const TIn* __restrict__ x = pcm_in + in_off[clip];
    float* __restrict__ y = pcm_out + out_off[clip];
    if (!persistent && run * tiles_per_block >= tiles_of(n)) return;

    // padded[ORIG * o0 + g] = x[ORIG * o0 + g - width] (zero outside the clip) -> rows[g % ORIG][g / ORIG].
    // All loads of a tile are issued together (one HBM round trip per tile, not one per element: the compiler cannot
    // move a load above an LDS store of the loop before it), and a tile ahead -- across items too: the next tile's
    // samples are requested right after this tile's have gone to LDS and arrive under its 168 FMAs per thread.
    // int16 mono: one aligned 32-bit load brings two samples (single 2-byte loads ran at half the rate of the float
    // path).  Pairs are aligned in source-index space (clip starts are 4-byte aligned).
    constexpr bool kIs16 = sizeof(TIn) == 2;
    constexpr int kPairs = (ORIG * kRowLen + 2 + 1) / 2;
    constexpr int kPairsPerThread = (kPairs + kThreads - 1) / kThreads;
    constexpr int kPerThread = (ORIG * kRowLen + kThreads - 1) / kThreads;
    uint32_t stage2[kIs16 ? kPairsPerThread : 1];
    uint32_t stage_left = 0;
    float stage[kPerThread];
    auto is_paired = [&](const TIn* xs) { return kIs16 && channels == 1 && (reinterpret_cast<uintptr_t>(xs) & 3) == 0; };
    // int16, whole span (and a sample either side) inside the clip: thread u takes the 4 ORIG consecutive samples
    // g = 4 ORIG u .., i.e. elements 4u .. 4u + 3 of every phase row, as 2 ORIG + 1 aligned 32-bit words (the span
    // starts on an odd sample in every other tile: the words are then shifted by 16 bits, wave-uniformly), and the
    // 16 ORIG samples behind the 256 chunks go one each to the first threads.  No g / ORIG, g % ORIG, no bounds and
    // ORIG 16-byte LDS writes per thread instead of 8 ORIG + 2 scalar ones: the staging arithmetic was what bound
    // the int16 path.
    constexpr int kChunk = 4 * ORIG;
    constexpr int kWords = kChunk / 2 + 1;
    constexpr int kLeft = ORIG * kRowLen - kChunk * kThreads;
    static_assert(!kIs16 || (kWords <= kPairsPerThread && kLeft >= 0 && kLeft <= kThreads), "vector staging fits the pair stage");
    auto interior16 = [&](const TIn* xs, int ns, int tile) {
        const int first = ORIG * tile * kTile - width;
        return kIs16 && is_paired(xs) && first >= 1 && first + ORIG * kRowLen + 1 <= ns;
    };
    struct u32x4a4 { uint32_t v[4]; } __attribute__((packed, aligned(4)));
    auto request = [&](const TIn* __restrict__ xs, int ns, int tile) {
        const int o0 = tile * kTile;
        if (kIs16 && interior16(xs, ns, tile)) {
            const int first = ORIG * o0 - width;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(xs + ((first + kChunk * tid) & ~1));
#pragma unroll
            for (int i = 0; i < (kIs16 ? kWords : 0); i += 4) {
                if (i + 4 <= kWords) {
                    u32x4a4 q;
                    __builtin_memcpy(&q, src + i, sizeof(q));
                    stage2[i] = q.v[0]; stage2[i + 1] = q.v[1]; stage2[i + 2] = q.v[2]; stage2[i + 3] = q.v[3];
                } else {
#pragma unroll
                    for (int k = i; k < kWords; ++k) stage2[k] = src[k];
                }
            }
            if (tid < kLeft) stage_left = (uint16_t)xs[first + kChunk * kThreads + tid];
        } else if (kIs16 && is_paired(xs)) {
            const int base = ORIG * o0 - width;                     // source index of g = 0
            const int e_base = base & ~1;                           // even (floor also for negative base)
#pragma unroll
            for (int i = 0; i < (kIs16 ? kPairsPerThread : 0); ++i) {
                const int e = e_base + 2 * (tid + i * kThreads);
                uint32_t w = 0;
                if (e >= 0 && e + 1 < ns) w = *reinterpret_cast<const uint32_t*>(xs + e);
                else if (e >= 0 && e < ns) w = (uint16_t)xs[e];
                else if (e + 1 >= 0 && e + 1 < ns) w = (uint32_t)(uint16_t)xs[e + 1] << 16;
                stage2[i] = w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i) {
                const int g = tid + i * kThreads;
                const int src = ORIG * o0 + g - width;
                stage[i] = (g < ORIG * kRowLen && src >= 0 && src < ns) ? mono_at(xs, ns, channels, src) : 0.f;
            }
        }
    };
    auto deposit = [&](const TIn* xs, int ns, int tile) {
        if (kIs16 && interior16(xs, ns, tile)) {
            const bool odd = (ORIG * tile * kTile - width) & 1;      // wave-uniform
            float f[kChunk];
#pragma unroll
            for (int i = 0; i < (kIs16 ? kChunk / 2 : 0); ++i) {
                const uint32_t w = odd ? __builtin_amdgcn_alignbit(stage2[i + 1], stage2[i], 16) : stage2[i];
                f[2 * i] = to_f32((int16_t)(w & 0xffffu));
                f[2 * i + 1] = to_f32((int16_t)(w >> 16));
            }
#pragma unroll
            for (int r = 0; r < ORIG; ++r)                          // sample q of the chunk: row q % ORIG, element 4u + q / ORIG
                *reinterpret_cast<f32x4*>(rows + r * kRowStride + 4 * tid) = f32x4{f[r], f[r + ORIG], f[r + 2 * ORIG], f[r + 3 * ORIG]};
            if (tid < kLeft) rows[(tid % ORIG) * kRowStride + 4 * kThreads + tid / ORIG] = to_f32((int16_t)stage_left);
        } else if (kIs16 && is_paired(xs)) {
            const int base = ORIG * tile * kTile - width;
            const int e_base = base & ~1;
#pragma unroll
            for (int i = 0; i < (kIs16 ? kPairsPerThread : 0); ++i) {
                const int g = e_base + 2 * (tid + i * kThreads) - base;
                const float lo = to_f32((int16_t)(stage2[i] & 0xffffu)), hi = to_f32((int16_t)(stage2[i] >> 16));
                if (g >= 0 && g < ORIG * kRowLen) rows[(g % ORIG) * kRowStride + g / ORIG] = lo;
                if (g + 1 >= 0 && g + 1 < ORIG * kRowLen) rows[((g + 1) % ORIG) * kRowStride + (g + 1) / ORIG] = hi;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i) {
                const int g = tid + i * kThreads;
                if (g < ORIG * kRowLen) rows[(g % ORIG) * kRowStride + g / ORIG] = stage[i];
            }
        }
    };
    // The item after the current one is held ready (clip, run, length, pointers: wave-uniform registers); the one
    // after that is looked up at the top of every item and its length / offsets are requested with VECTOR loads (a
    // lane offset the compiler cannot see through): scalar loads return out of order, so the first LDS read of the
    // tile loop would wait for them (lgkmcnt(0)) -- 4 us per item on config 3 -- whereas vector loads are waited for
    // in order, behind the tile's own samples, and have long arrived when the item becomes the next one.
    int lane_zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane_zero));
    int clip_nx = clip, run_nx = 0, n_nx = 0;
    const TIn* __restrict__ x_nx = x;
    float* __restrict__ y_nx = y;
    bool has_next = persistent && item + item_step < n_items;
    if (has_next) {
        decode(item + item_step, clip_nx, run_nx);
        n_nx = in_len[clip_nx];
        x_nx = pcm_in + in_off[clip_nx];
        y_nx = pcm_out + out_off[clip_nx];
    }
    request(x, n, run * tiles_per_block);
    RS_STAMP(7);
    while (true) {
        const int n_out = (n + ORIG - 1) / ORIG;
        const int tile_begin = run * tiles_per_block;
        const int tile_end = min(tile_begin + tiles_per_block, tiles_of(n));
        const int after_item = item + 2 * item_step;
        const bool has_after = has_next && after_item < n_items;
        int clip_nn = 0, run_nn = 0, v_n = 0;
        int64_t v_in = 0, v_out = 0;
        if (has_after) {
            decode(after_item, clip_nn, run_nn);
            v_n = in_len[clip_nn + lane_zero];
            v_in = in_off[clip_nn + lane_zero];
            v_out = out_off[clip_nn + lane_zero];
        }
        float peak = 0.f;
        for (int tile = tile_begin; tile < tile_end; ++tile) {
            const int o0 = tile * kTile;
            RS_STAMP(6);
            __syncthreads();                                        // previous tile's row reads are done
            RS_STAMP(5);
#ifdef STTFE_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            RS_STAMP(8);
#endif
            deposit(x, n, tile);
            RS_STAMP(0);
            __syncthreads();
            RS_STAMP(1);
            {
                const bool same = tile + 1 < tile_end;
                if (same || has_next) request(same ? x : x_nx, same ? n : n_nx, same ? tile + 1 : run_nx * tiles_per_block);
            }
            RS_STAMP(2);
            float acc[kOutPerThread] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < ORIG; ++r) {
                float v[20];
                const f32x4* src = reinterpret_cast<const f32x4*>(rows + r * kRowStride + kOutPerThread * tid);
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const f32x4 q = src[c];
                    v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w;
                }
#pragma unroll
                for (int j = 0; j < kTapsPerPhase; ++j) {
                    const float hj = h[ORIG * j + r];               // wave-uniform (scalar load)
#pragma unroll
                    for (int q = 0; q < kOutPerThread; ++q) acc[q] = fmaf(hj, v[q + j], acc[q]);
                }
            }
            RS_STAMP(3);
            const int o = o0 + kOutPerThread * tid;
            if (o + kOutPerThread <= n_out && ((reinterpret_cast<uintptr_t>(y + o) & 15) == 0)) {
                *reinterpret_cast<f32x4*>(y + o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
#pragma unroll
                for (int q = 0; q < kOutPerThread; ++q) peak = fmaxf(peak, fabsf(acc[q]));
            } else {
#pragma unroll
                for (int q = 0; q < kOutPerThread; ++q)
                    if (o + q < n_out) {
                        y[o + q] = acc[q];
                        peak = fmaxf(peak, fabsf(acc[q]));
                    }
            }
        }
        RS_STAMP(4);
        // one atomic per workgroup and item (items are clip-major: some 40 workgroups work on a clip at a time, and
        // device-scope read-modify-writes on one word queue behind each other)
        peak = wave_max(peak);
        if (lane == 0) s_peak[wave] = peak;
        __syncthreads();
        if (tid == 0) {
            float m = s_peak[0];
            for (int w = 1; w < kThreads / 64; ++w) m = fmaxf(m, s_peak[w]);
            if (m > 0.f) atomicMax(&peak_keys[clip], __float_as_uint(m));                   // peak >= 0: bit order == value order
        }
        if (!has_next) break;
        item += item_step;
        clip = clip_nx;
        run = run_nx;
        n = n_nx;
        x = x_nx;
        y = y_nx;
        has_next = has_after;
        if (has_after) {
            clip_nx = clip_nn;
            run_nx = run_nn;
            n_nx = __builtin_amdgcn_readfirstlane(v_n);
            const int64_t io = ((int64_t)__builtin_amdgcn_readfirstlane((int)(v_in >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)v_in);
            const int64_t oo = ((int64_t)__builtin_amdgcn_readfirstlane((int)(v_out >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)v_out);
            x_nx = pcm_in + io;
            y_nx = pcm_out + oo;
        }
    }
#ifdef STTFE_STAMPS
    if (g_rs_stamp_buf && lane == 0 && persistent) {
        RS_STAMP(6);
        stamp_sum[9] = stamp_prev - stamp_entry;
        unsigned long long* dst = g_rs_stamp_buf + ((size_t)blockIdx.x * (kThreads / 64) + wave) * 10;
        for (int i = 0; i < 10; ++i) dst[i] = stamp_sum[i];
    }
#endif
}

template <typename TIn>
__global__ __launch_bounds__(256) void resample_generic_kernel(
    const TIn* __restrict__ pcm_in, const int64_t* __restrict__ in_off, const int32_t* __restrict__ in_len,
    int channels, const ResamplePlanHeader* __restrict__ plan, float* __restrict__ pcm_out,
    const int64_t* __restrict__ out_off, int32_t* __restrict__ out_len, unsigned* __restrict__ peak_keys) {
    const int clip = blockIdx.y;
    const int n = in_len[clip];
    const int orig = plan->orig_rate, nw = plan->new_rate, width = plan->width, row = plan->taps_row;
    const int n_out = (int)(((int64_t)nw * n + orig - 1) / orig);
    if (blockIdx.x == 0 && threadIdx.x == 0) out_len[clip] = n_out;
    const int32_t* __restrict__ lo = reinterpret_cast<const int32_t*>(plan + 1);
    const int32_t* __restrict__ hi = lo + nw;
    const float* __restrict__ taps = reinterpret_cast<const float*>(hi + nw);
    const TIn* __restrict__ x = pcm_in + in_off[clip];
    float* __restrict__ y = pcm_out + out_off[clip];
    float peak = 0.f;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x) {
        const int i = o / nw, p = o - i * nw;
        const int base = i * orig - width;
        const float* __restrict__ hp = taps + (size_t)p * row;
        int k0 = lo[p], k1 = hi[p];
        k0 = max(k0, -base);
        k1 = min(k1, n - base);
        float acc = 0.f;
        for (int k = k0; k < k1; ++k) acc = fmaf(hp[k], mono_at(x, n, channels, base + k), acc);
        y[o] = acc;
        peak = fmaxf(peak, fabsf(acc));
    }
    __shared__ float s_peak[4];
    peak = wave_max(peak);
    if ((threadIdx.x & 63) == 0) s_peak[threadIdx.x >> 6] = peak;
    __syncthreads();
    if (threadIdx.x == 0) {                                        // one atomic per workgroup (see copy_peak_kernel)
        const float m = fmaxf(fmaxf(s_peak[0], s_peak[1]), fmaxf(s_peak[2], s_peak[3]));
        if (m > 0.f) atomicMax(&peak_keys[clip], __float_as_uint(m));
    }
}

// Reduced ratios beyond 1024 : 1024 (plans without a table, tables.cpp): the non-zero taps of an output sample --
// those with |t| < 6, about 12 max(orig, new) / new of them -- are evaluated where they are used, as the table builder
// evaluates them: t in float64, sin(pi t) / (pi t) * cos^2(pi t / 12) * scale rounded to float32 once.  sin and cos of
// consecutive taps advance by fixed float64 rotations (two sincos per output instead of two per tap; the rotation error,
// ~1e-16 per step over a dozen steps, moves the float32 rounding of a tap only when the exact value sits within 1e-15
// of a rounding boundary).  Same accumulation order as resample_generic_kernel (ascending k, fmaf).
template <typename TIn>
__global__ __launch_bounds__(256) void resample_onthefly_kernel(
    const TIn* __restrict__ pcm_in, const int64_t* __restrict__ in_off, const int32_t* __restrict__ in_len,
    int channels, const ResamplePlanHeader* __restrict__ plan, float* __restrict__ pcm_out,
    const int64_t* __restrict__ out_off, int32_t* __restrict__ out_len, unsigned* __restrict__ peak_keys) {
    const int clip = blockIdx.y;
    const int n = in_len[clip];
    const int orig = plan->orig_rate, nw = plan->new_rate, width = plan->width;
    const int n_out = (int)(((int64_t)nw * n + orig - 1) / orig);
    if (blockIdx.x == 0 && threadIdx.x == 0) out_len[clip] = n_out;
    const TIn* __restrict__ x = pcm_in + in_off[clip];
    float* __restrict__ y = pcm_out + out_off[clip];
    const double base = (double)(orig < nw ? orig : nw) * 0.99;
    const double scale = base / (double)orig, inv_orig = 1.0 / (double)orig;
    double rot_s, rot_c, rot_sw, rot_cw;
    sincos(M_PI * scale, &rot_s, &rot_c);                      // t advances by base / orig = scale per tap
    sincos(M_PI * scale / 12.0, &rot_sw, &rot_cw);
    const int half = (int)ceil(6.0 / scale) + 1;               // taps either side of t = 0 that can be non-zero
    float peak = 0.f;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x) {
        const int blk = o / nw, ph = o - blk * nw;
        const int64_t s0 = (int64_t)blk * orig - width;         // sample under tap k = 0
        // torch: arange(0, -new, -1) / new is an int64 / int true division -> a float32 quotient
        const double phase = (double)__fdiv_rn((float)(-ph), (float)nw);
        const int centre = width + (int)(((int64_t)ph * orig) / nw);         // tap nearest t = 0
        int k0 = centre - half, k1 = centre + half + 1;
        k0 = k0 < 0 ? 0 : k0;
        k1 = k1 > 2 * width + orig ? 2 * width + orig : k1;
        double sn, cs, sw, cw;
        {
            const double t0 = (phase + (double)(k0 - width) * inv_orig) * base;
            sincos(t0 * M_PI, &sn, &cs);
            sincos(t0 * M_PI / 12.0, &sw, &cw);
        }
        float acc = 0.f;
        for (int k = k0; k < k1; ++k) {
            const double t = (phase + (double)(k - width) * inv_orig) * base;
            const int64_t sidx = s0 + k;
            if (sidx >= 0 && sidx < n && fabs(t) < 6.0) {
                const double tp = t * M_PI;
                const double sinc = tp == 0.0 ? 1.0 : sn / tp;
                const float tap = (float)(sinc * (cw * cw) * scale);
                acc = fmaf(tap, mono_at(x, n, channels, (int)sidx), acc);
            }
            const double sn2 = sn * rot_c + cs * rot_s, sw2 = sw * rot_cw + cw * rot_sw;
            cs = cs * rot_c - sn * rot_s;
            cw = cw * rot_cw - sw * rot_sw;
            sn = sn2;
            sw = sw2;
        }
        y[o] = acc;
        peak = fmaxf(peak, fabsf(acc));
    }
    __shared__ float s_peak[4];
    peak = wave_max(peak);
    if ((threadIdx.x & 63) == 0) s_peak[threadIdx.x >> 6] = peak;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(s_peak[0], s_peak[1]), fmaxf(s_peak[2], s_peak[3]));
        if (m > 0.f) atomicMax(&peak_keys[clip], __float_as_uint(m));
    }
}

// Any ratio, LDS-tiled (44.1 / 22.05 / 8 kHz ... sources): the one-thread-per-output kernel above reads every tap and
// every sample from global memory with a different tap row per lane (2.9 ms for 512 Common-Voice-length clips at
// 44.1 kHz, twenty times the 48 kHz decimator).  Here a workgroup keeps the non-zero span of every phase's taps
// ([new_rate][lmax], lmax odd: lanes with consecutive phases read conflict-free) and the phases' spans in LDS for its
// whole life, stages the source span of a tile of `blocks` input blocks (blocks * orig + 2 width samples, zero outside
// the clip) per tile, and its threads take the tile's outputs in output order (coalesced stores): out[i * new + p] =
// sum_j taps[p][lo_p + j] * span[i * orig + lo_p + j], accumulated in the same order as the kernel above (samples
// outside the clip and the zero taps that pad a phase's row contribute acc + 0 == acc), so the two kernels give the
// same values (an exact zero may change sign).
template <typename TIn>
__global__ __launch_bounds__(256) void resample_generic_lds_kernel(
    const TIn* __restrict__ pcm_in, const int64_t* __restrict__ in_off, const int32_t* __restrict__ in_len,
    int channels, const ResamplePlanHeader* __restrict__ plan, float* __restrict__ pcm_out,
    const int64_t* __restrict__ out_off, int32_t* __restrict__ out_len, unsigned* __restrict__ peak_keys,
    int blocks, int lmax, int tiles_per_wg, int via_lds) {
    extern __shared__ __attribute__((aligned(16))) float g_smem[];
    const int clip = blockIdx.y, tid = threadIdx.x;
    const int n = in_len[clip];
    const int orig = plan->orig_rate, nw = plan->new_rate, width = plan->width, row = plan->taps_row;
    const int n_out = (int)(((int64_t)nw * n + orig - 1) / orig);
    if (blockIdx.x == 0 && tid == 0) out_len[clip] = n_out;
    const int n_blocks = (n_out + nw - 1) / nw;
    const int n_tiles = (n_blocks + blocks - 1) / blocks;
    const int tile_begin = blockIdx.x * tiles_per_wg;
    const int tile_end = min(tile_begin + tiles_per_wg, n_tiles);
    if (tile_begin >= tile_end) return;
    const int32_t* __restrict__ lo = reinterpret_cast<const int32_t*>(plan + 1);
    const int32_t* __restrict__ hi = lo + nw;
    const float* __restrict__ taps = reinterpret_cast<const float*>(hi + nw);
    float* taps_c = g_smem;                                            // [nw][lmax]
    int* lo_s = reinterpret_cast<int*>(g_smem + nw * lmax);            // [nw]
    float* xs = reinterpret_cast<float*>(lo_s + nw);                   // [blocks * orig + 2 width + lmax]
    float* out_s = xs + blocks * orig + 2 * width + lmax;              // [blocks * nw] (via_lds only)
    for (int p = tid; p < nw; p += 256) lo_s[p] = lo[p];
    for (int e = tid; e < nw * lmax; e += 256) {                       // rows zero-filled behind the phase's own span
        const int p = e / lmax, j = e - p * lmax;
        const int k = lo[p] + j;
        taps_c[e] = k < hi[p] ? taps[(size_t)p * row + k] : 0.f;
    }
    const TIn* __restrict__ x = pcm_in + in_off[clip];
    float* __restrict__ y = pcm_out + out_off[clip];
    const int span = blocks * orig + 2 * width;
    for (int e = tid; e < lmax; e += 256) xs[span + e] = 0.f;          // the padded taps of the last block read zeros here
    float peak = 0.f;
    // spans of up to 16 samples per thread are requested a tile ahead (registers): the HBM round trip of tile t + 1
    // runs under the arithmetic of tile t; longer spans are staged in place
    constexpr int kStage = 16;
    const bool ahead = span <= kStage * 256;
    float st[kStage];
    auto request = [&](int tile) {
        const int s0 = tile * blocks * orig - width;
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int e = tid + 256 * i, src = s0 + e;
            st[i] = (e < span && src >= 0 && src < n) ? mono_at(x, n, channels, src) : 0.f;
        }
    };
    if (ahead) request(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();                                               // the previous tile's reads are done (and the tables are in)
        if (ahead) {
#pragma unroll
            for (int i = 0; i < kStage; ++i)
                if (tid + 256 * i < span) xs[tid + 256 * i] = st[i];
        } else {
            const int s0 = tile * blocks * orig - width;
            for (int e = tid; e < span; e += 256) {
                const int src = s0 + e;
                xs[e] = (src >= 0 && src < n) ? mono_at(x, n, channels, src) : 0.f;
            }
        }
        __syncthreads();
        if (ahead && tile + 1 < tile_end) request(tile + 1);
        const int o_base = tile * blocks * nw;
        const int n_here = min(blocks * nw, n_out - o_base);
        auto dot = [&](int il, int p) {
            const float* __restrict__ xp = xs + il * orig + lo_s[p];
            const float* __restrict__ tp = taps_c + p * lmax;
            float acc = 0.f;
            // eight taps and eight samples in flight, then their FMAs in tap order; lmax - 1 is a multiple of 8 and the
            // taps behind a phase's span are 0.0 (acc + 0 * sample == acc), so the trip count is the same for every lane
            for (int j0 = 0; j0 < lmax - 1; j0 += 8) {
                float t8[8], x8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    t8[j] = tp[j0 + j];
                    x8[j] = xp[j0 + j];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = fmaf(t8[j], x8[j], acc);
            }
            return acc;
        };
        if (via_lds) {
            // odd orig >= 3 and tiles of >= 64 blocks: lanes take consecutive input blocks of ONE phase -- their sample reads are orig words apart
            // (conflict-free for odd orig; with consecutive phases per lane they start ~orig / new words apart and
            // collide about three ways) and the phase's taps are a broadcast; the outputs, new_rate apart for such
            // lanes, go through an LDS tile and leave in output order
            for (int e = tid; e < blocks * nw; e += 256) {
                const int p = e / blocks, il = e - p * blocks;
                const int ol = il * nw + p;
                if (ol < n_here) out_s[ol] = dot(il, p);
            }
            __syncthreads();
            for (int e = tid; e < n_here; e += 256) {
                const float v = out_s[e];
                y[o_base + e] = v;
                peak = fmaxf(peak, fabsf(v));
            }
        } else {
            for (int e = tid; e < n_here; e += 256) {
                const int il = e / nw, p = e - il * nw;
                const float acc = dot(il, p);
                y[o_base + e] = acc;
                peak = fmaxf(peak, fabsf(acc));
            }
        }
    }
    __shared__ float s_peak[4];
    peak = wave_max(peak);
    if ((tid & 63) == 0) s_peak[tid >> 6] = peak;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(s_peak[0], s_peak[1]), fmaxf(s_peak[2], s_peak[3]));
        if (m > 0.f) atomicMax(&peak_keys[clip], __float_as_uint(m));
    }
}

// The same for new_rate % 4 == 0 (441:160, 441:320 ...: the 44.1 / 22.05 / 11.025 kHz sources): a thread computes
// P = 4 consecutive outputs of one input block (P = 2 compiles too; it was slower than one output per thread).  Their spans start only orig / new samples apart, so they share one window
// of samples -- one LDS read per sample for four FMAs instead of one read per FMA -- and the four phases' taps for the
// window position m sit side by side ([quad][m][4], absolute tap index, zero outside a phase's own span): one 16-byte
// read.  Row stride 4 * lqs words with lqs odd: the 16 lanes of a b128 read pass hit all 64 banks.  Same accumulation
// order per output as the kernels above.
template <typename TIn, int kStage, int P>
__global__ __launch_bounds__(256) void resample_generic_quad_kernel(
    const TIn* __restrict__ pcm_in, const int64_t* __restrict__ in_off, const int32_t* __restrict__ in_len,
    int channels, const ResamplePlanHeader* __restrict__ plan, float* __restrict__ pcm_out,
    const int64_t* __restrict__ out_off, int32_t* __restrict__ out_len, unsigned* __restrict__ peak_keys,
    int blocks, int lqs, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float g_smem[];
    const int clip = blockIdx.y, tid = threadIdx.x;
    const int n = in_len[clip];
    const int orig = plan->orig_rate, nw = plan->new_rate, width = plan->width, row = plan->taps_row;
    static_assert(P == 2 || P == 4, "outputs per thread");
    typedef float fP __attribute__((ext_vector_type(P)));
    const int nq = nw / P;                                             // groups of P consecutive phases
    const int n_out = (int)(((int64_t)nw * n + orig - 1) / orig);
    if (blockIdx.x == 0 && tid == 0) out_len[clip] = n_out;
    const int n_blocks = (n_out + nw - 1) / nw;
    const int n_tiles = (n_blocks + blocks - 1) / blocks;
    const int tile_begin = blockIdx.x * tiles_per_wg;
    const int tile_end = min(tile_begin + tiles_per_wg, n_tiles);
    if (tile_begin >= tile_end) return;
    const int32_t* __restrict__ lo = reinterpret_cast<const int32_t*>(plan + 1);
    const int32_t* __restrict__ hi = lo + nw;
    const float* __restrict__ taps = reinterpret_cast<const float*>(hi + nw);
    float* taps_q = g_smem;                                            // [nq][lqs][P]
    int* w0_s = reinterpret_cast<int*>(g_smem + nq * lqs * P);         // [nq] first tap index of the group's window
    float* xs = reinterpret_cast<float*>(w0_s + nq);                   // [blocks * orig + 2 width + lqs]
    auto group_lo = [&](int q) {
        int l0 = lo[P * q];
#pragma unroll
        for (int c = 1; c < P; ++c) l0 = min(l0, lo[P * q + c]);
        return l0;
    };
    for (int q = tid; q < nq; q += 256) w0_s[q] = group_lo(q);
    for (int e = tid; e < nq * lqs * P; e += 256) {
        const int q = e / (lqs * P), r = e - q * (lqs * P);
        const int m = r / P, p = P * q + (r - m * P);
        const int k = group_lo(q) + m;
        taps_q[e] = (k >= lo[p] && k < hi[p]) ? taps[(size_t)p * row + k] : 0.f;
    }
    const TIn* __restrict__ x = pcm_in + in_off[clip];
    float* __restrict__ y = pcm_out + out_off[clip];
    const int span = blocks * orig + 2 * width;
    for (int e = tid; e < lqs; e += 256) xs[span + e] = 0.f;           // the zero taps at the end of the last block's rows read this
    const bool y_vec = (reinterpret_cast<uintptr_t>(y) & (4 * P - 1)) == 0;
    float peak = 0.f;
    // spans of up to kStage (16 or 24, the host picks the instantiation) samples per thread are requested a tile ahead
    // (registers), so that the HBM round trip of tile t + 1 runs under the arithmetic of tile t; longer spans are
    // staged in place
    const bool ahead = span <= kStage * 256;
    float st[kStage];
    auto request = [&](int tile) {
        const int s0 = tile * blocks * orig - width;
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int e = tid + 256 * i, src = s0 + e;
            st[i] = (e < span && src >= 0 && src < n) ? mono_at(x, n, channels, src) : 0.f;
        }
    };
    if (ahead) request(tile_begin);
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();
        if (ahead) {
#pragma unroll
            for (int i = 0; i < kStage; ++i)
                if (tid + 256 * i < span) xs[tid + 256 * i] = st[i];
        } else {
            const int s0 = tile * blocks * orig - width;
            for (int e = tid; e < span; e += 256) {
                const int src = s0 + e;
                xs[e] = (src >= 0 && src < n) ? mono_at(x, n, channels, src) : 0.f;
            }
        }
        __syncthreads();
        if (ahead && tile + 1 < tile_end) request(tile + 1);
        const int o_base = tile * blocks * nw;                          // a multiple of P
        const int n_here = min(blocks * nw, n_out - o_base);
        for (int e = tid; e < blocks * nq; e += 256) {
            const int il = e / nq, q = e - il * nq;
            const int ol = il * nw + P * q;
            if (ol >= n_here) continue;
            const float* __restrict__ xp = xs + il * orig + w0_s[q];
            const fP* __restrict__ tq = reinterpret_cast<const fP*>(taps_q + (size_t)q * lqs * P);
            fP acc = 0.f;
            for (int m0 = 0; m0 < lqs - 1; m0 += 4) {                   // lqs - 1 is a multiple of 4
                float x4[4];
                fP t4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x4[j] = xp[m0 + j];
                    t4[j] = tq[m0 + j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < P; ++c) acc[c] = fmaf(t4[j][c], x4[j], acc[c]);
            }
            if (y_vec && ol + P <= n_here) {
                *reinterpret_cast<fP*>(y + o_base + ol) = acc;
#pragma unroll
                for (int c = 0; c < P; ++c) peak = fmaxf(peak, fabsf(acc[c]));
            } else {
#pragma unroll
                for (int c = 0; c < P; ++c)
                    if (ol + c < n_here) {
                        y[o_base + ol + c] = acc[c];
                        peak = fmaxf(peak, fabsf(acc[c]));
                    }
            }
        }
    }
    __shared__ float s_peak[4];
    peak = wave_max(peak);
    if ((tid & 63) == 0) s_peak[tid >> 6] = peak;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(s_peak[0], s_peak[1]), fmaxf(s_peak[2], s_peak[3]));
        if (m > 0.f) atomicMax(&peak_keys[clip], __float_as_uint(m));
    }
}

// Same-rate path: optional mono mix / copy, and the per-clip peak.
template <typename TIn>
__global__ __launch_bounds__(256) void copy_peak_kernel(
    const TIn* __restrict__ pcm_in, const int64_t* __restrict__ in_off, const int32_t* __restrict__ in_len,
    int channels, float* __restrict__ pcm_out, const int64_t* __restrict__ out_off, int32_t* __restrict__ out_len,
    unsigned* __restrict__ peak_keys) {
    const int clip = blockIdx.y;
    const int n = in_len[clip];
    if (blockIdx.x == 0 && threadIdx.x == 0) out_len[clip] = n;
    const TIn* __restrict__ x = pcm_in + in_off[clip];
    float* __restrict__ y = pcm_out + out_off[clip];
    const bool store = (static_cast<const void*>(y) != static_cast<const void*>(x)) || channels != 1;
    float peak = 0.f;
    constexpr bool kIsF32 = sizeof(TIn) == 4;
    if (kIsF32 && channels == 1 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (!store || (reinterpret_cast<uintptr_t>(y) & 15) == 0)) {
        // mono float32: eight samples per thread and trip as two 16-byte loads
        const float* __restrict__ xf = reinterpret_cast<const float*>(x);
        for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8; i0 < n; i0 += gridDim.x * blockDim.x * 8) {
            if (i0 + 8 <= n) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(xf + i0), b = *reinterpret_cast<const f32x4*>(xf + i0 + 4);
                if (store) {
                    *reinterpret_cast<f32x4*>(y + i0) = a;
                    *reinterpret_cast<f32x4*>(y + i0 + 4) = b;
                }
                peak = fmaxf(peak, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                                         fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
            } else {
                for (int i = i0; i < n; ++i) {
                    const float v = xf[i];
                    if (store) y[i] = v;
                    peak = fmaxf(peak, fabsf(v));
                }
            }
        }
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const float v = mono_at(x, n, channels, i);
            if (store) y[i] = v;
            peak = fmaxf(peak, fabsf(v));
        }
    }
    // one atomic per workgroup: device-scope read-modify-writes on a clip's one word queue behind each other (with one
    // per wave and 83 workgroups per clip this pass ran at 0.7 TB/s)
    __shared__ float s_peak[4];
    peak = wave_max(peak);
    if ((threadIdx.x & 63) == 0) s_peak[threadIdx.x >> 6] = peak;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(s_peak[0], s_peak[1]), fmaxf(s_peak[2], s_peak[3]));
        if (m > 0.f) atomicMax(&peak_keys[clip], __float_as_uint(m));
    }
}

__global__ __launch_bounds__(256) void normalize_trim_kernel(
    float* __restrict__ pcm, const int64_t* __restrict__ offsets, const int32_t* __restrict__ lengths,
    const unsigned* __restrict__ peak_keys, int normalize, int trim, float thr_factor,
    int* __restrict__ first, int* __restrict__ last) {
    const int clip = blockIdx.y;
    const int n = lengths[clip];
    const float peak = __uint_as_float(peak_keys[clip]);
    if (!(peak > 0.f)) return;                  // src/data.py:80 `if max_abs > 0`; nothing is above 0 * thr either
    float* __restrict__ x = pcm + offsets[clip];
    const float d = peak + 1e-8f;               // src/data.py:81 (float32 tensor + python scalar -> float32)
    const float top = normalize ? __fdiv_rn(peak, d) : peak;      // max|y|: division is monotonic
    const float thr = top * thr_factor;         // src/data.py:86
    int lo = 0x7fffffff, hi = -1;
    // 8 elements per thread and trip: all loads are issued before the first store (in-place update: the
    // compiler cannot hoist the next load above a store to the same array)
    const int span = gridDim.x * blockDim.x * 8;
    const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8; i0 < n; i0 += span) {
        float v[8];
        if (vec && i0 + 8 <= n) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(x + i0), b = *reinterpret_cast<const f32x4*>(x + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = (i0 + c < n) ? x[i0 + c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (normalize) v[c] = __fdiv_rn(v[c], d);
            if (trim && i0 + c < n && fabsf(v[c]) > thr) {
                lo = min(lo, i0 + c);
                hi = max(hi, i0 + c);
            }
        }
        if (normalize) {
            if (vec && i0 + 8 <= n) {
                *reinterpret_cast<f32x4*>(x + i0) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(x + i0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (i0 + c < n) x[i0 + c] = v[c];
            }
        }
    }
    if (trim) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo = min(lo, __shfl_xor(lo, off, 64));
            hi = max(hi, __shfl_xor(hi, off, 64));
        }
        // One candidate per workgroup, into slot blockIdx.x % kTrimSlots of the clip: nearly every wave holds a sample
        // above the threshold, and two device-scope read-modify-writes per wave on the clip's two words queued behind
        // each other (config 3 with trimming: 0.186 ms for this pass against 0.054 ms without; reading the words first
        // to skip the atomics that cannot win was slower still, 0.39 ms -- an agent-scope load is as slow as the atomic).
        // trim_finalize_kernel reduces the slots.
        __shared__ int s_lo, s_hi;
        if (threadIdx.x == 0) {
            s_lo = 0x7fffffff;
            s_hi = -1;
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && hi >= 0) {
            atomicMin(&s_lo, lo);
            atomicMax(&s_hi, hi);
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_hi >= 0) {
            const size_t slot = (size_t)clip * kTrimSlots + (blockIdx.x & (kTrimSlots - 1));
            atomicMin(&first[slot], s_lo);
            atomicMax(&last[slot], s_hi);
        }
    }
}

// One wave per clip: min / max over the clip's slots, then the trimmed offset / length.
__global__ __launch_bounds__(256) void trim_finalize_kernel(int64_t* __restrict__ offsets, int32_t* __restrict__ lengths,
                                                            const int* __restrict__ first, const int* __restrict__ last,
                                                            int batch) {
    static_assert(kTrimSlots == 64, "one lane per slot");
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= batch) return;
    int f = first[(size_t)b * kTrimSlots + lane], l = last[(size_t)b * kTrimSlots + lane];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        f = min(f, __shfl_xor(f, off, 64));
        l = max(l, __shfl_xor(l, off, 64));
    }
    if (lane == 0 && l >= f && l >= 0) {        // src/data.py:87 `if mask.any()`
        offsets[b] += f;
        lengths[b] = l + 1 - f;
    }
}

}  // namespace sttfe

using namespace sttfe;

#ifdef STTFE_STAMPS
extern "C" int sttfe_debug_set_rs_stamp_buffer(void* dev_buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(dev_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_rs_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -4;
}
#endif

// workspace layout: [peak keys u32 x B][first i32 x B x kTrimSlots][last i32 x B x kTrimSlots], 256-byte aligned total
extern "C" int sttfe_preprocess_workspace_bytes(int batch, size_t* bytes) {
    if (batch < 0 || !bytes) return STTFE_E_INVALID_ARG;
    *bytes = ((size_t)batch * (1 + 2 * kTrimSlots) * sizeof(int32_t) + 255) / 256 * 256;
    return STTFE_OK;
}

namespace sttfe { int current_device_cus(); }

// Workgroups of the decimator that one CU holds (its registers and LDS decide), looked up once per instantiation.
template <int O, typename TIn>
static int decimate_resident_per_cu() {
    static const int v = [] {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, decimate_peak_kernel<O, TIn>, rs::kThreads, 0) != hipSuccess ||
            nb < 1)
            nb = 4;
        return nb;
    }();
    return v;
}

template <typename TIn>
static int resample_peak_impl(const TIn* pcm_in, const int64_t* in_offsets, const int32_t* in_lengths,
                              int batch, int channels, int64_t max_in_length,
                              const void* dev_plan, const void* host_plan,
                              float* pcm_out, const int64_t* out_offsets, int32_t* out_lengths,
                              void* workspace, size_t workspace_bytes, void* stream_) {
    if (batch < 0 || channels < 1 || channels > 64 || max_in_length < 0) return STTFE_E_INVALID_ARG;
    if (batch == 0) return STTFE_OK;
    if (!pcm_in || !in_offsets || !in_lengths || !pcm_out || !out_offsets || !out_lengths || !workspace)
        return STTFE_E_INVALID_ARG;
    if ((dev_plan == nullptr) != (host_plan == nullptr)) return STTFE_E_INVALID_ARG;
    if (batch > 65535) return STTFE_E_UNSUPPORTED;
    size_t need = 0;
    sttfe_preprocess_workspace_bytes(batch, &need);
    if (workspace_bytes < need) return STTFE_E_WORKSPACE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    unsigned* keys = reinterpret_cast<unsigned*>(workspace);
    (void)hipGetLastError();
    if (hip_failed(hipMemsetAsync(keys, 0, (size_t)batch * sizeof(unsigned), stream), "hipMemsetAsync(peak keys)"))
        return STTFE_E_LAUNCH;
    ScopedTiming t(kTimeResample, stream);
    if (!dev_plan) {
        int bx = (int)((max_in_length + 256 * 8 * 4 - 1) / (256 * 8 * 4));      // four 8-sample trips per thread
        bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
        hipLaunchKernelGGL(copy_peak_kernel<TIn>, dim3(bx, batch), dim3(256), 0, stream, pcm_in, in_offsets, in_lengths,
                           channels, pcm_out, out_offsets, out_lengths, keys);
    } else {
        const auto* hp = reinterpret_cast<const ResamplePlanHeader*>(host_plan);
        if (hp->magic != kResampleMagic) return STTFE_E_TABLES;
        const auto* dp = reinterpret_cast<const ResamplePlanHeader*>(dev_plan);
        const int64_t max_out = (hp->new_rate * max_in_length + hp->orig_rate - 1) / hp->orig_rate;
        const bool fast = hp->new_rate == 1 && hp->taps_row <= rs::kTapsPerPhase * hp->orig_rate &&
                          (hp->orig_rate == 2 || hp->orig_rate == 3 || hp->orig_rate == 4 || hp->orig_rate == 6);
        if (hp->taps_row == 0) {                                     // a plan without a table: reduced rates beyond 1024
            if (max_out > 0x7fffffffLL) return STTFE_E_UNSUPPORTED;
            int bx = (int)((max_out + 256 * 2 - 1) / (256 * 2));
            bx = bx < 1 ? 1 : (bx > 4096 ? 4096 : bx);
            hipLaunchKernelGGL(resample_onthefly_kernel<TIn>, dim3(bx, batch), dim3(256), 0, stream, pcm_in, in_offsets,
                               in_lengths, channels, dp, pcm_out, out_offsets, out_lengths, keys);
        } else if (fast) {
            static const int tpb_env = getenv("STTFE_DECIMATE_TPB") ? atoi(getenv("STTFE_DECIMATE_TPB")) : 0;   // tuning knobs
            // one item per workgroup by default: on the final kernel (8-tile items, one peak atomic per workgroup, vector
            // staging) the persistent form measured 3-4 % slower (float 0.183 / 0.176 ms, int16 0.125 / 0.122 ms) and, holding
            // every CU for the whole launch, overlaps less with another stream's kernels (config 3 fused, two batches in
            // flight: 0.378 / 0.357 ms per step); STTFE_DECIMATE_PERSIST=1 selects it
            static const int persist_env = getenv("STTFE_DECIMATE_PERSIST") ? atoi(getenv("STTFE_DECIMATE_PERSIST")) : 0;
            const int tiles_per_block = tpb_env > 0 ? tpb_env : 8;     // 4 / 6 / 8 / 12 / 16: 0.188 / 0.179 / 0.177 / 0.183 / 0.186 ms (config 3)
            const int64_t n_tiles = (max_out + rs::kTile - 1) / rs::kTile;
            int bx = (int)((n_tiles + tiles_per_block - 1) / tiles_per_block);
            bx = bx < 1 ? 1 : bx;
            const bool persistent = persist_env != 0 && batch <= rs::kMaxPlanClips;
            const dim3 block(rs::kThreads);
#define STTFE_LAUNCH_DECIMATE(O)                                                                                   \
    {                                                                                                               \
        dim3 grid(bx, batch);                                                                                       \
        if (persistent) {                                                                                           \
            const long resident = (long)decimate_resident_per_cu<O, TIn>() * current_device_cus();                  \
            const long most = (long)bx * batch;                                                                     \
            grid = dim3((unsigned)(most < resident ? most : resident));                                             \
        }                                                                                                           \
        hipLaunchKernelGGL((decimate_peak_kernel<O, TIn>), grid, block, 0, stream, pcm_in, in_offsets, in_lengths,  \
                           channels, dp, pcm_out, out_offsets, out_lengths, keys, tiles_per_block, batch,           \
                           persistent ? 1 : 0);                                                                     \
    }
            switch (hp->orig_rate) {
                case 2: STTFE_LAUNCH_DECIMATE(2); break;
                case 3: STTFE_LAUNCH_DECIMATE(3); break;
                case 4: STTFE_LAUNCH_DECIMATE(4); break;
                default: STTFE_LAUNCH_DECIMATE(6); break;
            }
#undef STTFE_LAUNCH_DECIMATE
        } else {
            // LDS-tiled form when the compact taps (new_rate x lmax), the spans and at least one input block fit 64 KB
            const int nw = hp->new_rate, orig = hp->orig_rate;
            const int32_t* hlo = reinterpret_cast<const int32_t*>(hp + 1);
            const int32_t* hhi = hlo + nw;
            int lmax = 1;
            for (int p = 0; p < nw; ++p) lmax = hhi[p] - hlo[p] > lmax ? hhi[p] - hlo[p] : lmax;
            lmax = ((lmax + 7) & ~7) + 1;                               // taps in groups of 8; odd row stride: consecutive phases land in different banks
            static const bool lds_off = getenv("STTFE_RESAMPLE_GENERIC_GLOBAL") != nullptr;     // A/B and test knob
            static const int tile_env = getenv("STTFE_RESAMPLE_GENERIC_TILE") ? atoi(getenv("STTFE_RESAMPLE_GENERIC_TILE")) : 0;
            const long words = (64 * 1024 - 256) / 4;                   // 64 KB of LDS per workgroup, minus the kernel's static words
            const long fixed = (long)nw * lmax + nw + 2L * hp->width + lmax;
            // outputs per tile: small tiles keep the LDS footprint low (more workgroups per CU hide the LDS latency of the
            // tap loop), large ones amortise the 2 width + lmax halo of the staged span
            const long target = tile_env > 0 ? tile_env : 1024;
            long blocks = (target + nw - 1) / nw;
            // phase-major lanes + an LDS output tile (see the kernel) pay when a wave stays inside one phase and the
            // sample stride is odd and not 1: 3:2 0.33 -> 0.27 ms; 441:160 (7 blocks per tile) 0.67 -> 0.86 ms without this test
            const bool via_lds = (orig & 1) && orig >= 3 && blocks >= 64 && fixed + blocks * (orig + nw) <= words;
            const long per_block = orig + (via_lds ? nw : 0);
            if (fixed + blocks * per_block > words) blocks = (words - fixed) / per_block;
            // even new_rate: P = 4 or 2 consecutive outputs per thread off one window of samples
            // (P = 2 for the other even rates -- 3:2, 1:2, 9:10, 11:10 -- measured slower than one output per thread:
            // 0.18 -> 0.23, 0.26 -> 0.28, 0.23 -> 0.26 ms)
            const int P = (nw & 3) == 0 ? 4 : 0;
            int lq = 0;
            if (P)
                for (int q = 0; q < nw / P; ++q) {
                    int l0 = hlo[P * q], h1 = hhi[P * q];
                    for (int c = 1; c < P; ++c) {
                        l0 = hlo[P * q + c] < l0 ? hlo[P * q + c] : l0;
                        h1 = hhi[P * q + c] > h1 ? hhi[P * q + c] : h1;
                    }
                    lq = h1 - l0 > lq ? h1 - l0 : lq;
                }
            const int lqs = ((lq + 3) & ~3) + 1;
            const long fixed_q = (long)nw * lqs + (P ? nw / P : 0) + 2L * hp->width + lqs;
            // a tile is (input block, phase group) items for 256 threads: the fewest blocks that leave at most a tenth
            // of the last round idle, while the span still fits the registers that prefetch it (24 samples per thread)
            // and the LDS
            long blocks_q = 0;
            if (P) {
                const long nq = nw / P;
                double best = 0.0;
                for (long b = 1; b <= 4096; ++b) {
                    if (fixed_q + b * orig > words || (b > 1 && b * orig + 2L * hp->width > 24 * 256)) break;
                    const long items = b * nq;
                    const double util = (double)items / (256.0 * ((items + 255) / 256));
                    if (util > best + 1e-9) {
                        best = util;
                        blocks_q = b;
                    }
                    if (util >= 0.9 && items >= (tile_env > 0 ? tile_env / 4 : 240)) break;
                }
            }
            static const bool quad_off = getenv("STTFE_RESAMPLE_GENERIC_NOQUAD") != nullptr;     // A/B knob
            if (!lds_off && !quad_off && lq > 0 && fixed_q < words && blocks_q >= 1) {
                const long tile_out = blocks_q * nw;
                const int tiles_per_wg = (int)(tile_out >= 32768 ? 1 : (32768 + tile_out - 1) / tile_out);
                const int64_t n_tiles = ((max_out + nw - 1) / nw + blocks_q - 1) / blocks_q;
                int bx = (int)((n_tiles + tiles_per_wg - 1) / tiles_per_wg);
                bx = bx < 1 ? 1 : bx;
                const size_t lds = (size_t)(fixed_q + blocks_q * orig) * 4;
                const bool s16k = blocks_q * orig + 2L * hp->width <= 16 * 256;
#define STTFE_LAUNCH_GROUPS(S, PP)                                                                                   \
    hipLaunchKernelGGL((resample_generic_quad_kernel<TIn, S, PP>), dim3(bx, batch), dim3(256), lds, stream, pcm_in,   \
                       in_offsets, in_lengths, channels, dp, pcm_out, out_offsets, out_lengths, keys, (int)blocks_q,  \
                       lqs, tiles_per_wg)
                if (s16k) STTFE_LAUNCH_GROUPS(16, 4);
                else STTFE_LAUNCH_GROUPS(24, 4);
#undef STTFE_LAUNCH_GROUPS
            } else if (!lds_off && fixed < words && blocks >= 1) {
                const long tile_out = blocks * nw;
                const int tiles_per_wg = (int)(tile_out >= 32768 ? 1 : (32768 + tile_out - 1) / tile_out);
                const int64_t n_tiles = ((max_out + nw - 1) / nw + blocks - 1) / blocks;
                int bx = (int)((n_tiles + tiles_per_wg - 1) / tiles_per_wg);
                bx = bx < 1 ? 1 : bx;
                const size_t lds = (size_t)(fixed + blocks * per_block) * 4;
                hipLaunchKernelGGL(resample_generic_lds_kernel<TIn>, dim3(bx, batch), dim3(256), lds, stream, pcm_in,
                                   in_offsets, in_lengths, channels, dp, pcm_out, out_offsets, out_lengths, keys,
                                   (int)blocks, lmax, tiles_per_wg, via_lds ? 1 : 0);
            } else {
                int bx = (int)((max_out + 256 * 4 - 1) / (256 * 4));
                bx = bx < 1 ? 1 : (bx > 2048 ? 2048 : bx);
                hipLaunchKernelGGL(resample_generic_kernel<TIn>, dim3(bx, batch), dim3(256), 0, stream, pcm_in, in_offsets,
                                   in_lengths, channels, dp, pcm_out, out_offsets, out_lengths, keys);
            }
        }
    }
    if (hip_failed(hipGetLastError(), "resample kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}

extern "C" int sttfe_resample_peak(const float* pcm_in, const int64_t* in_offsets, const int32_t* in_lengths,
                                   int batch, int channels, int64_t max_in_length,
                                   const void* dev_plan, const void* host_plan,
                                   float* pcm_out, const int64_t* out_offsets, int32_t* out_lengths,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return resample_peak_impl<float>(pcm_in, in_offsets, in_lengths, batch, channels, max_in_length, dev_plan, host_plan,
                                     pcm_out, out_offsets, out_lengths, workspace, workspace_bytes, stream);
}

extern "C" int sttfe_resample_peak_s16(const int16_t* pcm_in, const int64_t* in_offsets, const int32_t* in_lengths,
                                       int batch, int channels, int64_t max_in_length,
                                       const void* dev_plan, const void* host_plan,
                                       float* pcm_out, const int64_t* out_offsets, int32_t* out_lengths,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return resample_peak_impl<int16_t>(pcm_in, in_offsets, in_lengths, batch, channels, max_in_length, dev_plan,
                                       host_plan, pcm_out, out_offsets, out_lengths, workspace, workspace_bytes, stream);
}

extern "C" int sttfe_normalize_trim(float* pcm, int64_t* offsets, int32_t* lengths, int batch,
                                    int64_t max_length, int normalize, int trim, float silence_threshold,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
    if (batch < 0 || max_length < 0) return STTFE_E_INVALID_ARG;
    if (batch == 0 || (!normalize && !trim)) return STTFE_OK;
    if (!pcm || !offsets || !lengths || !workspace) return STTFE_E_INVALID_ARG;
    if (batch > 65535) return STTFE_E_UNSUPPORTED;
    size_t need = 0;
    sttfe_preprocess_workspace_bytes(batch, &need);
    if (workspace_bytes < need) return STTFE_E_WORKSPACE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    unsigned* keys = reinterpret_cast<unsigned*>(workspace);
    int* first = reinterpret_cast<int*>(keys + batch);
    int* last = first + (size_t)batch * kTrimSlots;
    (void)hipGetLastError();
    if (trim) {
        if (hip_failed(hipMemsetAsync(first, 0x7f, (size_t)batch * kTrimSlots * sizeof(int), stream), "hipMemsetAsync(first)") ||
            hip_failed(hipMemsetAsync(last, 0xff, (size_t)batch * kTrimSlots * sizeof(int), stream), "hipMemsetAsync(last)"))
            return STTFE_E_LAUNCH;
    }
    {
        ScopedTiming t(kTimeNormTrim, stream);
        int bx = (int)((max_length + 256 * 8 - 1) / (256 * 8));      // one 8-element trip per thread
        bx = bx < 1 ? 1 : (bx > 4096 ? 4096 : bx);
        hipLaunchKernelGGL(normalize_trim_kernel, dim3(bx, batch), dim3(256), 0, stream, pcm, offsets, lengths, keys,
                           normalize, trim, silence_threshold, first, last);
        if (trim)
            hipLaunchKernelGGL(trim_finalize_kernel, dim3((batch + 3) / 4), dim3(256), 0, stream, offsets, lengths,
                               first, last, batch);
    }
    if (hip_failed(hipGetLastError(), "normalize/trim kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}
