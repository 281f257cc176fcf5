// Pitch shift of the reference's waveform augmentation, batched and in place on packed PCM.
//
// Reference: src/data.py:133-136 builds torchaudio.transforms.PitchShift(sample_rate, n_steps=0) and
// src/data.py:163-167 overwrites its n_steps with random.randint(-2, 2) per clip before calling it.  torchaudio
// 2.5.1 computes (functional.pitch_shift): rate = 2^(-n_steps/12);
//   S = stft(x, 512, hop 128, periodic hann, center / reflect)                    -> pv_stft_kernel
//   S' = phase_vocoder(S, rate, linspace(0, pi*128, 257))                         -> pv_stretch_kernel (S kept as |S|, angle S)
//   y = istft(S', length = round(n / rate))                                       -> pv_synth_kernel
//   y = resample(y, int(sr / rate) -> sr), sinc_interp_hann, width 6, rolloff .99 -> pv_resample_kernel
//   crop / zero-pad y to n samples                                                -> fused into the last kernel
// STTFE_PITCH_AS_CONSTRUCTED is what the reference's transform object does: its orig_freq is fixed in the
// constructor, where n_steps is 0, so the resample is skipped and the clip comes back time-stretched and cropped /
// padded; STTFE_PITCH_FUNCTIONAL is the full functional.pitch_shift with the drawn n_steps.
// Parity unpinned (torchaudio absent): restated from the algorithm text; oracle/pitch_shift.py runs the same
// steps on the real torch.stft / torch.istft.
//
// Data: per active clip the workspace holds S [T][257] (abs, angle), S' [T'][257] complex and (functional mode)
// the stretched waveform; `layout` rows (built on the host by sttfe_pitch_shift_plan from the host's copy of the
// lengths) carry the offsets, so no kernel derives a size from device data: a clip whose device length differs
// from the planned one is left untouched.
// 512-point transforms: one wave per pair of real frames (z = x1 + i x2), 8 points per lane, three radix-8
// Stockham passes through a 4 KB LDS panel, twiddles and window from LDS tables (sincospif of exact arguments).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>

#include "../../include/sttfe.h"
#include "fft_core.h"
#include "timing.h"

// torch rounds every product and sum separately (see waveform_aug.hip)
#pragma clang fp contract(off)

namespace sttfe {

constexpr int kPvN = 512, kPvHop = 128, kPvBins = 257, kPvPad = 256;
constexpr int kPvWaves = 4;
constexpr int kLayoutWords = 8;        // int64 per clip: spec, stretch, (unused), wave offsets (floats); T | T'<<32;
                                       // n | len_stretch<<32; rate (double bits); orig | new<<32 (reduced rates)

struct PvClip {
    int64_t spec, stretch, wave;
    int T, T_out, n, len_stretch;
    double rate;
    int orig, neu;
};

__device__ __forceinline__ PvClip load_clip(const int64_t* __restrict__ layout, int clip) {
    const int64_t* r = layout + (int64_t)clip * kLayoutWords;
    PvClip c;
    c.spec = r[0];
    c.stretch = r[1];
    c.wave = r[3];
    c.T = (int)(r[4] & 0xffffffff);
    c.T_out = (int)(r[4] >> 32);
    c.n = (int)(r[5] & 0xffffffff);
    c.len_stretch = (int)(r[5] >> 32);
    c.rate = __longlong_as_double(r[6]);
    c.orig = (int)(r[7] & 0xffffffff);
    c.neu = (int)(r[7] >> 32);
    return c;
}

__device__ __forceinline__ void fill_tables(cf* tw, float* win) {
    for (int i = threadIdx.x; i < kPvN; i += blockDim.x) {
        float s, c;
        sincospif(-2.0f * (float)i / kPvN, &s, &c);                        // arguments exact in float32
        tw[i] = cf{c, s};
        win[i] = 0.5f - 0.5f * cospif(2.0f * (float)i / kPvN);             // torch.hann_window(512), periodic
    }
}

// A wave's LDS panel is private to it: its own writes and reads only need ordering within the wave.
__device__ __forceinline__ void wave_panel_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Forward 512-point DFT of the calling wave's panel (natural order in and out).
__device__ __forceinline__ void fft512_panel(cf* __restrict__ panel, const cf* __restrict__ tw, int lane) {
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int ns = pass == 0 ? 1 : (pass == 1 ? 8 : 64);
        const int k = lane & (ns - 1);
        const int tw_step = (kPvN / 8) / ns * k;                           // r * k * 512 / (8 ns)
        cf v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = panel[lane + 64 * r];
        if (pass > 0) {
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw[r * tw_step]);
        }
        dft8(v);
        const int j0 = (lane - k) * 8 + k;
        wave_panel_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) panel[j0 + r * ns] = v[r];
        wave_panel_sync();
    }
}

// ---- S = stft(x): grid (gx, batch), 4 waves, one pair of frames per wave and trip
__global__ __launch_bounds__(64 * kPvWaves) void pv_stft_kernel(const float* __restrict__ pcm,
                                                                const int64_t* __restrict__ offsets,
                                                                const int32_t* __restrict__ lengths,
                                                                const int64_t* __restrict__ layout,
                                                                float* __restrict__ ws) {
    __shared__ cf s_tw[kPvN];
    __shared__ float s_win[kPvN];
    __shared__ cf s_panel[kPvWaves][kPvN];
    const int clip = blockIdx.y;
    const PvClip c = load_clip(layout, clip);
    if (c.T <= 0 || lengths[clip] != c.n) return;                          // uniform over the workgroup
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fill_tables(s_tw, s_win);
    __syncthreads();
    const float* __restrict__ x = pcm + offsets[clip];
    cf* __restrict__ spec = reinterpret_cast<cf*>(ws + c.spec);
    cf* panel = s_panel[wave];
    const bool polar = c.rate != 1.0;
    const int n_pairs = (c.T + 1) / 2;
    const int stride = gridDim.x * kPvWaves;
    const int trips = (n_pairs + stride - 1) / stride;
    // the samples of a wave's next pair are requested before the transform of the current one (HBM latency under it)
    float xa[8], xb[8];
    auto fetch = [&](int pair) {
        const int t0 = 2 * pair, t1 = t0 + 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = lane + 64 * r;
            auto sample = [&](int t) {
                if (t >= c.T) return 0.0f;
                int idx = t * kPvHop + i - kPvPad;
                if (idx < 0) idx = -idx;
                if (idx >= c.n) idx = 2 * (c.n - 1) - idx;
                return x[idx];
            };
            xa[r] = sample(t0);
            xb[r] = sample(t1);
        }
    };
    fetch(blockIdx.x * kPvWaves + wave);
    for (int it = 0; it < trips; ++it) {
        const int pair = it * stride + blockIdx.x * kPvWaves + wave;
        const int t0 = 2 * pair, t1 = t0 + 1;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = lane + 64 * r;
            panel[i] = cf{xa[r] * s_win[i], xb[r] * s_win[i]};
        }
        if (it + 1 < trips) fetch(pair + stride);
        wave_panel_sync();
        fft512_panel(panel, s_tw, lane);
        // X1[k] = (Z[k] + conj Z[N-k]) / 2, X2[k] = (Z[k] - conj Z[N-k]) / 2i
        for (int k = lane; k < kPvBins; k += 64) {
            const cf a = panel[k], b = panel[(kPvN - k) & (kPvN - 1)];
            cf x0{0.5f * (a.x + b.x), 0.5f * (a.y - b.y)}, x1{0.5f * (a.y + b.y), 0.5f * (b.x - a.x)};
            if (polar) {                                                   // what the vocoder reads: abs() and angle()
                x0 = cf{hypotf(x0.x, x0.y), atan2f(x0.y, x0.x)};
                x1 = cf{hypotf(x1.x, x1.y), atan2f(x1.y, x1.x)};
            }
            if (t0 < c.T) spec[(int64_t)t0 * kPvBins + k] = x0;
            if (t1 < c.T) spec[(int64_t)t1 * kPvBins + k] = x1;
        }
        wave_panel_sync();
    }
}

// ---- S' = phase_vocoder(S).  S arrives as (|S|, angle S) (see pv_stft_kernel), so an element costs two loads,
// a wrap and one sincos.  The phase of output frame j is a running sum over the frames before it.  A workgroup
// owns 32 bins of one clip (256-byte runs of S and S') and walks the time axis in chunks of 8 segments x 8 frames:
// every thread requests the 16 values of its segment at once, computes the (magnitude, phase increment) pairs into
// registers and their float64 sum, the segment sums go through LDS, and the thread then turns its pairs into
// mag * (cos, sin)(float32(running sum)), the running sum starting from the chunk's carry + the segments before its
// own.  S is read once, S' written once.  (Measured: 16 bins x 16 segments x 16 frames 0.271 ms, 32 x 8 x 16
// 0.258, 32 x 8 x 8 0.231, 32 x 8 x 4 0.234 ms on 256 x 5.5 s clips: shorter segments keep more waves resident.)
// (torch.cumsum sums float32 in float64 on the CPU and rounds each prefix to float32; the segmented sum differs
// from the sequential one by float64 rounding only.  cos / sin: torch takes them of the float32 prefix; here that
// float32 value is reduced to a fraction of a turn in float64 first and given to sincospif.)
constexpr int kPvSegs = 8, kPvBinsPerGroup = 32, kPvSegSteps = 8;
__global__ __launch_bounds__(kPvSegs * kPvBinsPerGroup) void pv_stretch_kernel(const int32_t* __restrict__ lengths,
                                                                              const int64_t* __restrict__ layout,
                                                                              float* __restrict__ ws) {
    __shared__ double s_sum[kPvSegs][kPvBinsPerGroup];
    const int clip = blockIdx.y;
    const PvClip c = load_clip(layout, clip);
    if (c.T <= 0 || lengths[clip] != c.n) return;                          // uniform over the workgroup
    const int kk = threadIdx.x & (kPvBinsPerGroup - 1), seg = threadIdx.x / kPvBinsPerGroup;
    const int k = blockIdx.x * kPvBinsPerGroup + kk;
    const bool live = k < kPvBins;
    const cf* __restrict__ spec = reinterpret_cast<const cf*>(ws + c.spec);
    cf* __restrict__ out = reinterpret_cast<cf*>(ws + c.stretch);
    if (c.rate == 1.0) {                                                   // phase_vocoder returns its input (raw S here)
        if (live)
            for (int t = seg; t < c.T_out; t += kPvSegs) out[(int64_t)t * kPvBins + k] = spec[(int64_t)t * kPvBins + k];
        return;
    }
    // torch.linspace(0, pi * hop, 257)[k], float32
    const float end = (float)(M_PI * kPvHop);
    const float step = end / (float)(kPvBins - 1);
    const float adv = k < kPvBins / 2 ? step * (float)k : end - step * (float)(kPvBins - 1 - k);
    const float two_pi = (float)(2.0 * M_PI);
    auto at = [&](int t) { return t < c.T ? spec[(int64_t)t * kPvBins + k] : cf{0.f, 0.f}; };   // (abs, angle); pad = 0
    double carry = live ? (double)at(0).y : 0.0;                           // phase_0 = angle(S[..., 0])
    for (int jc = 0; jc < c.T_out; jc += kPvSegs * kPvSegSteps) {          // uniform trip count
        const int j0 = jc + seg * kPvSegSteps;
        float mg[kPvSegSteps], ph[kPvSegSteps];
        double sum = 0.0;
        // all 32 loads of the segment are issued before the first use: addresses are clamped instead of branched
        // around (rows >= T are the zero padding of torch's pad(S, [0, 2]); dead lanes and steps read row 0)
        cf lo[kPvSegSteps], hi[kPvSegSteps];
        float al[kPvSegSteps];
        const int kc = live ? k : 0;
#pragma unroll
        for (int i = 0; i < kPvSegSteps; ++i) {
            const int j = min(j0 + i, c.T_out - 1);
            const float ts = (float)((double)j * c.rate);
            const int i0 = (int)ts;
            al[i] = ts - floorf(ts);
            lo[i] = spec[(int64_t)min(i0, c.T - 1) * kPvBins + kc];
            hi[i] = spec[(int64_t)min(i0 + 1, c.T - 1) * kPvBins + kc];
            if (i0 >= c.T) lo[i] = cf{0.f, 0.f};
            if (i0 + 1 >= c.T) hi[i] = cf{0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < kPvSegSteps; ++i) {
            const bool on = live && j0 + i < c.T_out;
            float d = hi[i].y - lo[i].y - adv;
            d = d - two_pi * rintf(d / two_pi);
            ph[i] = on ? d + adv : 0.f;
            mg[i] = on ? al[i] * hi[i].x + (1.0f - al[i]) * lo[i].x : 0.f;
            sum += (double)ph[i];
        }
        s_sum[seg][kk] = sum;
        __syncthreads();
        double acc = carry, total = carry;
#pragma unroll
        for (int s = 0; s < kPvSegs; ++s) {
            const double v = s_sum[s][kk];
            total += v;
            if (s < seg) acc += v;
        }
        carry = total;
#pragma unroll
        for (int i = 0; i < kPvSegSteps; ++i) {
            const int j = j0 + i;
            if (live && j < c.T_out) {
                // cos / sin of float32(acc), as torch takes them, with the reduction done here in float64 turns
                // (sincosf's own large-argument path is ~300 instructions); within 2e-7 rad of the exact value
                const double u = (double)(float)acc * (0.5 / M_PI);
                float sn, cs;
                sincospif((float)(2.0 * (u - rint(u))), &sn, &cs);
                out[(int64_t)j * kPvBins + k] = cf{mg[i] * cs, mg[i] * sn};
            }
            acc += (double)ph[i];
        }
        __syncthreads();                                                   // s_sum is rewritten by the next chunk
    }
}

// ---- y = istft(S'): inverse transforms, window, overlap-add, window envelope, trim by n_fft / 2, cut or zero-pad
// to `count` samples -- one kernel, the windowed frames never leave LDS.  A workgroup owns kPvRun hops (3072
// samples) of the output: the kPvRun + 3 frames that cover them are inverse-transformed into LDS (two frames per
// 512-point transform: Z = ext(S0) + i ext(S1), ifft(Z) = conj(fft(conj Z)) / N), then every thread gathers groups of
// four consecutive samples from the (at most four) frames over them, in ascending frame order like torch's fold,
// and divides by the summed squared window.  The three halo frames are transformed by two workgroups (+ 12 %
// transforms) in exchange for not writing and re-reading 2048 bytes per frame.
// as-constructed mode: dst = the clip itself, count = n; functional mode: dst = the wave buffer, count = len_stretch.
constexpr int kPvRun = 24, kPvSlots = kPvRun + 3;
__global__ __launch_bounds__(64 * kPvWaves, 2) void pv_synth_kernel(float* __restrict__ pcm,
                                                                    const int64_t* __restrict__ offsets,
                                                                    const int32_t* __restrict__ lengths,
                                                                    const int64_t* __restrict__ layout,
                                                                    float* __restrict__ ws, int to_wave) {
    __shared__ cf s_tw[kPvN];
    __shared__ __attribute__((aligned(16))) float s_win[kPvN];
    __shared__ cf s_panel[kPvWaves][kPvN];
    __shared__ __attribute__((aligned(16))) float s_frames[kPvSlots][kPvN];
    const int clip = blockIdx.y;
    const PvClip c = load_clip(layout, clip);
    if (c.T <= 0 || lengths[clip] != c.n) return;                          // uniform over the workgroup
    const int count = to_wave ? c.len_stretch : c.n;
    const int m0 = blockIdx.x * (kPvHop * kPvRun);
    if (m0 >= count) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    fill_tables(s_tw, s_win);
    __syncthreads();
    const cf* __restrict__ S = reinterpret_cast<const cf*>(ws + c.stretch);
    float* __restrict__ dst = to_wave ? ws + c.wave : pcm + offsets[clip];
    cf* panel = s_panel[wave];
    // sample m sits at pos = m + 256 of the untrimmed signal; frame t covers pos in [128 t, 128 t + 512)
    const int t_first = m0 / kPvHop - 1;                                   // (m0 + 256) / 128 - 3
    if (m0 < c.len_stretch) {                                              // uniform; beyond it only zeros are written
        // the bins of a wave's next pair are requested before the transform of the current one
        cf za[8], zb[8];
        auto fetch = [&](int pr) {
            const int t0 = t_first + 2 * pr, t1 = t0 + 1;
            const bool ok0 = t0 >= 0 && t0 < c.T_out, ok1 = t1 >= 0 && t1 < c.T_out && 2 * pr + 1 < kPvSlots;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = lane + 64 * r;
                const int m = i <= kPvN / 2 ? i : kPvN - i;
                za[r] = ok0 ? S[(int64_t)t0 * kPvBins + m] : cf{0.f, 0.f};
                zb[r] = ok1 ? S[(int64_t)t1 * kPvBins + m] : cf{0.f, 0.f};
            }
        };
        fetch(wave);
        for (int pr = wave; 2 * pr < kPvSlots; pr += kPvWaves) {
            const int t0 = t_first + 2 * pr, t1 = t0 + 1;
            const bool ok0 = t0 >= 0 && t0 < c.T_out, ok1 = t1 >= 0 && t1 < c.T_out && 2 * pr + 1 < kPvSlots;
            const bool any = ok0 || ok1;                                   // wave-uniform
            if (any) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int i = lane + 64 * r;
                    const bool edge = i == 0 || i == kPvN / 2;           // c2r ignores the imaginary part of DC / Nyquist
                    const float sgn = i <= kPvN / 2 ? 1.0f : -1.0f;
                    const cf a{za[r].x, edge ? 0.0f : za[r].y * sgn}, b{zb[r].x, edge ? 0.0f : zb[r].y * sgn};
                    panel[i] = cf{a.x - b.y, -(a.y + b.x)};
                }
            }
            if (2 * (pr + kPvWaves) < kPvSlots) fetch(pr + kPvWaves);
            if (!any) continue;
            wave_panel_sync();
            fft512_panel(panel, s_tw, lane);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = lane + 64 * r;
                const cf v = panel[i];
                const float w = s_win[i];
                if (ok0) s_frames[2 * pr][i] = (v.x * (1.0f / kPvN)) * w;
                if (ok1) s_frames[2 * pr + 1][i] = (-v.y * (1.0f / kPvN)) * w;
            }
            wave_panel_sync();
        }
    }
    __syncthreads();
    const bool dst_vec = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int g = threadIdx.x; g < kPvHop * kPvRun / 4; g += blockDim.x) {
        const int m = m0 + 4 * g;
        if (m >= count) break;
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (m < c.len_stretch) {
            const int pos = m + kPvPad;
            const int t_lo = pos >= kPvN ? (pos - kPvN) / kPvHop + 1 : 0;
            int t_hi = pos / kPvHop;
            if (t_hi > c.T_out - 1) t_hi = c.T_out - 1;
            float sum[4] = {0.f, 0.f, 0.f, 0.f}, env[4] = {0.f, 0.f, 0.f, 0.f};
            for (int t = t_lo; t <= t_hi; ++t) {
                const int o = pos - t * kPvHop;
                const float4 f = *reinterpret_cast<const float4*>(&s_frames[t - t_first][o]);
                const float4 w = *reinterpret_cast<const float4*>(s_win + o);
                sum[0] = sum[0] + f.x;
                sum[1] = sum[1] + f.y;
                sum[2] = sum[2] + f.z;
                sum[3] = sum[3] + f.w;
                env[0] = env[0] + w.x * w.x;
                env[1] = env[1] + w.y * w.y;
                env[2] = env[2] + w.z * w.z;
                env[3] = env[3] + w.w * w.w;
            }
            if (t_hi >= t_lo) {
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = m + i < c.len_stretch ? sum[i] / env[i] : 0.0f;
            }
        }
        if (dst_vec && m + 4 <= count) {
            *reinterpret_cast<float4*>(dst + m) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m + i < count) dst[m + i] = y[i];
        }
    }
}

// ---- functional mode: sinc resample of the stretched waveform, taps evaluated per output sample in float64
// (torchaudio builds its table in float64 and rounds it to float32), crop / zero-pad to n, into the clip.
__global__ __launch_bounds__(256) void pv_resample_kernel(float* __restrict__ pcm, const int64_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ lengths,
                                                          const int64_t* __restrict__ layout,
                                                          const float* __restrict__ ws) {
    const int clip = blockIdx.y;
    const PvClip c = load_clip(layout, clip);
    if (c.T <= 0 || lengths[clip] != c.n) return;
    const float* __restrict__ src = ws + c.wave;
    float* __restrict__ dst = pcm + offsets[clip];
    if (c.orig == c.neu) {
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < c.n; j += gridDim.x * blockDim.x)
            dst[j] = j < c.len_stretch ? src[j] : 0.0f;
        return;
    }
    const double base = (double)(c.orig < c.neu ? c.orig : c.neu) * 0.99;
    const int width = (int)ceil(6.0 * (double)c.orig / base);
    const double scale = base / (double)c.orig, inv_orig = 1.0 / (double)c.orig;
    const int64_t n_out = ((int64_t)c.neu * c.len_stretch + c.orig - 1) / c.orig;
    // consecutive taps are base / orig apart in t: sin(pi t) and cos(pi t / 12) advance by fixed rotations (float64;
    // 2 * width + 3 steps), so an output costs two sincos instead of two per tap
    double rot_s, rot_c, rot_sw, rot_cw;
    sincos(M_PI * scale, &rot_s, &rot_c);
    sincos(M_PI * scale / 12.0, &rot_sw, &rot_cw);
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < c.n; j += gridDim.x * blockDim.x) {
        float y = 0.0f;
        if (j < n_out) {
            const int blk = j / c.neu, ph = j - blk * c.neu;
            const int centre = (int)(((int64_t)ph * c.orig) / c.neu);     // source offset nearest t = 0
            const int64_t s0 = (int64_t)blk * c.orig;
            // torch: arange(0, -new, -1) / new is an int64 / int true division -> a float32 quotient
            const double phase = (double)__fdiv_rn((float)(-ph), (float)c.neu);
            const int kk0 = centre - width - 1;
            double sn, cs, sw, cw;
            {
                const double t0 = (phase + (double)kk0 * inv_orig) * base;
                sincos(t0 * M_PI, &sn, &cs);
                sincos(t0 * M_PI / 12.0, &sw, &cw);
            }
            for (int d = 0; d <= 2 * width + 2; ++d) {
                const int kk = kk0 + d;                                    // tap position relative to the block start
                const int64_t s = s0 + kk;
                const double t = (phase + (double)kk * inv_orig) * base;
                // outside the table row / zero padding / beyond the window (|t| >= 6: cos^2(pi/2) = 4e-33 in the table)
                if (kk >= -width && kk < width + c.orig && s >= 0 && s < c.len_stretch && fabs(t) < 6.0) {
                    // sin and cos come from the float64 rotations; the quotient and the products are float32 (the
                    // table entry is a float32 anyway: this costs one more rounding, not a float64 division per tap)
                    const float tp = (float)(t * M_PI), cwf = (float)cw;
                    const float sinc = tp == 0.0f ? 1.0f : (float)sn / tp;
                    y = y + src[s] * (sinc * (cwf * cwf) * (float)scale);
                }
                const double sn2 = sn * rot_c + cs * rot_s, sw2 = sw * rot_cw + cw * rot_sw;
                cs = cs * rot_c - sn * rot_s;
                cw = cw * rot_cw - sw * rot_sw;
                sn = sn2;
                sw = sw2;
            }
        }
        dst[j] = y;
    }
}

}  // namespace sttfe

using namespace sttfe;

namespace {
constexpr int64_t kAlignFloats = 64;
int64_t align_up(int64_t v) { return (v + kAlignFloats - 1) / kAlignFloats * kAlignFloats; }
int gcd_int(int a, int b) {
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    return a;
}
}  // namespace

extern "C" int sttfe_pitch_shift_plan(const int32_t* lengths, const int32_t* n_steps, int batch, int sample_rate,
                                      int mode, int64_t* layout, size_t* workspace_bytes) {
    if (batch < 0 || sample_rate <= 0 || !workspace_bytes) return STTFE_E_INVALID_ARG;
    if (mode != STTFE_PITCH_AS_CONSTRUCTED && mode != STTFE_PITCH_FUNCTIONAL) return STTFE_E_INVALID_ARG;
    if (batch > 0 && (!lengths || !n_steps || !layout)) return STTFE_E_INVALID_ARG;
    int64_t cursor = 0;
    for (int b = 0; b < batch; ++b) {
        int64_t* r = layout + (int64_t)b * kLayoutWords;
        std::memset(r, 0, sizeof(int64_t) * kLayoutWords);
        if (n_steps[b] == STTFE_PITCH_NONE) continue;
        const int n = lengths[b];
        if (n <= kPvPad) return STTFE_E_INVALID_ARG;                       // reflect padding needs n > n_fft / 2 (torch.stft raises)
        if (n_steps[b] < -48 || n_steps[b] > 48) return STTFE_E_UNSUPPORTED;
        const double rate = std::pow(2.0, -(double)n_steps[b] / 12.0);
        const int64_t T = 1 + n / kPvHop;
        const int64_t T_out = rate == 1.0 ? T : (int64_t)std::ceil((double)T / rate);   // len(arange(0, T, rate))
        const int64_t len_stretch = (int64_t)std::nearbyint((double)n / rate);           // Python round(): half to even
        if (T_out > 0x7fffffff / kPvN || len_stretch > 0x7fffffff) return STTFE_E_UNSUPPORTED;
        int orig = (int)((double)sample_rate / rate), neu = sample_rate;
        const int g = gcd_int(orig, neu);
        orig /= g;
        neu /= g;
        r[0] = cursor;
        cursor += align_up(T * kPvBins * 2);
        r[1] = cursor;
        cursor += align_up(T_out * kPvBins * 2);
        r[2] = cursor;                                                     // (no frame buffer: pv_synth_kernel keeps them in LDS)
        r[3] = cursor;
        if (mode == STTFE_PITCH_FUNCTIONAL) cursor += align_up(len_stretch);
        r[4] = T | (T_out << 32);
        r[5] = (int64_t)n | (len_stretch << 32);
        std::memcpy(&r[6], &rate, sizeof(double));
        r[7] = (int64_t)orig | ((int64_t)neu << 32);
    }
    *workspace_bytes = (size_t)cursor * sizeof(float);
    return STTFE_OK;
}

extern "C" int sttfe_pitch_shift(float* pcm, const int64_t* offsets, const int32_t* lengths, int batch,
                                 const int64_t* layout, const int64_t* dev_layout, int mode, void* workspace,
                                 size_t workspace_bytes, void* stream_) {
    if (batch < 0) return STTFE_E_INVALID_ARG;
    if (mode != STTFE_PITCH_AS_CONSTRUCTED && mode != STTFE_PITCH_FUNCTIONAL) return STTFE_E_INVALID_ARG;
    if (batch == 0) return STTFE_OK;
    if (!pcm || !offsets || !lengths || !layout || !dev_layout || !workspace) return STTFE_E_INVALID_ARG;
    if (batch > 65535) return STTFE_E_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(workspace) % 16 != 0) return STTFE_E_INVALID_ARG;
    // re-derive the extent of every row from the host layout: nothing on the device is trusted for sizes
    int64_t need = 0;
    int max_T = 0, max_T_out = 0, max_count = 0, active = 0;
    for (int b = 0; b < batch; ++b) {
        const int64_t* r = layout + (int64_t)b * kLayoutWords;
        const int64_t T = r[4] & 0xffffffff, T_out = r[4] >> 32, n = r[5] & 0xffffffff, ls = r[5] >> 32;
        if (T == 0) continue;
        if (T < 0 || T_out <= 0 || n <= kPvPad || ls < 0 || T != 1 + n / kPvHop) return STTFE_E_INVALID_ARG;
        if (r[0] != need) return STTFE_E_INVALID_ARG;
        need += align_up(T * kPvBins * 2);
        if (r[1] != need) return STTFE_E_INVALID_ARG;
        need += align_up(T_out * kPvBins * 2);
        if (r[2] != need || r[3] != need) return STTFE_E_INVALID_ARG;
        if (mode == STTFE_PITCH_FUNCTIONAL) need += align_up(ls);
        if ((r[7] & 0xffffffff) <= 0 || (r[7] >> 32) <= 0 || (r[7] & 0xffffffff) > (1 << 24) || (r[7] >> 32) > (1 << 24))
            return STTFE_E_INVALID_ARG;
        double rate;
        std::memcpy(&rate, &r[6], sizeof(double));
        if (!(rate >= 1.0 / 16.0 && rate <= 16.0)) return STTFE_E_INVALID_ARG;             // also rejects NaN
        if (T_out != (rate == 1.0 ? T : (int64_t)std::ceil((double)T / rate))) return STTFE_E_INVALID_ARG;
        ++active;
        if (T > max_T) max_T = (int)T;
        if (T_out > max_T_out) max_T_out = (int)T_out;
        const int64_t cnt = n > ls ? n : ls;
        if (cnt > max_count) max_count = (int)cnt;
    }
    if ((size_t)need * sizeof(float) > workspace_bytes) return STTFE_E_WORKSPACE;
    if (active == 0) return STTFE_OK;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    (void)hipGetLastError();
    float* ws = static_cast<float*>(workspace);
    {
        ScopedTiming t(kTimePitchShift, stream);
        auto pair_blocks = [](int frames) {
            int g = ((frames + 1) / 2 + kPvWaves * 4 - 1) / (kPvWaves * 4);      // about four pairs per wave
            return g < 1 ? 1 : (g > 1024 ? 1024 : g);
        };
        hipLaunchKernelGGL(pv_stft_kernel, dim3(pair_blocks(max_T), batch), dim3(64 * kPvWaves), 0, stream, pcm,
                           offsets, lengths, dev_layout, ws);
        hipLaunchKernelGGL(pv_stretch_kernel, dim3((kPvBins + kPvBinsPerGroup - 1) / kPvBinsPerGroup, batch),
                           dim3(kPvSegs * kPvBinsPerGroup), 0, stream, lengths,
                           dev_layout, ws);
        const int to_wave = mode == STTFE_PITCH_FUNCTIONAL;
        const int run = kPvHop * kPvRun;
        hipLaunchKernelGGL(pv_synth_kernel, dim3((max_count + run - 1) / run, batch), dim3(64 * kPvWaves), 0, stream,
                           pcm, offsets, lengths, dev_layout, ws, to_wave);
        int bx = (max_count + 256 * 4 - 1) / (256 * 4);
        bx = bx < 1 ? 1 : (bx > 1024 ? 1024 : bx);
        if (to_wave)
            hipLaunchKernelGGL(pv_resample_kernel, dim3(bx, batch), dim3(256), 0, stream, pcm, offsets, lengths,
                               dev_layout, ws);
    }
    if (hip_failed(hipGetLastError(), "pitch shift kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}
