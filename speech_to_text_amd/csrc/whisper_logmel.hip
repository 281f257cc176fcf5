// Whisper log-mel front-end for gfx950 (MI355X): framing + Hann + 400-point STFT + mel + log10,
// then a per-clip clamp(max - 8) / (x + 4) / 4 (+ optional SpecAugment masks) fix-up pass.
//
// Replaces HF WhisperFeatureExtractor._torch_extract_fbank_features
// (HF:models/whisper/feature_extraction_whisper.py:135-168) as reached from the reference at
// src/data.py:338-343, inference.py:148-152 and src/processors.py:180-221.
//
// Kernel 1 (stft_mel): one 512-thread workgroup (8 waves) per 48-frame tile of one clip.
//   * A wave owns 6 frames = 3 pairs; two real frames a, b are packed as z = a + i b and
//     transformed by ONE complex 400-point DFT, factored 20 x 20: lane (p, j) of a pair first runs
//     a register-resident 20-point DFT over n1 of z[20 n1 + j] (fft_core.h, twiddle-free 4 x 5
//     prime-factor form), multiplies by W_400^(j k1), exchanges through a wave-private,
//     bank-conflict-free LDS slab ([pair][k1][n2], row stride 21, pair stride 420 float2), and runs
//     the second 20-point DFT over n2.  Lane j then holds Z[j + 20 k2]; the spectra of a and b are
//     separated with one cross-lane exchange (lane j <-> lane 20 - j) and only bins 0..200 kept.
//   * |A|^2, |B|^2 go to a [frame][bin] LDS panel (row stride 202 floats: conflict-free for both
//     the scattered writes and the MFMA operand reads) that aliases the FFT slabs.
//   * The mel projection is a banded product on the matrix cores: exact-f32
//     v_mfma_f32_16x16x4_f32 tiles (16 mel rows x 16 frames), only over the bins each 16-row band
//     of the triangular bank touches (57 of 255 k-steps for 80 mels); (mel tile, frame tile) units
//     are spread over the 8 waves by a host-built plan (tables.cpp).
//   * Epilogue per tile: log10(max(mel, 1e-10)) -> (x + 4) / 4 stored unclamped; the running
//     per-clip max of log10 goes to a per-clip key with ONE atomicMax per wave.
//   Tiles that lie entirely in the zero padding skip the FFT and store the constant.
// Kernel 2 (clamp_mask): streams the clip's output once, applies max(y, (max - 8 + 4) / 4) and the
//   SpecAugment masks.  Launched per chunk of clips small enough that kernel 1's output is still
//   in the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sttfe.h"
#include "fft_core.h"
#include "tables.h"
#include "timing.h"

namespace sttfe {

constexpr int kThreads = 512;
constexpr int kTileFrames = 48;
constexpr int kFramesPerWave = 6;
constexpr int kHop = 160;
constexpr int kNfft = 400;
constexpr int kMaxSamples = 480000;          // 30 s at 16 kHz (HF:...whisper.py:91)
constexpr int kRowStride = 21;               // float2 units, [k1][n2] rows
constexpr int kPairStride = 420;             // float2 units per frame pair (== 4 mod 32)
constexpr int kSlabPerWave = 3 * kPairStride;
constexpr int kPStride = 202;                // floats per frame row of the power panel (== 10 mod 16)
constexpr int kLdsBytes = kWavesPerBlock * kSlabPerWave * 8;     // 80640 B -> 2 workgroups per CU
static_assert(kTileFrames * kPStride * 4 <= kLdsBytes, "power panel must fit inside the FFT slabs");

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned float_to_key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // order-preserving float -> uint
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Orders this wave's LDS writes before its later LDS reads (lanes of one wave run in lockstep and
// the LDS queue is in-order per wave; this only stops the compiler from moving accesses across).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(kThreads, 4) void whisper_stft_mel_kernel(
    const float* __restrict__ pcm, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ lengths, int clip0, int n_mels, int n_frames,
    const WhisperTablesHeader* __restrict__ tab, float* __restrict__ out,
    unsigned* __restrict__ max_keys, int tiles_per_block) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
    cf* slab_all = reinterpret_cast<cf*>(smem);
    float* panel = reinterpret_cast<float*>(smem);

    const int clip = clip0 + blockIdx.y;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane / 20;                 // pair within the wave (3 = idle lanes 60..63)
    const int j = lane - p * 20;
    const bool active = lane < 60;

    const float* __restrict__ x = pcm + offsets[clip];
    int n_valid = lengths[clip];
    n_valid = n_valid < 0 ? 0 : (n_valid > kMaxSamples ? kMaxSamples : n_valid);

    const float* __restrict__ afrag = reinterpret_cast<const float*>(tab + 1);
    const MelPlan& plan = tab->plan;

    // Per-lane constants (window column j, twiddles W_400^(j k1)) are re-read from the L1-resident
    // tables every tile instead of living in 60 VGPRs: the loads are lane-contiguous (80 / 160 B).
    const float* __restrict__ win_col = tab->window + j;
    const cf* __restrict__ tw_col = reinterpret_cast<const cf*>(tab->twiddle) + j;
    const int partner = active ? (p * 20 + ((20 - j) % 20)) : lane;
    cf* slab = slab_all + wave * kSlabPerWave + (active ? p : 0) * kPairStride;

    float run_max = -3.0e38f;
    const int n_tiles = (n_frames + kTileFrames - 1) / kTileFrames;
    const int tile_begin = blockIdx.x * tiles_per_block;
    const int tile_end = min(tile_begin + tiles_per_block, n_tiles);

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int t0 = tile * kTileFrames;
        const int first_sample = kHop * t0 - kNfft / 2;
        float* __restrict__ out_clip = out + (size_t)clip * n_mels * n_frames;

        if (first_sample >= n_valid) {
            // Whole tile reads only zero padding: mel = 0 -> log10(1e-10) = -10 -> (x + 4) / 4.
            const int width = min(kTileFrames, n_frames - t0);
            for (int e = tid; e < n_mels * kTileFrames; e += kThreads) {
                const int m = e / kTileFrames, f = e - m * kTileFrames;
                if (f < width) out_clip[(size_t)m * n_frames + t0 + f] = -1.5f;
            }
            run_max = fmaxf(run_max, -10.0f);
            continue;
        }

        // ---------------- load 28 samples per lane (frame a = n1 0..19, frame b = n1 8..27) ----
        const int fa = t0 + wave * kFramesPerWave + 2 * (active ? p : 0);
        const int base = kHop * fa - kNfft / 2 + j;
        float raw[28];
        const bool interior = (first_sample >= 0) &&
                              (kHop * (t0 + kTileFrames - 1) - kNfft / 2 + kNfft <= n_valid);
        if (interior) {
#pragma unroll
            for (int m = 0; m < 28; ++m) raw[m] = x[base + 20 * m];
        } else {
#pragma unroll
            for (int m = 0; m < 28; ++m) {
                int idx = base + 20 * m;
                idx = idx < 0 ? -idx : idx;                       // reflect padding (center=True) ...
                idx = idx >= kMaxSamples ? 2 * (kMaxSamples - 1) - idx : idx;   // ... of the 30 s buffer
                raw[m] = (idx < n_valid) ? x[idx] : 0.0f;         // zero padding to 30 s
            }
        }

        // ---------------- stage 1: 20-point DFT over n1, twiddle, to the slab -------------------
        cf v[20], y[20];
        int keep = 0;
        asm volatile("" : "+v"(keep));    // keeps the table loads inside the tile loop (no hoist -> no spills)
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            const float w = win_col[20 * i + keep];
            v[i] = cf{raw[i] * w, raw[i + 8] * w};
        }
        dft20(v, y);
        if (active) {
            slab[j] = y[0];
#pragma unroll
            for (int k1 = 1; k1 < 20; ++k1) slab[k1 * kRowStride + j] = cmul(y[k1], tw_col[20 * k1 + keep]);
        }
        wave_lds_sync();

        // ---------------- stage 2: 20-point DFT over n2 --------------------------------------
#pragma unroll
        for (int n2 = 0; n2 < 20; ++n2) v[n2] = slab[j * kRowStride + n2];
        dft20(v, y);          // y[k2] = Z[j + 20 k2]

        // ---------------- separate the two real spectra, power, bins 0..200 ----------------------
        float pa[11], pb[11];
#pragma unroll
        for (int k2 = 0; k2 < 10; ++k2) {
            float qr = __shfl(y[19 - k2].x, partner, 64);
            float qi = __shfl(y[19 - k2].y, partner, 64);
            if (j == 0) {                                         // own Z[400 - 20 k2]
                qr = (k2 == 0) ? y[0].x : y[20 - k2].x;
                qi = (k2 == 0) ? y[0].y : y[20 - k2].y;
            }
            const float sr = y[k2].x + qr, dr = y[k2].x - qr;
            const float si = y[k2].y + qi, di = y[k2].y - qi;
            pa[k2] = sr * sr + di * di;                           // 4 |A[k]|^2
            pb[k2] = si * si + dr * dr;                           // 4 |B[k]|^2
        }
        pa[10] = 4.0f * y[10].x * y[10].x;                        // bin 200 (lane j == 0 only)
        pb[10] = 4.0f * y[10].y * y[10].y;

        __syncthreads();      // every wave is done reading its slab: the panel may overwrite them
        if (active) {
            float* row = panel + (wave * kFramesPerWave + 2 * p) * kPStride + j;
#pragma unroll
            for (int k2 = 0; k2 < 10; ++k2) {
                row[20 * k2] = pa[k2];
                row[kPStride + 20 * k2] = pb[k2];
            }
            if (j == 0) {
                row[200] = pa[10];
                row[kPStride + 200] = pb[10];
            }
        }
        __syncthreads();

        // ---------------- banded mel projection on the matrix cores + log epilogue --------------
        const int nu = plan.wave_nunits[wave];
        for (int u = 0; u < nu; ++u) {
            const int code = plan.wave_units[wave][u];
            const int mt = code & 0xff, ft = code >> 8;
            const int ks = plan.kstart[mt], ns = plan.nsteps[mt];
            const float* __restrict__ a_ptr = afrag + plan.aoff[mt] + lane;
            const float* b_ptr = panel + (ft * 16 + (lane & 15)) * kPStride + ks + (lane >> 4);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s = 0; s < ns; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_ptr[s * 64], b_ptr[4 * s], acc, 0, 0, 0);
            const int frame = t0 + ft * 16 + (lane & 15);
            const int mel0 = mt * 16 + (lane >> 4) * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mel = mel0 + r;
                if (mel < n_mels && frame < n_frames) {
                    const float lg = __log2f(fmaxf(acc[r], 1e-10f)) * 0.30102999566398120f;
                    run_max = fmaxf(run_max, lg);
                    out_clip[(size_t)mel * n_frames + frame] = (lg + 4.0f) * 0.25f;
                }
            }
        }
        __syncthreads();      // panel reads done before the next tile's slab writes
    }

    // one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) run_max = fmaxf(run_max, __shfl_xor(run_max, off, 64));
    if (lane == 0 && tile_begin < tile_end) atomicMax(&max_keys[clip], float_to_key(run_max));
}

// y <- max(y, ((max - 8) + 4) / 4); SpecAugment masks -> 0.0.  One workgroup column per clip.
__global__ __launch_bounds__(256) void whisper_clamp_mask_kernel(
    float* __restrict__ out, const unsigned* __restrict__ max_keys, int clip0, int n_mels,
    int n_frames, const int32_t* __restrict__ mask_tbl, int n_masks) {
    const int clip = clip0 + blockIdx.y;
    const float lmax = key_to_float(max_keys[clip]);
    const float floor_y = ((lmax - 8.0f) + 4.0f) * 0.25f;
    float* __restrict__ o = out + (size_t)clip * n_mels * n_frames;
    const int total = n_mels * n_frames;
    int m_axis[8], m_s[8], m_e[8];
    const int nm = mask_tbl ? (n_masks > 8 ? 8 : n_masks) : 0;
    for (int i = 0; i < nm; ++i) {
        const int32_t* m = mask_tbl + ((size_t)clip * n_masks + i) * 3;
        m_axis[i] = m[0];
        m_s[i] = m[1];
        m_e[i] = m[2];
    }
    const int stride = gridDim.x * blockDim.x * 4;
    for (int e = (blockIdx.x * blockDim.x + threadIdx.x) * 4; e < total; e += stride) {
        // rows are n_frames long and n_frames % 4 == 0 is required by the host wrapper, so the
        // four elements share a mel row and the clip base is 16-byte aligned.
        f32x4 v = *reinterpret_cast<f32x4*>(o + e);
        f32x4 w;
        const int mel = e / n_frames, fr = e - mel * n_frames;
        bool changed = false;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float y = fmaxf(v[c], floor_y);
            for (int i = 0; i < nm; ++i) {
                const int idx = (m_axis[i] == STTFE_AXIS_TIME) ? fr + c : mel;
                if (idx >= m_s[i] && idx < m_e[i]) y = 0.0f;
            }
            changed |= (y != v[c]);
            w[c] = y;
        }
        if (changed) *reinterpret_cast<f32x4*>(o + e) = w;
    }
}

}  // namespace sttfe

using namespace sttfe;

extern "C" int sttfe_whisper_workspace_bytes(int batch, size_t* bytes) {
    if (batch < 0 || !bytes) return STTFE_E_INVALID_ARG;
    *bytes = ((size_t)batch * sizeof(unsigned) + 255) / 256 * 256;
    return STTFE_OK;
}

// Clips per launch pair: kernel 1's output for a chunk (<= 96 x 1.5 MB) should still sit in the
// 256 MiB Infinity Cache when kernel 2 re-reads it.
static int g_whisper_chunk = 96;
extern "C" int sttfe_set_whisper_chunk(int clips) {
    if (clips < 1) return STTFE_E_INVALID_ARG;
    g_whisper_chunk = clips;
    return STTFE_OK;
}

extern "C" int sttfe_whisper_logmel(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                    int batch, int n_mels, int n_frames, const void* dev_tables,
                                    float* out, const int32_t* mask_tbl, int n_masks,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
    if (batch < 0 || n_mels < 1 || n_mels > 128 || n_frames < 1 || (n_frames & 3) || n_masks < 0 || n_masks > 8)
        return STTFE_E_INVALID_ARG;
    if (batch == 0) return STTFE_OK;
    if (!pcm || !offsets || !lengths || !dev_tables || !out || !workspace) return STTFE_E_INVALID_ARG;
    if ((size_t)kHop * (size_t)n_frames > (size_t)kMaxSamples) return STTFE_E_UNSUPPORTED;
    size_t need = 0;
    sttfe_whisper_workspace_bytes(batch, &need);
    if (workspace_bytes < need) return STTFE_E_WORKSPACE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    unsigned* keys = reinterpret_cast<unsigned*>(workspace);
    (void)hipGetLastError();      // drop any stale error left by earlier, unrelated HIP calls of this thread
    if (hip_failed(hipMemsetAsync(keys, 0, need, stream), "hipMemsetAsync(max keys)")) return STTFE_E_LAUNCH;

    const int n_tiles = (n_frames + kTileFrames - 1) / kTileFrames;
    const int tiles_per_block = 4;
    const int blocks_x = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    const auto* tab = reinterpret_cast<const WhisperTablesHeader*>(dev_tables);
    for (int c0 = 0; c0 < batch; c0 += g_whisper_chunk) {
        const int nc = (batch - c0 < g_whisper_chunk) ? batch - c0 : g_whisper_chunk;
        {
            ScopedTiming t(kTimeWhisperStftMel, stream);
            hipLaunchKernelGGL(whisper_stft_mel_kernel, dim3(blocks_x, nc), dim3(kThreads), 0, stream,
                               pcm, offsets, lengths, c0, n_mels, n_frames, tab, out, keys, tiles_per_block);
        }
        const int total4 = n_mels * n_frames / 4;
        int bx = (total4 + 255) / 256;
        if (bx > 32) bx = 32;
        {
            ScopedTiming t(kTimeWhisperClamp, stream);
            hipLaunchKernelGGL(whisper_clamp_mask_kernel, dim3(bx, nc), dim3(256), 0, stream, out, keys, c0,
                               n_mels, n_frames, mask_tbl, n_masks);
        }
    }
    if (hip_failed(hipGetLastError(), "whisper kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}
