// Speech2Text front-end for gfx950 (MI355X): Kaldi-style 80-bin log-mel fbank + utterance CMVN.
//
// Replaces HF Speech2TextFeatureExtractor._extract_fbank_features + utterance_cmvn
// (HF:models/speech_to_text/feature_extraction_speech_to_text.py:104-163, framing loop
// HF:audio_utils.py:953-1017) as reached from the reference at src/data.py:346-358, plus the
// zero padding / attention mask / transpose of the collator (src/data.py:468-496).
//
// Kernel 1 (kaldi_fbank): persistent 256-thread workgroups (4 waves) walking (clip, run of 32-frame tiles) items.
//   * snip-edges frames of 400 samples, hop 160; per frame: remove DC, pre-emphasis 0.97, Povey
//     window, zero-pad to 512 (all fused into the load stage; the x * 2^15 scaling is folded into
//     the mel bank as 2^30).
//   * a wave owns 8 frames = 4 pairs; two real frames are packed into one complex 512-point DFT,
//     factored 16 x 32: stage 1 (two rounds of 2 pairs x 32 lanes) = 16-point DFT over n1 of
//     z[32 n1 + n2], twiddle W_512^(n2 k1), conflict-free LDS slab [pair][k1][n2] (row stride 33,
//     pair stride 528 float2); stage 2 (4 pairs x 16 lanes) = 32-point DFT over n2.  Lane k1 then
//     holds Z[k1 + 16 k2]; spectra separated with one cross-lane exchange (k1 <-> 16 - k1).
//   * |.|^2 for bins 0..256 -> [frame][bin] LDS panel (row stride 258) aliasing the slabs.
//   * banded mel projection on v_mfma_f32_16x16x4_f32 (frames as rows, mel bins as columns, so the
//     (B, T, 80) time-major output is written in 64-byte runs), ln(max(mel, eps)) epilogue, raw
//     log-mel stored; per-clip per-bin sum / sum of squares accumulated with f64 atomics.
// Kernel 2 (kaldi_cmvn): (x - mean) / std per clip and bin over the valid frames, padded rows <- 0,
//   SpecAugment masks <- 0, attention mask written (int32 or int64).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sttfe.h"
#include "fft_core.h"
#include "tables.h"
#include "timing.h"

namespace sttfe {

namespace kaldi {
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTileFrames = 32;            // 4 waves x 8 frames = 2 MFMA frame tiles
constexpr int kFramesPerWave = 8;
constexpr int kHop = 160;
constexpr int kFrame = 400;
constexpr int kRowStride = 33;             // float2 units, [k1][n2] rows
constexpr int kPairStride = 528;           // float2 units per pair (== 16 mod 32)
constexpr int kSlabFloats = 4 * kPairStride * 2 + 16; // 4240 floats per wave
constexpr int kPStride = 258;              // floats per frame row of the power panel (== 2 mod 32)
static_assert(kFramesPerWave * kPStride <= kSlabFloats, "a wave's panel rows must fit inside its own slab");
static_assert((kSlabFloats - kFramesPerWave * kPStride) % 32 == 0, "panel rows keep one bank pattern across waves");
constexpr int kWinOff = kWaves * kSlabFloats;          // float offsets of the tables inside the LDS array
constexpr int kTwOff = kWinOff + 400;
constexpr int kAfragOff = kTwOff + 1024;
constexpr int kMaxAfragFloats = 0;         // the mel weights are read from global memory (L1 / L2 resident)
constexpr int kDescOff = kAfragOff + kMaxAfragFloats;  // per wave: 8 x {kstart, nsteps, aoff, mt | ft << 8}, then the unit counts
constexpr int kLdsFloats = kDescOff + kWaves * kMaxUnitsPerWave * 4 + kWaves;     // 74 KB: two workgroups per CU
constexpr int kUnitsCap = 4;               // units one wave owns: ceil(8 mel tiles x 2 frame tiles / 4 waves) at most
constexpr float kMelFloor = 1.1920928955078125e-07f;
}  // namespace kaldi

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int int4k __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_sync2() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int kaldi_num_frames(int n, int max_frames, int t_max) {
    int t = n >= kaldi::kFrame ? 1 + (n - kaldi::kFrame) / kaldi::kHop : 0;
    if (max_frames > 0 && t > max_frames) t = max_frames;
    return t > t_max ? t_max : t;
}

// Persistent 256-thread workgroups (4 waves, <= 256 VGPRs, 74 KB LDS -> two per CU, grid = 2 x CUs).  A workgroup
// stages window, twiddles and the per-wave unit plan in LDS once and then takes work items -- (clip, run of
// consecutive 32-frame tiles), clip-major -- from a ticket counter in the workspace: ragged batches (config 2:
// 2-10 s clips) are spread evenly over the chip instead of the longest clip setting the time (0.141 -> 0.114 ms).
// The banded mel weights (MFMA B-operand order) are read from global memory (L1 / L2 resident); the CMVN sums of
// an item are kept in registers (float64) and reach memory with one atomic per mel bin and wave at the end of the
// item.  The loop ends after at most n_items tickets; all threads of a workgroup take the same path.
// (Requesting the next item's ticket, length and first PCM tile an item ahead, as the Whisper kernel does, needs
// one of the two PCM passes to be fetched late for register room and was measured slower: 0.127 ms.)
__global__ __launch_bounds__(kaldi::kThreads, 2) void kaldi_fbank_kernel(
    const float* __restrict__ pcm, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ lengths, int n_mels, int t_max, int max_frames,
    const KaldiTablesHeader* __restrict__ tab, float* __restrict__ out, double* __restrict__ stats,
    int tiles_per_block, int n_runs, int n_items, unsigned* __restrict__ ticket) {
    using namespace kaldi;
    __shared__ __attribute__((aligned(16))) float smem[kLdsFloats];
    __shared__ unsigned s_next[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int item = blockIdx.x;
    if (item >= n_items) return;
    // per-item state (set at the top of the item loop; the lambdas below read it by reference)
    const float* __restrict__ x = pcm;
    int n_valid = 0;

    // stage-1 lane roles: half-wave h = pair within the round, n2 = column
    const int h = lane >> 5, n2 = lane & 31;
    // PCM of one pass (2 frame pairs per wave): x[n] and x[n - 1] (pre-emphasis) for frame a = rows 0..12 and
    // frame b = rows 5..17 of x[base + 32 m].  Requested a whole tile ahead, so that the HBM latency lies under
    // the mel phase of the previous tile (and, for the first tile, under the table staging).
    auto load_pass = [&](int t0, int r, float (&raw)[18], float (&prv)[18]) {
        const int pair = 2 * r + h;
        const int fa = t0 + wave * kFramesPerWave + 2 * pair;
        const int base = kHop * fa + n2;
        // last sample any lane of this wave may touch (frame b of the last pair, m = 17, n2 = 31)
        const bool interior = (kHop * (t0 + wave * kFramesPerWave + kFramesPerWave - 1) + kFrame + 16 <= n_valid);
        if (interior) {
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                const int idx = base + 32 * m;
                raw[m] = x[idx];
                prv[m] = x[idx > 0 ? idx - 1 : 0];
            }
        } else {
#pragma unroll
            for (int m = 0; m < 18; ++m) {
                const int idx = base + 32 * m;
                raw[m] = (idx < n_valid) ? x[idx] : 0.0f;
                prv[m] = (idx - 1 < n_valid) ? x[idx > 0 ? idx - 1 : 0] : 0.0f;
            }
        }
    };
    float raw0[18], prv0[18], raw1[18], prv1[18];

    // The pad cells of the panel rows are read by zero-weight MFMA steps and must be finite.  A panel row aliases
    // the wave's exchange slab, whose cells are all rewritten with finite values by every tile except the pad
    // column of its [k1][n2] rows (33rd complex of each) and the tail of the region: zero exactly those, once.
    for (int i = tid; i < kWaves * 4 * 16; i += kThreads) {
        const int w = i >> 6, p = (i >> 4) & 3, row = i & 15;
        float* cell = smem + w * kSlabFloats + 2 * (p * kPairStride + row * kRowStride + 32);
        cell[0] = 0.f;
        cell[1] = 0.f;
    }
    for (int i = tid; i < kWaves * 16; i += kThreads) smem[(i >> 4) * kSlabFloats + 4 * kPairStride * 2 + (i & 15)] = 0.f;
    for (int i = tid; i < 400; i += kThreads) smem[kWinOff + i] = tab->window[i];
    for (int i = tid; i < 1024; i += kThreads) smem[kTwOff + i] = tab->twiddle[i];
    {
        const float* __restrict__ afrag_g = reinterpret_cast<const float*>(tab + 1);
        (void)afrag_g;
        int* desc = reinterpret_cast<int*>(smem + kDescOff);
        if (tid < kWaves * kMaxUnitsPerWave)
            reinterpret_cast<int4k*>(desc)[tid] = reinterpret_cast<const int4k*>(&tab->plan.desc[0][0][0])[tid];
        if (tid < kWaves) desc[4 * kWaves * kMaxUnitsPerWave + tid] = tab->plan.wave_nunits[tid];
    }
    __syncthreads();

    const float* __restrict__ wfrag = reinterpret_cast<const float*>(tab + 1);
    const int4k* desc = reinterpret_cast<const int4k*>(smem + kDescOff) + wave * kMaxUnitsPerWave;
    const int nu = min(kUnitsCap, __builtin_amdgcn_readfirstlane(
        reinterpret_cast<const int*>(smem + kDescOff)[4 * kWaves * kMaxUnitsPerWave + wave]));
    const cf* tw_tab = reinterpret_cast<const cf*>(smem + kTwOff);
    const float* win = smem + kWinOff;
    float* my_rows = smem + wave * kSlabFloats;
    cf* slab_w = reinterpret_cast<cf*>(my_rows);

    // stage-2 lane roles: q = pair, k1 = row
    const int q = lane >> 4, k1 = lane & 15;
    const int partner = q * 16 + ((16 - k1) & 15);
    // LDS float offset of this lane's A-operand row (frame ft * 16 + (lane & 15), k = lane >> 4) per frame tile
    auto p_row = [&](int ft) {
        const int fl = ft * 16 + (lane & 15), fw = fl / kFramesPerWave;
        return fw * kSlabFloats + (fl - fw * kFramesPerWave) * kPStride + (lane >> 4);
    };
    const int prow0 = p_row(0), prow1 = p_row(1);

    unsigned iter = 0;
    while (item < n_items) {
    unsigned my_ticket = 0;
    if (tid == 0) my_ticket = atomicAdd(ticket, 1u);               // the workgroup's next item, needed at the end of this one
    const int clip = __builtin_amdgcn_readfirstlane(item / n_runs), run = item - clip * n_runs;
    n_valid = __builtin_amdgcn_readfirstlane(lengths[clip]);
    {
        const int64_t o = offsets[clip];
        x = pcm + (((int64_t)__builtin_amdgcn_readfirstlane((int)(o >> 32)) << 32) |
                   (uint32_t)__builtin_amdgcn_readfirstlane((int)o));
    }
    const int T = kaldi_num_frames(n_valid, max_frames, t_max);
    const int n_tiles = (T + kTileFrames - 1) / kTileFrames;
    const int tile_begin = run * tiles_per_block;
    const int tile_end = min(tile_begin + tiles_per_block, n_tiles);
    if (tile_begin < tile_end) {
    float* __restrict__ out_clip = out + (size_t)clip * t_max * n_mels;
    double* __restrict__ st = stats + (size_t)clip * 2 * n_mels;
    load_pass(tile_begin * kTileFrames, 0, raw0, prv0);
    load_pass(tile_begin * kTileFrames, 1, raw1, prv1);

    // CMVN sums of this wave's units over the item (a wave owns the same (mel tile, frame tile) units in every tile)
    double s1[kUnitsCap], s2[kUnitsCap];
#pragma unroll
    for (int u = 0; u < kUnitsCap; ++u) s1[u] = s2[u] = 0.0;

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int t0 = tile * kTileFrames;
        auto stage1 = [&](int r, const float (&raw)[18], const float (&prv)[18]) {
            const int pair = 2 * r + h;
            // window column and twiddles: all LDS reads are issued before the first slab write (the compiler
            // cannot move an LDS read above an LDS write of the same array)
            float wv[13];
#pragma unroll
            for (int i = 0; i < 13; ++i) wv[i] = win[32 * i + (i == 12 ? (n2 & 15) : n2)];
            cf tw[16];
#pragma unroll
            for (int kk = 1; kk < 16; ++kk) tw[kk] = tw_tab[32 * kk + n2];
            // per-frame mean over n = 0..399 (n = 32 n1 + n2; n1 = 12 only for n2 < 16)
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                sa += raw[i];
                sb += raw[i + 5];
            }
            const bool lowcol = n2 < 16;
            sa += lowcol ? raw[12] : 0.f;
            sb += lowcol ? raw[17] : 0.f;
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                sa += __shfl_xor(sa, off, 64);
                sb += __shfl_xor(sb, off, 64);
            }
            const float mu_a = sa * (1.0f / 400.0f), mu_b = sb * (1.0f / 400.0f);
            cf v[16], y[16];
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                // (x[n] - mu) - 0.97 (x[n-1] - mu); n == 0: 0.03 (x[0] - mu)
                float ea = (raw[i] - 0.97f * prv[i]) - 0.03f * mu_a;
                float eb = (raw[i + 5] - 0.97f * prv[i + 5]) - 0.03f * mu_b;
                if (i == 0 && n2 == 0) {
                    ea = 0.03f * (raw[0] - mu_a);
                    eb = 0.03f * (raw[5] - mu_b);
                }
                if (i == 12 && !lowcol) {
                    ea = 0.f;
                    eb = 0.f;
                }
                v[i] = cf{ea * wv[i], eb * wv[i]};
            }
            v[13] = v[14] = v[15] = cf{0.f, 0.f};
            dft16(v, y);
            cf* slab = slab_w + pair * kPairStride + n2;
            slab[0] = y[0];
#pragma unroll
            for (int kk = 1; kk < 16; ++kk) slab[kk * kRowStride] = cmul(y[kk], tw[kk]);
        };
        stage1(0, raw0, prv0);
        stage1(1, raw1, prv1);
        wave_lds_sync2();

        // ---------------- stage 2: 32-point DFT over n2 ------------------------------------------
        {
            cf v[32], y[32];
            const cf* row = slab_w + q * kPairStride + k1 * kRowStride;
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = row[i];
            wave_lds_sync2();     // this wave's panel rows (written below) alias its slab
            dft32(v, y);          // y[k2] = Z[k1 + 16 k2]
            float* prow = my_rows + 2 * q * kPStride + k1;
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) {
                float qr = __shfl(y[31 - k2].x, partner, 64);
                float qi = __shfl(y[31 - k2].y, partner, 64);
                if (k1 == 0) {
                    qr = (k2 == 0) ? y[0].x : y[32 - k2].x;
                    qi = (k2 == 0) ? y[0].y : y[32 - k2].y;
                }
                const float sr = y[k2].x + qr, dr = y[k2].x - qr;
                const float si = y[k2].y + qi, di = y[k2].y - qi;
                prow[16 * k2] = sr * sr + di * di;
                prow[kPStride + 16 * k2] = si * si + dr * dr;
            }
            if (k1 == 0) {                         // bin 256
                prow[256] = 4.0f * y[16].x * y[16].x;
                prow[kPStride + 256] = 4.0f * y[16].y * y[16].y;
            }
        }
        // next tile's PCM: requested now (few live registers from here on), lands during the mel phase
        if (tile + 1 < tile_end) {
            load_pass(t0 + kTileFrames, 0, raw0, prv0);
            load_pass(t0 + kTileFrames, 1, raw1, prv1);
        }
        __syncthreads();

        // ---------------- banded mel projection (frames x mels) + ln + CMVN statistics -----------
        int4k d = desc[0];                                          // {kstart, nsteps, aoff, mt | ft << 8}
#pragma unroll
        for (int u = 0; u < kUnitsCap; ++u) {
            if (u < nu) {
                const int ks = __builtin_amdgcn_readfirstlane(d.x);
                const int ns = __builtin_amdgcn_readfirstlane(d.y);     // multiple of 4 (tables.cpp)
                const int ao = __builtin_amdgcn_readfirstlane(d.z);
                const int code = __builtin_amdgcn_readfirstlane(d.w);
                d = desc[u + 1 < kMaxUnitsPerWave ? u + 1 : u];
                const int mt = code & 0xff, ft = code >> 8;
                const float* w_ptr = wfrag + ao + lane;
                const float* p_ptr = smem + (ft == 0 ? prow0 : prow1) + ks;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
                float pv[4], wq[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pv[i] = p_ptr[4 * i];
                    wq[i] = w_ptr[i * 64];
                }
                for (int s = 4; s < ns; s += 4) {
                    float pn[4], wn[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pn[i] = p_ptr[4 * (s + i)];
                        wn[i] = w_ptr[(s + i) * 64];
                    }
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i], wq[i], acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i + 1], wq[i + 1], acc2, 0, 0, 0);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        pv[i] = pn[i];
                        wq[i] = wn[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i], wq[i], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i + 1], wq[i + 1], acc2, 0, 0, 0);
                }
                acc += acc2;
                const int mel = mt * 16 + (lane & 15);
                const int frame0 = t0 + ft * 16 + (lane >> 4) * 4;
                const unsigned o0 = (unsigned)(frame0 * n_mels + mel);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (mel < n_mels && frame0 + r < T) {
                        const float lg = __log2f(fmaxf(acc[r], kMelFloor)) * 0.69314718055994531f;
                        out_clip[o0 + (unsigned)(r * n_mels)] = lg;
                        s1[u] += (double)lg;         // f64 from the first add: var = E[x^2] - mean^2 cancels ~3 digits
                        s2[u] += (double)lg * (double)lg;
                    }
                }
            }
        }
        __syncthreads();      // panel reads done before the next tile's slab writes
    }

    // ---------------- CMVN sums of the run: lanes l, l+16, l+32, l+48 share a mel bin ---------------
#pragma unroll
    for (int u = 0; u < kUnitsCap; ++u) {
        if (u < nu) {
            double a = s1[u], b2 = s2[u];
            a += __shfl_xor(a, 16, 64);
            b2 += __shfl_xor(b2, 16, 64);
            a += __shfl_xor(a, 32, 64);
            b2 += __shfl_xor(b2, 32, 64);
            const int code = reinterpret_cast<const int*>(desc)[4 * u + 3];
            const int mel = (code & 0xff) * 16 + (lane & 15);
            if (lane < 16 && mel < n_mels) {
                atomicAdd(&st[mel], a);
                atomicAdd(&st[n_mels + mel], b2);
            }
        }
    }
    }   // item has tiles
    // hand-over of the next item through one of two alternating LDS slots (a slot is rewritten only after the
    // barrier of the following item, which every thread reaches after this read)
    if (tid == 0) s_next[iter & 1u] = my_ticket + gridDim.x;
    __syncthreads();
    item = (int)min(s_next[iter & 1u], (unsigned)n_items);
    ++iter;
    }   // items
}

// out[b][t][m] <- (x - mean_m) / std_m for t < T_b, 0 for padded rows; masks <- 0; attention mask.
__global__ __launch_bounds__(256) void kaldi_cmvn_kernel(
    float* __restrict__ out, const int32_t* __restrict__ lengths, const double* __restrict__ stats,
    int n_mels, int t_max, int max_frames, int flags, void* __restrict__ attn, int attn_bytes,
    const int32_t* __restrict__ mask_tbl, int n_masks) {
    const int clip = blockIdx.y;
    const int T = kaldi_num_frames(lengths[clip], max_frames, t_max);
    float* __restrict__ o = out + (size_t)clip * t_max * n_mels;
    const double* st = stats + (size_t)clip * 2 * n_mels;
    int m_axis[8], m_s[8], m_e[8];
    const int nm = mask_tbl ? (n_masks > 8 ? 8 : n_masks) : 0;
    for (int i = 0; i < nm; ++i) {
        const int32_t* m = mask_tbl + ((size_t)clip * n_masks + i) * 3;
        m_axis[i] = m[0];
        m_s[i] = m[1];
        m_e[i] = m[2];
    }
    const int total = t_max * n_mels;
    const int stride = gridDim.x * blockDim.x;
    // per-bin mean and population std of the clip, once per workgroup (float64 sums -> float32, as NumPy's
    // float32 mean / std of HF:...speech_to_text.py:150-155 up to rounding)
    __shared__ float s_mean[128], s_std[128];
    if ((int)threadIdx.x < n_mels) {
        const double inv_n = T > 0 ? 1.0 / (double)T : 0.0;
        const double mean = st[threadIdx.x] * inv_n;
        double var = st[n_mels + threadIdx.x] * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_mean[threadIdx.x] = (float)mean;
        s_std[threadIdx.x] = (float)sqrt(var);
    }
    __syncthreads();
    const bool no_cmvn = flags & STTFE_KALDI_NO_CMVN;
    auto norm = [&](float x, int t, int mel) {
        float y = no_cmvn ? x : (x - s_mean[mel]) / s_std[mel];
        for (int i = 0; i < nm; ++i) {
            const int idx = (m_axis[i] == STTFE_AXIS_TIME) ? t : mel;
            if (idx >= m_s[i] && idx < m_e[i]) y = 0.0f;
        }
        return y;
    };
    if ((n_mels & 3) == 0 && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
        // four bins of one row per thread and trip (a row is a whole number of 16-byte groups); padded rows are
        // written without being read
        const int groups = total >> 2, row_groups = n_mels >> 2;
        for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
            const int t = g / row_groups, mel = (g - t * row_groups) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < T) {
                v = reinterpret_cast<const float4*>(o)[g];
                v.x = norm(v.x, t, mel);
                v.y = norm(v.y, t, mel + 1);
                v.z = norm(v.z, t, mel + 2);
                v.w = norm(v.w, t, mel + 3);
            }
            reinterpret_cast<float4*>(o)[g] = v;
        }
    } else {
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
            const int t = e / n_mels, mel = e - t * n_mels;
            o[e] = t < T ? norm(o[e], t, mel) : 0.0f;
        }
    }
    if (attn) {
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < t_max; t += stride) {
            if (attn_bytes == 8) reinterpret_cast<int64_t*>(attn)[(size_t)clip * t_max + t] = t < T ? 1 : 0;
            else reinterpret_cast<int32_t*>(attn)[(size_t)clip * t_max + t] = t < T ? 1 : 0;
        }
    }
}

}  // namespace sttfe

using namespace sttfe;

extern "C" int sttfe_kaldi_workspace_bytes(int batch, int n_mels, size_t* bytes) {
    if (batch < 0 || n_mels < 1 || n_mels > 128 || !bytes) return STTFE_E_INVALID_ARG;
    // per-clip per-bin float64 sums, then the work-item ticket counter
    *bytes = ((size_t)batch * 2 * n_mels * sizeof(double) + 255) / 256 * 256 + 256;
    return STTFE_OK;
}

extern "C" int sttfe_kaldi_fbank_cmvn(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                      int batch, int n_mels, int t_max, int max_frames, int flags,
                                      const void* dev_tables, float* out, void* attn_mask, int attn_mask_bytes,
                                      const int32_t* mask_tbl, int n_masks,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    if (batch < 0 || n_mels < 1 || n_mels > 128 || t_max < 0 || n_masks < 0 || n_masks > 8) return STTFE_E_INVALID_ARG;
    if (attn_mask && attn_mask_bytes != 4 && attn_mask_bytes != 8) return STTFE_E_INVALID_ARG;
    if (batch == 0 || t_max == 0) return STTFE_OK;
    if (!pcm || !offsets || !lengths || !dev_tables || !out || !workspace) return STTFE_E_INVALID_ARG;
    if (batch > 65535) return STTFE_E_UNSUPPORTED;
    size_t need = 0;
    sttfe_kaldi_workspace_bytes(batch, n_mels, &need);
    if (workspace_bytes < need) return STTFE_E_WORKSPACE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    double* stats = reinterpret_cast<double*>(workspace);
    (void)hipGetLastError();
    if (hip_failed(hipMemsetAsync(stats, 0, need, stream), "hipMemsetAsync(cmvn stats)")) return STTFE_E_LAUNCH;
    const auto* tab = reinterpret_cast<const KaldiTablesHeader*>(dev_tables);
    const int n_tiles = (t_max + kaldi::kTileFrames - 1) / kaldi::kTileFrames;
    static const int resident = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
            cus = 256;
        return 2 * cus;
    }();
    // persistent workgroups, two per CU; short runs keep ragged batches balanced: about eight items per workgroup,
    // taking half of batch x n_tiles as the batch's tiles (the lengths are on the device), within [2, 8] -- 2 tiles
    // (64 frames) for config 2's 256 clips (1: 0.123, 2: 0.111, 3: 0.119 ms), 3 at 1024 clips, 8 at 4096
    // (1.505 -> 1.408 ms).
    static const int tpb_env = [] {
        const char* e = getenv("STTFE_KALDI_TILES_PER_ITEM");      // tuning knob (0 / unset = automatic)
        const int v = e ? atoi(e) : 0;
        return v < 1 ? 0 : v;
    }();
    int tiles_per_block = tpb_env;
    if (tiles_per_block == 0) {
        const long est = (long)batch * n_tiles / 2 / (8L * resident);
        tiles_per_block = est < 2 ? 2 : (est > 8 ? 8 : (int)est);
    }
    const int n_runs = (n_tiles + tiles_per_block - 1) / tiles_per_block;
    if ((long)n_runs * batch > 0x3fffffffL) return STTFE_E_UNSUPPORTED;
    const int n_items = n_runs * batch;
    const int grid = n_items < resident ? n_items : resident;
    unsigned* ticket = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) + need - 256);
    {
        ScopedTiming t(kTimeKaldiFbank, stream);
        hipLaunchKernelGGL(kaldi_fbank_kernel, dim3(grid), dim3(kaldi::kThreads), 0, stream, pcm, offsets,
                           lengths, n_mels, t_max, max_frames, tab, out, stats, tiles_per_block, n_runs, n_items, ticket);
    }
    {
        ScopedTiming t(kTimeKaldiCmvn, stream);
        int cx = (t_max * n_mels + 255) / 256;
        if (cx > 64) cx = 64;
        hipLaunchKernelGGL(kaldi_cmvn_kernel, dim3(cx, batch), dim3(256), 0, stream, out, lengths, stats, n_mels,
                           t_max, max_frames, flags, attn_mask, attn_mask_bytes, mask_tbl, n_masks);
    }
    if (hip_failed(hipGetLastError(), "kaldi kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}
