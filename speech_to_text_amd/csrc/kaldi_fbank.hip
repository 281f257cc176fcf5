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
// STTFE_KALDI_F64 (the default of HipSpeech2TextFeatureExtractor): the same result computed the way HF's NumPy path
//   computes it -- float64 framing / window / transform, complex64 rounding, float64 mel + log (kaldi_fbank64_kernel),
//   and NumPy-order float32 CMVN statistics (kaldi_cmvn_seq_kernel, or kaldi_stats_seq_kernel + kaldi_cmvn_kernel for
//   shapes the fused kernel does not take); bit-identical to the HF goldens, see the comments at those kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/sttfe.h"
#include <type_traits>

#include "fft_core.h"
#include "tables.h"
#include "timing.h"

namespace sttfe {

int current_device_cus();      // whisper_logmel.hip

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
constexpr int kLdsFloats = (kDescOff + kWaves * kMaxUnitsPerWave * 4 + kWaves + 3) & ~3;    // 74 KB: two workgroups per CU
constexpr int kUnitsCap = 4;               // units one wave owns: ceil(8 mel tiles x 2 frame tiles / 4 waves) at most
constexpr float kMelFloor = 1.1920928955078125e-07f;
}  // namespace kaldi

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int int4k __attribute__((ext_vector_type(4)));
typedef double double2k __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// ---- work items of the two persistent Kaldi kernels ----
// Dense items are (clip, run of tiles_per_block 32-frame tiles) for the runs the LONGEST clip reaches; on a ragged
// batch (config 2: 2-10 s clips) 40 % of them are empty, and an empty item still costs a ticket round trip and a
// barrier.  Batches of up to kKaldiPlanClips clips are compacted instead: every workgroup builds the exclusive prefix of
// the clips' own run counts in LDS and finds an item's clip with two 64-way ballot steps (as the Whisper kernel does).
constexpr int kKaldiPlanClips = 1024;
struct KaldiItems {
    const int* plan;        // LDS: plan[c] = items before clip c, plan[batch] = all items (compacted mode)
    int n_items, n_runs, batch;
    bool planned;
};
// Called by all 256 threads of the workgroup (contains barriers).  `plan` has kKaldiPlanClips + 1 ints, `scratch` 4.
__device__ __forceinline__ KaldiItems kaldi_build_items(int* plan, int* scratch, const int32_t* __restrict__ lengths,
                                                       int batch, int t_max, int max_frames, int tiles_per_block,
                                                       int n_runs_dense, int n_items_dense, int tile_frames = kaldi::kTileFrames) {
    KaldiItems it{plan, n_items_dense, n_runs_dense, batch, batch <= kKaldiPlanClips};
    if (!it.planned) return it;
    constexpr int kPer = kKaldiPlanClips / kaldi::kThreads;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (batch + kaldi::kThreads - 1) / kaldi::kThreads;
    int lens[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) lens[k] = lengths[min(tid * per + k, batch - 1)];      // all requested at once
    int before[kPer], sum = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        before[k] = sum;
        if (k < per && tid * per + k < batch) {
            const int T = kaldi_num_frames(lens[k], max_frames, t_max);
            const int tiles = (T + tile_frames - 1) / tile_frames;
            sum += (tiles + tiles_per_block - 1) / tiles_per_block;                        // 0 for a clip without a frame
        }
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int w = 0; w < wave; ++w) base += scratch[w];
#pragma unroll
    for (int k = 0; k < kPer; ++k)
        if (k < per && tid * per + k < batch) plan[tid * per + k] = base + before[k];
    if (tid == kaldi::kThreads - 1) plan[batch] = base + sum;
    __syncthreads();
    it.n_items = plan[batch];
    return it;
}
// item -> (clip, run); all 64 lanes of the wave take part
__device__ __forceinline__ void kaldi_decode_item(const KaldiItems& it, int item, int& clip, int& run) {
    if (it.planned) {
        // largest c with plan[c] <= item (clips without a frame have no items and lose the tie to their successor)
        const int lane = threadIdx.x & 63;
        const int stride = (it.batch + 63) >> 6;
        const int c1 = lane * stride;
        const bool le1 = c1 < it.batch && it.plan[min(c1, it.batch)] <= item;
        const int block = __popcll(__ballot(le1)) - 1;
        const int c2 = block * stride + lane;
        const bool le2 = lane < stride && c2 < it.batch && it.plan[min(c2, it.batch)] <= item;
        const int c = block * stride + __popcll(__ballot(le2)) - 1;
        clip = __builtin_amdgcn_readfirstlane(c);
        run = __builtin_amdgcn_readfirstlane(item - it.plan[c]);
    } else {
        clip = __builtin_amdgcn_readfirstlane(item / it.n_runs);
        run = __builtin_amdgcn_readfirstlane(item - clip * it.n_runs);
    }
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
    int tiles_per_block, int n_runs, int n_items, unsigned* __restrict__ ticket, int batch) {
    using namespace kaldi;
    __shared__ __attribute__((aligned(16))) float smem[kLdsFloats];
    __shared__ unsigned s_next[2];
    __shared__ int s_plan[kKaldiPlanClips + 1], s_scan[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tab->magic != kKaldiMagic || tab->n_mels != n_mels) return;      // the CMVN pass poisons the output
    const KaldiItems items = kaldi_build_items(s_plan, s_scan, lengths, batch, t_max, max_frames, tiles_per_block, n_runs, n_items);
    n_items = items.n_items;
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

    // The first two items of a workgroup are static (blockIdx.x, gridDim.x + blockIdx.x), tickets start at the third;
    // the ticket, clip, length and offset of the NEXT item are requested at the top of the current one.
    unsigned iter = 0, my_ticket = 0;
    int next_item = min((int)(blockIdx.x + gridDim.x), n_items);
    int clip_cur, run_cur, clip_nx = 0, run_nx = 0;
    kaldi_decode_item(items, item, clip_cur, run_cur);
    if (next_item < n_items) kaldi_decode_item(items, next_item, clip_nx, run_nx);
    int len_cur = lengths[clip_cur], len_nx = 0;
    int64_t off_cur = offsets[clip_cur], off_nx = 0;
    while (item < n_items) {
    if (tid == 0 && next_item < n_items) my_ticket = atomicInc(ticket, 0xffffffffu);   // -> the item after the next
    if (next_item < n_items) {
        len_nx = lengths[clip_nx];
        off_nx = offsets[clip_nx];
    }
    const int clip = clip_cur, run = run_cur;
    n_valid = __builtin_amdgcn_readfirstlane(len_cur);
    x = pcm + (((int64_t)__builtin_amdgcn_readfirstlane((int)(off_cur >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)off_cur));
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
    // (float64 mode: kaldi_stats_seq_kernel reproduces NumPy's sequential float32 reductions instead)
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
    if (tid == 0) s_next[iter & 1u] = next_item < n_items ? my_ticket + 2 * gridDim.x : (unsigned)n_items;
    __syncthreads();
    const int after_next = (int)min(s_next[iter & 1u], (unsigned)n_items);
    ++iter;
    item = next_item;
    clip_cur = clip_nx;
    run_cur = run_nx;
    len_cur = len_nx;
    off_cur = off_nx;
    next_item = after_next;
    if (next_item < n_items) kaldi_decode_item(items, next_item, clip_nx, run_nx);
    }   // items
}

// =====================================================================================================================
// STTFE_KALDI_F64: HF's NumPy path frames, windows and transforms in float64 (HF:audio_utils.py:957-988) and only then
// rounds each spectrum to complex64; the mel projection and the logarithm are float64 again and the result is rounded
// to float32 once (:993-1015).  This kernel does the same arithmetic, a wave at a time, without matrix cores.
//
// A wave owns 8 frames of a 32-frame tile and handles them as 4 frame pairs: the two real frames of a pair are the
// real and imaginary part of ONE complex 512-point DFT, done as 8 x 8 x 8: lane l holds z[l + 64 m]; three
// register-resident 8-point DFTs with two transposes through an 8 KB per-wave exchange area (16-byte complex cells,
// XOR-swizzled columns so that both the row-wise writes and the column-wise reads are bank-conflict free).  The pair's
// two float64 power rows then alias the exchange area, and the band-sparse mel sums + logarithm read them from there.
// Everything between PCM and log-mel is wave-local: no workgroup barrier inside an item.
//
// 40 KB of LDS and <= 128 VGPRs: four workgroups (16 waves) per CU, which is what hides the LDS round trips and the
// PCM latency (at two workgroups per CU the same arithmetic took 0.235 ms instead of 0.1x ms).
// =====================================================================================================================
namespace kaldi64 {
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kHop = 160;
constexpr int kFrame = 400;
constexpr int kPowStride = 258;            // doubles per power row (257 bins)
constexpr int kMaxBandWeights = 520;       // a triangular bank over 257 bins has at most 2 non-zero weights per bin
constexpr int kLogEntries = 64;
constexpr int kWgPerCu = 3;
constexpr int kExchBytes = 8192;           // per wave: 512 complex doubles
constexpr int kTw2Off = kWaves * kExchBytes;            // 64 x complex double: W64^(j q) at [q][j]
constexpr int kBandOff = kTw2Off + 64 * 16;             // 128 x {first bin | bins << 16, weight offset}
constexpr int kWtsOff = kBandOff + 128 * 8;             // float64 band weights
constexpr int kLogOff = kWtsOff + kMaxBandWeights * 8;  // 64 x {1 / c, ln c}
constexpr int kWinOff = kLogOff + kLogEntries * 16;     // float64 window, zero-padded to 448 entries
constexpr int kLdsBytes = kWinOff + 448 * 8;
static_assert(3 * (kLdsBytes + 16) <= 160 * 1024, "three workgroups per CU");
static_assert(2 * kPowStride * 8 <= kExchBytes, "a pair's two power rows alias the wave's exchange area");
}  // namespace kaldi64

struct cd {
    double x, y;
};
typedef int int2k __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cd operator+(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmuld(cd a, cd b) { return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ void dft4d(cd& a0, cd& a1, cd& a2, cd& a3) {
    const cd s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = cd{d02.x + d13.y, d02.y - d13.x};      // d02 - i d13
    a3 = cd{d02.x - d13.y, d02.y + d13.x};      // d02 + i d13
}
// 8-point forward DFT in place, natural order in / out (2 x 4 Cooley-Tukey, as dft8 of fft_core.h)
__device__ __forceinline__ void dft8d(cd (&v)[8]) {
    constexpr double kH = 0.70710678118654752440;
    cd e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cd o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4d(e0, e1, e2, e3);
    dft4d(o0, o1, o2, o3);
    const cd w1{(o1.x + o1.y) * kH, (o1.y - o1.x) * kH};       // * (1 - i) / sqrt 2
    const cd w2{o2.y, -o2.x};                                  // * -i
    const cd w3{(o3.y - o3.x) * kH, -(o3.x + o3.y) * kH};      // * (-1 - i) / sqrt 2
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    v[1] = e1 + w1;
    v[5] = e1 - w1;
    v[2] = e2 + w2;
    v[6] = e2 - w2;
    v[3] = e3 + w3;
    v[7] = e3 - w3;
}
// ln x for a normal x > 0 with an absolute error of a few 1e-16 (the caller rounds the result to float32, whose ulp at
// the |ln| <= 30 of mel energies is >= 1e-10 apart from results next to 0): x = 2^e m, m in [1, 2), c = the centre of
// m's 1/64-wide interval, r = m / c - 1 (|r| <= 2^-7), ln x = e ln 2 + ln c + log1p(r) with log1p to r^7
// (the next term is < 2e-18).  Fifteen float64 instructions instead of the library routine's ~100.
__device__ __forceinline__ double log_tab64(double x, const double2k* __restrict__ ltab) {
    const long long bits = __double_as_longlong(x);
    const int hi = (int)(bits >> 32);
    const int e = (hi >> 20) - 1023;
    const double m = __longlong_as_double((bits & 0x000FFFFFFFFFFFFFll) | 0x3FF0000000000000ll);
    const double2k t = ltab[(hi >> 14) & (kaldi64::kLogEntries - 1)];
    const double r = fma(m, t.x, -1.0);
    double p = fma(r, 1.0 / 7.0, -1.0 / 6.0);
    p = fma(r, p, 0.2);
    p = fma(r, p, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    p = fma(r, p, 1.0);
    return fma(r, p, fma((double)e, 0.693147180559945309417, t.y));
}
// sum of a float64 over the wave, the same value in every lane: rotations inside the rows of 16 lanes (DPP), then the
// four row sums through scalar registers -- no LDS traffic, unlike the shuffle form
__device__ __forceinline__ double wave_sum_f64(double v) {
    auto ror = [](double a, auto ctrl) {
        const long long b = __double_as_longlong(a);
        const int lo = __builtin_amdgcn_mov_dpp((int)b, decltype(ctrl)::value, 0xf, 0xf, false);      // every lane has a source:
        const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), decltype(ctrl)::value, 0xf, 0xf, false);   // no "old" value to set up
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    v += ror(v, std::integral_constant<int, 0x128>{});      // row_ror:8
    v += ror(v, std::integral_constant<int, 0x124>{});      // row_ror:4
    v += ror(v, std::integral_constant<int, 0x122>{});      // row_ror:2
    v += ror(v, std::integral_constant<int, 0x121>{});      // row_ror:1
    auto row = [&](int l) {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_readlane((int)b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    return (row(0) + row(16)) + (row(32) + row(48));
}

__global__ __launch_bounds__(kaldi64::kThreads, kaldi64::kWgPerCu) void kaldi_fbank64_kernel(
    const float* __restrict__ pcm, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ lengths, int n_mels, int t_max, int max_frames,
    const KaldiTablesHeader* __restrict__ tab, float* __restrict__ out,
    int n_runs, int n_items, int batch, int item_frames) {
    using namespace kaldi64;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kLdsBytes];
    __shared__ int s_plan[kKaldiPlanClips + 1], s_scan[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tab->magic != kKaldiMagic || tab->n_mels != n_mels) return;      // the CMVN pass poisons the output
    // Start of a launch: the float64 section of the tables blob (window[400], W512 (cos, -sin)[512], then the mel bands
    // and their weights) is REQUESTED here, before the item plan is built -- the plan's own loads (the clip lengths) and
    // barriers then run under that round trip instead of after it (copy loops behind the plan were three more dependent
    // round trips at the top of every launch; a 16-clip call 0.0176 -> 0.0159 ms, nothing measurable at 256 clips).
    const double* __restrict__ t64 = reinterpret_cast<const double*>(
        reinterpret_cast<const char*>(tab + 1) + (((size_t)tab->afrag_floats * 4 + 7) & ~(size_t)7));
    const double* __restrict__ w512 = t64 + kFrame;
    const int nm = min(n_mels, 128);
    const int nw = min(tab->reserved, kMaxBandWeights);
    const int4k* __restrict__ bg = reinterpret_cast<const int4k*>(t64 + kFrame + 1024);
    const double* __restrict__ wg = t64 + kFrame + 1024 + 2 * nm;
    static_assert(448 <= 2 * kThreads && kMaxBandWeights <= 3 * kThreads && 128 <= kThreads, "one pass of registers");
    double win_r[2], wt_r[3];
#pragma unroll
    for (int k = 0; k < 2; ++k) win_r[k] = t64[min(tid + k * kThreads, kFrame - 1)];
#pragma unroll
    for (int k = 0; k < 3; ++k) wt_r[k] = wg[min(tid + k * kThreads, max(nw - 1, 0))];
    const int4k band_r = bg[min(tid, nm - 1)];
    const int tw_i = (8 * (tid & 7) * ((tid >> 3) & 7)) & 511;          // [q][j]: W512^(8 j q)
    const double tw_re = w512[2 * tw_i], tw_im = w512[2 * tw_i + 1];
    // Items belong to WAVES, not workgroups: an item is item_frames consecutive frames of one clip (1, 2, 4 ... frame
    // pairs; the host picks few pairs for small batches) and wave w of the launch takes items w, w + waves, w + 2 waves,
    // ... of the compacted item space -- nothing in the item loop is collective (no barrier, no LDS hand-over) and there
    // is no ticket: the items cost the same, so striding balances them to within one item, and one counter serves only
    // ~88 tickets per microsecond (3072 waves asking for one-pair items: 0.85 ms for config 2 instead of 0.17).  With
    // workgroup-wide items of 32 frames config 2 gave a workgroup 6.25 items: the kernel ended when those with seven
    // did, and smaller workgroup-wide items paid a barrier per frame pair (256 clips: 0.173 ms whatever the size).
    // (the host launches this kernel on at most kKaldiPlanClips clips at a time: striding needs the compacted item space --
    // over dense items a stride that is a multiple of n_runs would pin a wave to one run index, mostly past the clips' ends)
    const KaldiItems items = kaldi_build_items(s_plan, s_scan, lengths, batch, t_max, max_frames, 1, n_runs, n_items, item_frames);
    n_items = items.n_items;
    const int n_workers = (int)gridDim.x * kWaves;
    int item = (int)blockIdx.x * kWaves + wave;
    if ((int)blockIdx.x * kWaves >= n_items) return;                   // (the whole workgroup: there is a barrier below)
    {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kThreads;
            if (i < 448) reinterpret_cast<double*>(smem + kWinOff)[i] = i < kFrame ? win_r[k] : 0.0;
        }
        if (tid < nm) reinterpret_cast<int2k*>(smem + kBandOff)[tid] = int2k{band_r.x | (band_r.y << 16), band_r.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = tid + k * kThreads;
            if (i < nw) reinterpret_cast<double*>(smem + kWtsOff)[i] = wt_r[k];
        }
        if (tid < 64) {
            reinterpret_cast<cd*>(smem + kTw2Off)[tid] = cd{tw_re, tw_im};
        } else if (tid < 64 + kLogEntries) {
            const double c = 1.0 + ((double)(tid - 64) + 0.5) * (1.0 / kLogEntries);
            reinterpret_cast<double2k*>(smem + kLogOff)[tid - 64] = double2k{1.0 / c, log(c)};
        }
        __syncthreads();
    }
    cd* exch = reinterpret_cast<cd*>(smem + wave * kExchBytes);
    double* pp = reinterpret_cast<double*>(exch);
    const cd* tw2 = reinterpret_cast<const cd*>(smem + kTw2Off) + (lane & 7);
    const int2k* band64 = reinterpret_cast<const int2k*>(smem + kBandOff);
    const double* wts64 = reinterpret_cast<const double*>(smem + kWtsOff);
    const double2k* ltab = reinterpret_cast<const double2k*>(smem + kLogOff);
    const int mel_rounds = (2 * n_mels + 63) >> 6;

    // the next item's clip, length and offset are requested at the top of the current item
    if (item >= n_items) return;                                      // (per wave: nothing below is collective)
    int next_item = min(item + n_workers, n_items);
    int clip_cur, run_cur, clip_nx = 0, run_nx = 0;
    kaldi_decode_item(items, item, clip_cur, run_cur);
    if (next_item < n_items) kaldi_decode_item(items, next_item, clip_nx, run_nx);
    int len_cur = lengths[clip_cur], len_nx = 0;
    int64_t off_cur = offsets[clip_cur], off_nx = 0;
    // the samples of the pair a wave works on next live across items: the last pair of an item requests the first pair
    // of the NEXT item (its clip, length and offset were requested at the top of this one), so an item does not begin
    // with an exposed PCM round trip -- which is what lets small batches use small items
    float xa0[7], xp0[7], xb0[7], xq0[7];
    bool prefetched = false;
    while (item < n_items) {
        if (next_item < n_items) {
            len_nx = lengths[clip_nx];
            off_nx = offsets[clip_nx];
        }
        const int clip = clip_cur, run = run_cur;
        const int n_valid = __builtin_amdgcn_readfirstlane(len_cur);
        const float* __restrict__ x = pcm + (((int64_t)__builtin_amdgcn_readfirstlane((int)(off_cur >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)off_cur));
        const int T = kaldi_num_frames(n_valid, max_frames, t_max);
        // the item's frames
        const int f_first = run * item_frames;
        const int f_run_end = min(f_first + item_frames, T);
        float* __restrict__ out_clip = out + (size_t)clip * t_max * n_mels;

        // samples x[n], x[n - 1] of the pair's two frames for n = lane + 64 m (frame b starts one hop later).  Frame a is
        // a valid frame (fa < T), so all of its samples are inside the clip; frame b of an odd last pair is not: frame a
        // is loaded again in its place (its results are not stored).  No per-sample bounds anywhere.
        auto load_pair = [&](const float* __restrict__ x, int T, int n_valid, int fa, float (&xa)[7], float (&xp)[7],
                             float (&xb)[7], float (&xq)[7]) {
            const int hop_b = (fa + 1 < T) ? kHop : 0;                // wave-uniform
            const int i0 = kHop * fa + lane;
            const float* __restrict__ pa = x + i0;
            const float* __restrict__ pb = pa + hop_b;
            xa[0] = pa[0];
            xb[0] = pb[0];
            xp[0] = x[max(i0 - 1, 0)];                                // (lane 0, m 0) is never used: buffer[0] *= 0.03
            xq[0] = x[max(i0 + hop_b - 1, 0)];
#pragma unroll
            for (int m = 1; m < 6; ++m) {
                xa[m] = pa[64 * m];
                xp[m] = pa[64 * m - 1];
                xb[m] = pb[64 * m];
                xq[m] = pb[64 * m - 1];
            }
            // m = 6: only lanes 0..15 are inside the frame; the others read a clamped (valid) address and drop the value
            // -- a divergent block here makes the compiler wait for ALL of the loads at its end
            const bool in6 = lane < kFrame - 6 * 64;
            const int last = n_valid - 1;
            const float a6 = x[min(i0 + 64 * 6, last)], p6 = x[min(i0 + 64 * 6 - 1, last)];
            const float b6 = x[min(i0 + hop_b + 64 * 6, last)], q6 = x[min(i0 + hop_b + 64 * 6 - 1, last)];
            xa[6] = in6 ? a6 : 0.0f;
            xp[6] = in6 ? p6 : 0.0f;
            xb[6] = in6 ? b6 : 0.0f;
            xq[6] = in6 ? q6 : 0.0f;
        };
        // frame index of the wave's j-th pair of the run
        auto pair_frame = [&](int j) { return f_first + 2 * j; };
        // one frame pair: window stage on (xa, xp, xb, xq), the NEXT pair's samples requested into (na, np, nb, nq)
        auto do_pair = [&](int fa, int fa_next, const float (&xa)[7], const float (&xp)[7], const float (&xb)[7],
                           const float (&xq)[7], float (&na)[7], float (&np)[7], float (&nb)[7], float (&nq)[7]) {
            // per-frame mean over the 400 samples (HF:audio_utils.py:979-980), float64
            double sa = 0.0, sb = 0.0;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                sa += (double)xa[m];
                sb += (double)xb[m];
            }
            const double mu_a = wave_sum_f64(sa) * (1.0 / 400.0), mu_b = wave_sum_f64(sb) * (1.0 / 400.0);   // (a multiply: 1e-16 from np.mean's division)
            const double ca = (1.0 - 0.97) * mu_a, cb = (1.0 - 0.97) * mu_b;
            cd v[8];
            // the lane's window column comes from LDS per pair (opaque offset: the compiler would otherwise keep the
            // loop-invariant column in 14 registers, which this kernel does not have)
            int lane_w = lane;
            asm volatile("" : "+v"(lane_w));
            const double* win64 = reinterpret_cast<const double*>(smem + kWinOff) + lane_w;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                // buffer - mean; buffer[1:] -= 0.97 buffer[:-1]; buffer[0] *= 1 - 0.97; * window (:979-986);
                // (x - mu) - 0.97 (x' - mu) is evaluated as (x - 0.97 x') - 0.03 mu: one operation less per sample, and
                // 1e-16 relative away from HF's own float64 rounding, far below the complex64 rounding that follows
                double ea = fma(-0.97, (double)xp[m], (double)xa[m]) - ca;
                double eb = fma(-0.97, (double)xq[m], (double)xb[m]) - cb;
                if (m == 0 && lane == 0) {
                    ea = ((double)xa[m] - mu_a) * (1.0 - 0.97);
                    eb = ((double)xb[m] - mu_b) * (1.0 - 0.97);
                }
                const double wm = win64[64 * m];                      // 0 past the frame
                v[m] = cd{ea * wm, eb * wm};
            }
            v[7] = cd{0.0, 0.0};
            // first-pass twiddles W512^(l q) from the table in global memory (8 KB, L1 resident), requested BEFORE
            // the next pair's PCM so that waiting for them does not wait for HBM
            cd tw1[8];
            int lane_o = lane;                                        // opaque per pair: keeps the (loop-invariant) twiddle
            asm volatile("" : "+v"(lane_o));                          // loads inside the loop instead of 60 registers
#pragma unroll
            for (int q = 1; q < 8; ++q) {
                const int i1 = (int)__umul24((unsigned)lane_o, (unsigned)q) & 511;      // (24-bit multiply: full rate)
                const double2k w = *reinterpret_cast<const double2k*>(w512 + 2 * i1);
                tw1[q] = cd{w.x, w.y};
            }
            // unconditional (past the run's end the current, valid, pair is requested again and never used): a
            // conditional block makes the compiler wait for the loads at its end to merge them into the loop registers
            // (the last pair of the item requests the first pair of the next item; without one, itself again)
            const float* __restrict__ lx = x;
            int lT = T, lnv = n_valid, lfa = fa_next < f_run_end ? fa_next : fa;
            if (fa_next >= f_run_end && next_item < n_items) {        // wave-uniform; scalar work only, no loads inside
                const int nv = __builtin_amdgcn_readfirstlane(len_nx);
                const int Tn = kaldi_num_frames(nv, max_frames, t_max);
                const int fn = run_nx * item_frames;
                if (fn < Tn) {
                    lx = pcm + (((int64_t)__builtin_amdgcn_readfirstlane((int)(off_nx >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)off_nx));
                    lT = Tn;
                    lnv = nv;
                    lfa = fn;
                    prefetched = true;
                }
            }
            load_pair(lx, lT, lnv, lfa, na, np, nb, nq);
            dft8d(v);                                                 // over m: Y_l[q]
            wave_lds_sync2();                                         // after the previous pair's mel reads
            {
                cd* row = exch + lane * 8;
                const int sw = lane & 7;
                row[sw] = v[0];
#pragma unroll
                for (int q = 1; q < 8; ++q) row[q ^ sw] = cmuld(v[q], tw1[q]);
            }
            wave_lds_sync2();
            {
                const int l0 = lane & 7, q2 = lane >> 3;              // lane = l0 + 8 q: rows l0 + 8 l1, column q
                const cd* col = exch + l0 * 8 + (q2 ^ l0);
                cd t2[8];
#pragma unroll
                for (int s2 = 1; s2 < 8; ++s2) t2[s2] = tw2[8 * s2 + (lane_o - lane)];
#pragma unroll
                for (int l1 = 0; l1 < 8; ++l1) v[l1] = col[64 * l1];
                wave_lds_sync2();
                dft8d(v);                                             // over l1: U[l0][s][q]
                cd* dst = exch + q2 * 64;
                const int sw = l0 ^ q2;
                dst[sw] = v[0];
#pragma unroll
                for (int s2 = 1; s2 < 8; ++s2) dst[s2 * 8 + (sw ^ s2)] = cmuld(v[s2], t2[s2]);
            }
            wave_lds_sync2();
            {
                const int q3 = lane & 7, s3 = lane >> 3;              // lane = q + 8 s = bin index mod 64
                const cd* src = exch + q3 * 64 + s3 * 8;
                const int sw = s3 ^ q3;
#pragma unroll
                for (int l0 = 0; l0 < 8; ++l0) v[l0] = src[l0 ^ sw];
                dft8d(v);                                             // over l0: v[t] = Z[lane + 64 t]
            }
            wave_lds_sync2();                                         // the power rows overwrite the exchange area
            // spectra of the two real frames: A[k] = (Z[k] + conj Z[512 - k]) / 2, B[k] = (Z[k] - conj Z[512 - k]) / 2i,
            // each rounded to complex64 as HF stores them (:966, :988), then |.|^2 (the 1/4 is in the mel bank)
            {
                const int partner = (64 - lane) & 63;
                double* prow = pp + lane;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    double qr = __shfl(v[7 - t].x, partner, 64);
                    double qi = __shfl(v[7 - t].y, partner, 64);
                    if (lane == 0) {
                        qr = v[(8 - t) & 7].x;
                        qi = v[(8 - t) & 7].y;
                    }
                    const double sr = (double)(float)(v[t].x + qr), dr = (double)(float)(v[t].x - qr);
                    const double si = (double)(float)(v[t].y + qi), di = (double)(float)(v[t].y - qi);
                    prow[64 * t] = sr * sr + di * di;                 // np.abs(complex64, dtype=float64) ** 2 (:993)
                    prow[kPowStride + 64 * t] = si * si + dr * dr;
                }
                if (lane == 0) {                                      // bin 256
                    const double ar = (double)(float)(2.0 * v[4].x), br = (double)(float)(2.0 * v[4].y);
                    prow[256] = ar * ar;
                    prow[kPowStride + 256] = br * br;
                }
            }
            wave_lds_sync2();
            // mel projection and log in float64, rounded to float32 once (:998-1015): lane pairs = (mel bin, frame a | b),
            // widest bands first so that the lanes of a round have bands of similar length
            for (int r = 0; r < mel_rounds; ++r) {
                const int o = lane + 64 * r;
                const int mel = n_mels - 1 - (o >> 1), fr = o & 1;
                if (mel >= 0) {
                    const int2k b = band64[mel];
                    const int k0 = b.x & 0xffff, cnt = b.x >> 16;
                    const double* w = wts64 + b.y;
                    const double* pr = pp + fr * kPowStride + k0;
                    double acc = 0.0;
                    int jj = 0;
                    for (; jj + 4 <= cnt; jj += 4) {
                        const double w0 = w[jj], w1 = w[jj + 1], w2 = w[jj + 2], w3 = w[jj + 3];
                        const double p0 = pr[jj], p1 = pr[jj + 1], p2 = pr[jj + 2], p3 = pr[jj + 3];
                        acc += w0 * p0;
                        acc += w1 * p1;
                        acc += w2 * p2;
                        acc += w3 * p3;
                    }
                    for (; jj < cnt; ++jj) acc += w[jj] * pr[jj];
                    const int t = fa + fr;
                    if (t < T) out_clip[(unsigned)(t * n_mels + mel)] = (float)log_tab64(fmax(acc, 1.192092955078125e-07), ltab);
                }
            }
        };
        // one register set in the source: the next pair's samples are requested after the window stage has consumed
        // this pair's (the compiler lands them in temporaries and copies at the end of the iteration; alternating two
        // explicit sets instead spilled 17 loop-invariant LDS addresses and was slower, 0.231 vs 0.188 ms)
        int j = 0;
        int fa = pair_frame(0);
        if (fa < f_run_end && !prefetched) load_pair(x, T, n_valid, fa, xa0, xp0, xb0, xq0);
        prefetched = false;
        while (fa < f_run_end) {
            const int fa_next = pair_frame(j + 1);
            do_pair(fa, fa_next, xa0, xp0, xb0, xq0, xa0, xp0, xb0, xq0);
            fa = fa_next;
            ++j;
        }
        const int after_next = min(next_item + n_workers, n_items);
        item = next_item;
        clip_cur = clip_nx;
        run_cur = run_nx;
        len_cur = len_nx;
        off_cur = off_nx;
        next_item = after_next;
        if (next_item < n_items) kaldi_decode_item(items, next_item, clip_nx, run_nx);
    }
}

// float64 mode: per-clip per-bin mean and standard deviation exactly as HF's utterance_cmvn computes them
// (HF:...speech_to_text.py:150-155): NumPy reduces a (T, n_mels) float32 array over axis 0 sequentially in t with a
// float32 accumulator, and its std is sqrt(sum((x' - sum(x') / T)^2) / T) of x' = x - mean, all float32.  For long
// clips that mean is 1e-5 away from the exact one and bins that hardly vary over the clip (std ~ 0.01) amplify such
// differences a hundredfold, so matching HF to 1e-4 on every cell means doing the same additions in the same order.
// One thread per (clip, mel bin), three dependent passes over the clip's raw log-mel column; loads are batched 16 deep.
// stats (as float32): [clip][0][mel] = mean, [clip][1][mel] = std, inside the clip's 2 * n_mels doubles of workspace.
__global__ __launch_bounds__(64) void kaldi_stats_seq_kernel(
    const float* __restrict__ out, const int32_t* __restrict__ lengths, float* __restrict__ stats_f32,
    int batch, int n_mels, int t_max, int max_frames, int flags) {
#pragma clang fp contract(off)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * n_mels) return;
    const int clip = idx / n_mels, mel = idx - clip * n_mels;
    const int T = kaldi_num_frames(lengths[clip], max_frames, t_max);
    float* st = stats_f32 + (size_t)clip * 4 * n_mels;          // the clip's region is 2 * n_mels doubles
    if (T < 1) {
        st[mel] = 0.0f;
        st[n_mels + mel] = 1.0f;
        return;
    }
    const float* __restrict__ p = out + (size_t)clip * t_max * n_mels + mel;
    // two batches of kDepth loads in flight: the adds of one batch run under the loads of the other (each pass is
    // one dependent chain of T additions per thread, and a clip's 80 columns are all the parallelism there is)
    constexpr int kDepth = 32;
    auto sweep = [&](auto&& term) {
        float acc = 0.0f;
        float a[kDepth], b[kDepth];
        auto load = [&](float (&v)[kDepth], int t0) {
#pragma unroll
            for (int i = 0; i < kDepth; ++i) v[i] = p[(size_t)min(t0 + i, T - 1) * n_mels];   // unconditional: no branch
        };
        auto consume = [&](float (&v)[kDepth], int t0) {
#pragma unroll
            for (int i = 0; i < kDepth; ++i) v[i] = (t0 + i < T) ? term(v[i]) : 0.0f;    // independent of the chain
#pragma unroll
            for (int i = 0; i < kDepth; ++i) acc = acc + v[i];                           // acc + 0.0f is exact
        };
        load(a, 0);
        for (int t0 = 0; t0 < T; t0 += 2 * kDepth) {
            load(b, t0 + kDepth);
            consume(a, t0);
            load(a, t0 + 2 * kDepth);
            consume(b, t0 + kDepth);
        }
        return acc;
    };
    const float fT = (float)T;
    // normalize_means / normalize_vars individually off (HF:...speech_to_text.py:141-163): without the mean step the
    // std is taken of x itself (x - 0.0f == x), without the variance step nothing is divided (x / 1.0f == x)
    float mean = 0.0f, sd = 1.0f;
    if (!(flags & STTFE_KALDI_NO_MEAN)) mean = sweep([](float x) { return x; }) / fT;      // x.mean(axis=0)
    if (!(flags & STTFE_KALDI_NO_VAR)) {
        const float arrmean = sweep([mean](float x) { return x - mean; }) / fT;            // _var: mean of x - mean
        const float ssq = sweep([mean, arrmean](float x) {
            const float d = (x - mean) - arrmean;
            return d * d;
        });
        sd = sqrtf(ssq / fT);
    }
    st[mel] = mean;
    st[n_mels + mel] = sd;
}

// out[b][t][m] <- (x - mean_m) / std_m for t < T_b, 0 for padded rows; masks <- 0; attention mask.
__global__ __launch_bounds__(256) void kaldi_cmvn_kernel(
    float* __restrict__ out, const int32_t* __restrict__ lengths, const double* __restrict__ stats,
    int n_mels, int t_max, int max_frames, int flags, void* __restrict__ attn, int attn_bytes,
    const int32_t* __restrict__ mask_tbl, int n_masks, const KaldiTablesHeader* __restrict__ tab) {
    const int clip = blockIdx.y;
    if (tab->magic != kKaldiMagic || tab->n_mels != n_mels) {
        // tables built for another extractor: NaN everywhere instead of uninitialised memory (see sttfe.h)
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < t_max * n_mels; e += gridDim.x * blockDim.x)
            out[(size_t)clip * t_max * n_mels + e] = __builtin_nanf("");
        return;
    }
    const int T = kaldi_num_frames(lengths[clip], max_frames, t_max);
    float* __restrict__ o = out + (size_t)clip * t_max * n_mels;
    const double* st = stats + (size_t)clip * 2 * n_mels;
    int m_axis[STTFE_MAX_MASKS], m_s[STTFE_MAX_MASKS], m_e[STTFE_MAX_MASKS];
    const int nm = mask_tbl ? (n_masks > STTFE_MAX_MASKS ? STTFE_MAX_MASKS : n_masks) : 0;
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
    if ((flags & STTFE_KALDI_F64) && (int)threadIdx.x < n_mels) {       // precomputed by kaldi_stats_seq_kernel
        const float* stf = reinterpret_cast<const float*>(st);
        s_mean[threadIdx.x] = stf[threadIdx.x];
        s_std[threadIdx.x] = stf[n_mels + threadIdx.x];
    } else if ((int)threadIdx.x < n_mels) {
        const double inv_n = T > 0 ? 1.0 / (double)T : 0.0;
        const double mean = st[threadIdx.x] * inv_n;
        double var = st[n_mels + threadIdx.x] * inv_n - mean * mean;
        var = var > 0.0 ? var : 0.0;
        s_mean[threadIdx.x] = (flags & STTFE_KALDI_NO_MEAN) ? 0.0f : (float)mean;
        s_std[threadIdx.x] = (flags & STTFE_KALDI_NO_VAR) ? 1.0f : (float)sqrt(var);
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

// float64 mode, fused form of the two kernels above for the common shapes (n_mels a multiple of G, a clip's G columns
// fit in LDS): a workgroup owns G mel columns of one clip, reads them ONCE into an LDS panel [t][G], lanes 0..G-1 run
// the three sequential float32 passes of kaldi_stats_seq_kernel on the panel (same additions in the same order), and
// all threads then normalise the panel into the output (masks, zero padding and the attention mask as in
// kaldi_cmvn_kernel).  One read and one write of the features instead of four reads and one write.
template <int G>
__global__ __launch_bounds__(256) void kaldi_cmvn_seq_kernel(
    float* __restrict__ out, const int32_t* __restrict__ lengths, int batch, int n_mels, int t_max, int max_frames,
    void* __restrict__ attn, int attn_bytes, const int32_t* __restrict__ mask_tbl, int n_masks,
    const KaldiTablesHeader* __restrict__ tab, int flags) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float panel[];      // [T][G]
    __shared__ float s_mean[G], s_std[G];
    constexpr int Q = G / 4;                                            // float4 groups per panel row
    // workgroup -> (clip, column group): consecutive workgroup ids go round the 8 XCDs, so the groups of ONE clip are
    // given ids 8 apart -- they run at the same time on the same XCD and the G-float pieces they read from and write
    // to each output row meet in one L2.  (Measured neutral on config 2.  Round 3, timing-only variants on one box: with
    // BOTH the panel load and the output store made contiguous the kernel takes 0.0635 instead of 0.0653 ms -- the 32-byte
    // pieces are not what it waits for; without the chain it takes 0.041 ms: see DESIGN.md 4.3b.)
    const int tid = threadIdx.x, n_groups = n_mels / G;
    const int slot = blockIdx.x >> 3, clip = (slot / n_groups) * 8 + (blockIdx.x & 7);
    if (clip >= batch) return;
    const int g0 = (slot % n_groups) * G;
    float* __restrict__ o = out + (size_t)clip * t_max * n_mels + g0;
    if (tab->magic != kKaldiMagic || tab->n_mels != n_mels) {
        // tables built for another extractor: NaN everywhere instead of uninitialised memory (see sttfe.h)
        for (int i = tid; i < t_max * G; i += 256) o[(size_t)(i / G) * n_mels + (i % G)] = __builtin_nanf("");
        return;
    }
    const int T = kaldi_num_frames(lengths[clip], max_frames, t_max);
    for (int base = tid; base < T * Q; base += 8 * 256) {               // eight loads in flight per thread
        float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + 256 * k, t = i / Q, c = i - t * Q;
            if (i < T * Q) r[k] = *reinterpret_cast<const float4*>(o + (size_t)t * n_mels + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = base + 256 * k;
            if (i < T * Q) reinterpret_cast<float4*>(panel)[i] = r[k];
        }
    }
    __syncthreads();
    // The chain runs in ONE wave of the workgroup (which one rotates with the workgroup index, so that the chains of the
    // workgroups that share a CU land on different SIMDs instead of queueing on the first), on all 64 lanes: lane
    // 8 c + r holds the term of row t0 + r of column c of an eight-row block, and lane (c, 0) adds the eight terms in row
    // order with the DPP operand form `v_add_f32 acc, v row_shl:k, acc` -- the same float32 additions in the same order
    // as NumPy's axis-0 reduction (the other lanes add up values nobody reads).  One LDS read and one term per EIGHT rows
    // and one dependent instruction per row, where eight lanes that each walked their own column spent a read, a term
    // and an addition on every row (kaldi_cmvn_seq_kernel 0.077 -> 0.065 ms on config 2).  The accumulator is the plain
    // operand: a DPP read of a register the previous instruction wrote would cost two wait states per row.
    const int chain_lane = tid - 64 * (slot & 3);
    static_assert(G == 8, "lane = 8 c + r: eight columns of eight-row blocks");
    if (chain_lane >= 0 && chain_lane < 64) {
        const int c = chain_lane >> 3, r = chain_lane & 7;
        const float* cell = panel + r * G + c;                        // + t0 * G: 64 consecutive floats per block, no conflicts
        // v of lane (c, k) brought to lane (c, 0): row_shl:k inside the 16-lane DPP row that holds columns 2 i, 2 i + 1
        auto from_row = [](float v, auto k) {
            return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + decltype(k)::value, 0xf, 0xf, true));
        };
        const int n_full = T >> 3, rem = T & 7;
        auto sweep = [&](auto&& term) {
            float acc = 0.0f;
            auto block = [&](float v) {
                acc = acc + v;
                acc = acc + from_row(v, std::integral_constant<int, 1>{});
                acc = acc + from_row(v, std::integral_constant<int, 2>{});
                acc = acc + from_row(v, std::integral_constant<int, 3>{});
                acc = acc + from_row(v, std::integral_constant<int, 4>{});
                acc = acc + from_row(v, std::integral_constant<int, 5>{});
                acc = acc + from_row(v, std::integral_constant<int, 6>{});
                acc = acc + from_row(v, std::integral_constant<int, 7>{});
            };
            int b = 0;
            // four blocks' rows requested ahead of the four blocks' additions
            float xa[4], xb[4];
            auto load4 = [&](int b0, float (&x)[4]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = cell[(b0 + i) * 8 * G];
            };
            if (n_full >= 4) load4(0, xa);
            while (b + 4 <= n_full) {
                if (b + 8 <= n_full) load4(b + 4, xb);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) block(term(xa[i]));
                b += 4;
                if (b + 4 > n_full) break;
                if (b + 8 <= n_full) load4(b + 4, xa);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) block(term(xb[i]));
                b += 4;
            }
            for (; b < n_full; ++b) block(term(cell[b * 8 * G]));
            // the last rows: the missing ones add + 0.0f, which changes no sum (the sums here are never - 0.0f)
            if (rem) block(r < rem ? term(cell[n_full * 8 * G]) : 0.0f);
            return __shfl(acc, 8 * c, 64);                            // every lane of column c gets the column's sum
        };
        float mean = 0.0f, sd = 1.0f;
        if (T > 0) {
            const float fT = (float)T;
            // (normalize_means / normalize_vars individually off: see kaldi_stats_seq_kernel)
            if (!(flags & STTFE_KALDI_NO_MEAN)) mean = sweep([](float x) { return x; }) / fT;             // x.mean(axis=0)
            if (!(flags & STTFE_KALDI_NO_VAR)) {
                const float arrmean = sweep([mean](float x) { return x - mean; }) / fT;       // _var: mean of x - mean
                const float ssq = sweep([mean, arrmean](float x) {
                    const float d = (x - mean) - arrmean;
                    return d * d;
                });
                sd = sqrtf(ssq / fT);
            }
        }
        if (r == 0) {
            s_mean[c] = mean;
            s_std[c] = sd;
        }
    }
    __syncthreads();
    int m_axis[STTFE_MAX_MASKS], m_s[STTFE_MAX_MASKS], m_e[STTFE_MAX_MASKS];
    const int nm = mask_tbl ? (n_masks > STTFE_MAX_MASKS ? STTFE_MAX_MASKS : n_masks) : 0;
    for (int i = 0; i < nm; ++i) {
        const int32_t* m = mask_tbl + ((size_t)clip * n_masks + i) * 3;
        m_axis[i] = m[0];
        m_s[i] = m[1];
        m_e[i] = m[2];
    }
    auto norm = [&](float x, int t, int c) {
        float y = (x - s_mean[c]) / s_std[c];
        for (int i = 0; i < nm; ++i) {
            const int idx = (m_axis[i] == STTFE_AXIS_TIME) ? t : g0 + c;
            if (idx >= m_s[i] && idx < m_e[i]) y = 0.0f;
        }
        return y;
    };
    for (int i = tid; i < t_max * Q; i += 256) {
        const int t = i / Q, c = (i - t * Q) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < T) {
            v = reinterpret_cast<const float4*>(panel)[i];
            v.x = norm(v.x, t, c);
            v.y = norm(v.y, t, c + 1);
            v.z = norm(v.z, t, c + 2);
            v.w = norm(v.w, t, c + 3);
        }
        *reinterpret_cast<float4*>(o + (size_t)t * n_mels + c) = v;
    }
    if (attn && g0 == 0) {
        for (int t = tid; t < t_max; t += 256) {
            if (attn_bytes == 8) reinterpret_cast<int64_t*>(attn)[(size_t)clip * t_max + t] = t < T ? 1 : 0;
            else reinterpret_cast<int32_t*>(attn)[(size_t)clip * t_max + t] = t < T ? 1 : 0;
        }
    }
}

// float64 mode, clips of up to 8 NB frames: kaldi_cmvn_seq_kernel's arithmetic with the columns held in REGISTERS.  One
// wave owns 8 mel columns of one clip; lane 8 c + r keeps rows r, r + 8, r + 16, ... of column c (T / 8 registers), which
// is exactly the operand the chain wants: block b's term of lane (c, r) is a function of register b, and lane (c, 0) adds
// the eight terms of a block in row order through the DPP operand (see kaldi_cmvn_seq_kernel).  No LDS panel, so nothing
// limits residency but registers (NB = 128: 3 waves per SIMD, 3072 waves on the chip -- config 2's 2560 column groups run
// in ONE round, where the LDS kernel's 32 KB panels forced two rounds that each paid for their longest chain), no
// barrier, and the loads / chains / stores of different waves overlap freely.
//   * Memory side: a lane that fetched its own cells would issue T / 8 four-byte loads and as many stores (the address
//     unit takes a wave instruction four lanes a cycle whatever its width: the loads alone were 0.027 of this kernel's
//     0.054 ms on config 2).  Instead 16 bytes per lane: lane 2 t + q moves columns 4 q .. 4 q + 3 of row t of a 32-row
//     piece, loads as `global_load_lds_dwordx4` straight into a wave-private staging slot (two slots of 128 rows: the
//     next chunk is in flight while this one is dealt to the (c, r) registers with conflict-free ds_read), stores from
//     the same slots (ds_write by (c, r), ds_read_b128 + global_store_dwordx4 by (t, q)).
//   * The division by the deviation is the compiler's own IEEE sequence with the parts that depend on the divisor alone
//     (reciprocal and its refinement) taken out of the loop -- valid where v_div_scale would not scale, which a
//     wave-uniform test on the deviation and the mean guarantees (see `fast`); otherwise the plain division.
// Same additions in the same order, same quotients: the three forms (this, the LDS form, the split pair) are
// bit-identical (tests/test_kaldi_gpu.py).
template <int G, int NB>
__global__ __launch_bounds__(64, NB <= 128 ? 12 : 8) void kaldi_cmvn_reg_kernel(
    float* __restrict__ out, const int32_t* __restrict__ lengths, int batch, int n_mels, int t_max, int max_frames,
    void* __restrict__ attn, int attn_bytes, const int32_t* __restrict__ mask_tbl, int n_masks,
    const KaldiTablesHeader* __restrict__ tab, int flags) {
#pragma clang fp contract(off)
    // G = 8 columns in blocks of R = 8 rows, or (longer clips: T / R registers) G = 4 columns in blocks of R = 16 rows
    constexpr int R = 64 / G, Q = G / 4, kPieceRows = 64 / Q, kChunkRows = 4 * kPieceRows, kChunks = NB / 16;
    static_assert((G == 8 || G == 4) && NB % 16 == 0 && kChunkRows == 16 * R, "a chunk = four 1 KB pieces = 16 blocks");
    __shared__ __attribute__((aligned(16))) float stage[2][1024];
    const int lane = threadIdx.x, c = lane / R, r = lane % R;         // chain side: (column, row of the block)
    const int tq = lane / Q, q = lane % Q;                            // memory side: (row of the piece, column quad)
    // workgroup -> (clip, column group) as in kaldi_cmvn_seq_kernel: the groups of one clip run on one XCD
    const int n_groups = n_mels / G;
    const int slot = blockIdx.x >> 3, clip = (slot / n_groups) * 8 + (blockIdx.x & 7);
    if (clip >= batch) return;
    const int g0 = (slot % n_groups) * G;
    const size_t row = (size_t)n_mels;
    float* __restrict__ base = out + (size_t)clip * t_max * n_mels + g0;       // (row 0, column g0)
    float* __restrict__ mine = base + 4 * q;                                  // memory side, + row * n_mels
    if (tab->magic != kKaldiMagic || tab->n_mels != n_mels) {
        const float nan = __builtin_nanf("");                                 // see sttfe.h
        for (int t = tq; t < t_max; t += kPieceRows) *reinterpret_cast<f32x4*>(mine + (size_t)t * row) = f32x4{nan, nan, nan, nan};
        return;
    }
    const int T = kaldi_num_frames(lengths[clip], max_frames, t_max);
    const int n_full = T / R, rem = T % R;
    const int n_chunks_in = (T + kChunkRows - 1) / kChunkRows;
    // the rows left over after the full blocks, in a register of their own (requested first: the waits below count DMAs)
    const float xt = (r < rem) ? base[(size_t)(n_full * R + r) * row + c] : 0.0f;
    auto dma_chunk = [&](int ch, float* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = min(kChunkRows * ch + kPieceRows * i + tq, t_max - 1);      // (rows past the clip are never used)
            __builtin_amdgcn_global_load_lds(mine + (size_t)t * row, dst + 256 * i, 16, 0, 0);
        }
    };
    if (n_chunks_in > 0) dma_chunk(0, stage[0]);
    if (n_chunks_in > 1) dma_chunk(1, stage[1]);
    float x[NB];
#pragma unroll
    for (int ch = 0; ch < kChunks; ++ch) {
        if (ch < n_chunks_in) {
            // loads land in order: chunk ch is in LDS once at most the four DMAs of chunk ch + 1 are outstanding
            // (nothing else orders a ds_read behind a pending DMA)
            if (ch + 1 < n_chunks_in) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const float* sl = stage[ch & 1] + r * G + c;                      // 64 consecutive floats per block: no conflicts
#pragma unroll
            for (int bb = 0; bb < 16; ++bb) x[16 * ch + bb] = sl[bb * 64];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // the slot is read before it is refilled
            if (ch + 2 < n_chunks_in) dma_chunk(ch + 2, stage[ch & 1]);
        }
    }
    auto from_row = [](float v, auto k) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + decltype(k)::value, 0xf, 0xf, true));
    };
    auto sweep = [&](auto&& term) {
        float acc = 0.0f;
        auto block = [&](float v) {
            acc = acc + v;
            acc = acc + from_row(v, std::integral_constant<int, 1>{});
            acc = acc + from_row(v, std::integral_constant<int, 2>{});
            acc = acc + from_row(v, std::integral_constant<int, 3>{});
            acc = acc + from_row(v, std::integral_constant<int, 4>{});
            acc = acc + from_row(v, std::integral_constant<int, 5>{});
            acc = acc + from_row(v, std::integral_constant<int, 6>{});
            acc = acc + from_row(v, std::integral_constant<int, 7>{});
            if constexpr (R == 16) {
                acc = acc + from_row(v, std::integral_constant<int, 8>{});
                acc = acc + from_row(v, std::integral_constant<int, 9>{});
                acc = acc + from_row(v, std::integral_constant<int, 10>{});
                acc = acc + from_row(v, std::integral_constant<int, 11>{});
                acc = acc + from_row(v, std::integral_constant<int, 12>{});
                acc = acc + from_row(v, std::integral_constant<int, 13>{});
                acc = acc + from_row(v, std::integral_constant<int, 14>{});
                acc = acc + from_row(v, std::integral_constant<int, 15>{});
            }
        };
#pragma unroll
        for (int k = 0; k < kChunks; ++k) {
            if (16 * k < n_full) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (16 * k + i < n_full) block(term(x[16 * k + i]));
            }
        }
        // the last rows: the missing ones add + 0.0f, which changes no sum (the sums here are never - 0.0f)
        if (rem) block(r < rem ? term(xt) : 0.0f);
        return __shfl(acc, R * c, 64);                                // every lane of column c gets the column's sum
    };
    float mean = 0.0f, sd = 1.0f;
    if (T > 0) {
        const float fT = (float)T;
        // (normalize_means / normalize_vars individually off: see kaldi_stats_seq_kernel)
        if (!(flags & STTFE_KALDI_NO_MEAN)) mean = sweep([](float v) { return v; }) / fT;                 // x.mean(axis=0)
        if (!(flags & STTFE_KALDI_NO_VAR)) {
            const float arrmean = sweep([mean](float v) { return v - mean; }) / fT;           // _var: mean of x - mean
            const float ssq = sweep([mean, arrmean](float v) {
                const float d = (v - mean) - arrmean;
                return d * d;
            });
            sd = sqrtf(ssq / fT);
        }
    }
    // (x - mean) / sd.  The compiler's float32 division is: sb = div_scale(sd), sn = div_scale(n), r0 = rcp(sb),
    // rp = fma(fma(-sb, r0, 1), r0, r0), q = sn rp, q1 = fma(fma(-sb, q, sn), rp, q), fma(fma(-sb, q1, sn), rp, q1), then
    // div_fmas / div_fixup, which undo the scaling and patch zero / infinite / NaN operands.  v_div_scale leaves both
    // operands alone unless the divisor is denormal or huge, the numerator below 2^-103 or the exponents 96 apart: with
    // 2^-30 <= sd, |mean| <= 2^30 in every lane a non-zero numerator x - mean is at least 2^-55 (x within a factor two
    // of the mean: the difference is a multiple of the smaller ulp; otherwise it is at least half the mean) and at most
    // about sd sqrt(T) <= 2^35 (each squared deviation is a term of the sum under the root; with normalize_vars off
    // sd is 1 and both sequences return the numerator itself), and a zero numerator is + 0 and stays + 0 through the
    // sequence -- so there the reciprocal part is taken out of the loop and five instructions per cell remain, with
    // the bits of the full sequence.
    const float kLo = 9.31322574615478515625e-10f, kHi = 1073741824.0f;        // 2^-30, 2^30
    const bool fast = __all(sd >= kLo && sd <= kHi && fabsf(mean) >= kLo && fabsf(mean) <= kHi);
    const float r0 = __builtin_amdgcn_rcpf(sd);
    const float rp = __builtin_fmaf(__builtin_fmaf(-sd, r0, 1.0f), r0, r0);
    auto norm = [&](float xv, auto is_fast) {
        const float n = xv - mean;
        if constexpr (decltype(is_fast)::value) {
            const float q0 = n * rp;
            const float q1 = __builtin_fmaf(__builtin_fmaf(-sd, q0, n), rp, q0);
            return __builtin_fmaf(__builtin_fmaf(-sd, q1, n), rp, q1);
        } else {
            return n / sd;
        }
    };
    // SpecAugment: a masked mel column is zero from top to bottom (chain side), a masked frame on the memory side
    const int nm = mask_tbl ? (n_masks > STTFE_MAX_MASKS ? STTFE_MAX_MASKS : n_masks) : 0;
    bool col_masked = false;
    for (int i = 0; i < nm; ++i) {
        const int32_t* m = mask_tbl + ((size_t)clip * n_masks + i) * 3;
        if (m[0] != STTFE_AXIS_TIME && g0 + c >= m[1] && g0 + c < m[2]) col_masked = true;
    }
    auto store_all = [&](auto is_fast) {
        const float yt = (r < rem && !col_masked) ? norm(xt, is_fast) : 0.0f;
#pragma unroll
        for (int ch = 0; ch < kChunks; ++ch) {
            if (kChunkRows * ch < t_max) {
                float* sl = stage[ch & 1] + r * G + c;                        // (LDS runs a wave's operations in order)
#pragma unroll
                for (int bb = 0; bb < 16; ++bb) {
                    const int b = 16 * ch + bb;
                    float y = 0.0f;                                           // rows past the clip: padding
                    if (b < n_full) y = col_masked ? 0.0f : norm(x[b], is_fast);
                    else if (b == n_full) y = yt;
                    sl[bb * 64] = y;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = kChunkRows * ch + kPieceRows * i + tq;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stage[ch & 1] + 256 * i + 4 * lane);
                    for (int k = 0; k < nm; ++k) {
                        const int32_t* m = mask_tbl + ((size_t)clip * n_masks + k) * 3;
                        if (m[0] == STTFE_AXIS_TIME && t >= m[1] && t < m[2]) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    if (t < t_max) *reinterpret_cast<f32x4*>(mine + (size_t)t * row) = v;
                }
            }
        }
    };
    if (fast) store_all(std::true_type{});
    else store_all(std::false_type{});
    if (attn && g0 == 0) {
        for (int t = lane; t < t_max; t += 64) {
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
    if (batch < 0 || n_mels < 1 || n_mels > 128 || t_max < 0 || n_masks < 0 || n_masks > STTFE_MAX_MASKS) return STTFE_E_INVALID_ARG;
    if (flags & ~(STTFE_KALDI_NO_CMVN | STTFE_KALDI_F64 | STTFE_KALDI_NO_MEAN | STTFE_KALDI_NO_VAR)) return STTFE_E_INVALID_ARG;
    if ((flags & STTFE_KALDI_NO_MEAN) && (flags & STTFE_KALDI_NO_VAR)) flags |= STTFE_KALDI_NO_CMVN;     // nothing left to do
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
    // The float64 kernel's items belong to waves and are 1 ... 8 frame pairs long: with equal-sized items the kernel ends
    // when the waves with one item more than the others do, so a batch that would give a wave only a few long items is
    // cut finer, while an item switch costs ~0.6 us -- about five items per wave measured best (config 2's clip lengths,
    // 64 / 256 / 1024 clips: 1-2 / 4 / 8-16 pairs, profiles/r3_kaldi_items.txt).  STTFE_KALDI_ITEM_PAIRS: tuning knob.
    auto item_frames64 = [&](int clips) {
        static const int ip_env = [] {
            const char* e = getenv("STTFE_KALDI_ITEM_PAIRS");
            const int v = e ? atoi(e) : 0;
            return (v >= 1 && v <= 64) ? v : 0;
        }();
        const long waves64 = (long)kaldi64::kWaves * kaldi64::kWgPerCu * current_device_cus();
        const long pairs = (long)clips * ((t_max + 1) / 2) / 2;            // half of the padded batch: the lengths are on the device
        long per = pairs / (5 * waves64);
        per = per < 1 ? 1 : (per > 8 ? 8 : per);
        return 2 * (ip_env ? ip_env : (int)per);
    };
    const int tile_frames = kaldi::kTileFrames;
    const int n_tiles = (t_max + tile_frames - 1) / tile_frames;
    const int resident = 2 * current_device_cus();
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
        // (float32 kernel: 1 / 2 / 3 / 4 tiles per item 0.110 / 0.106 / 0.112 / 0.116 ms on config 2; the float64 kernel's
        // items are sized above and are one "tile" each)
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
        if (flags & STTFE_KALDI_F64) {
            // at most kKaldiPlanClips clips per launch: the kernel strides through the COMPACTED item space
            const int res64 = kaldi64::kWgPerCu * current_device_cus();
            for (int c0 = 0; c0 < batch; c0 += kKaldiPlanClips) {
                const int bc = batch - c0 < kKaldiPlanClips ? batch - c0 : kKaldiPlanClips;
                const int frames = item_frames64(bc);
                const int runs64 = (t_max + frames - 1) / frames;
                const int items64 = runs64 * bc;
                const int want64 = (items64 + kaldi64::kWaves - 1) / kaldi64::kWaves;      // items belong to waves
                hipLaunchKernelGGL(kaldi_fbank64_kernel, dim3(want64 < res64 ? want64 : res64), dim3(kaldi64::kThreads), 0,
                                   stream, pcm, offsets + c0, lengths + c0, n_mels, t_max, max_frames, tab,
                                   out + (size_t)c0 * t_max * n_mels, runs64, items64, bc, frames);
            }
        } else {
            hipLaunchKernelGGL(kaldi_fbank_kernel, dim3(grid), dim3(kaldi::kThreads), 0, stream, pcm, offsets,
                               lengths, n_mels, t_max, max_frames, tab, out, stats, tiles_per_block, n_runs, n_items, ticket, batch);
        }
    }
    {
        ScopedTiming t(kTimeKaldiCmvn, stream);
        constexpr int kG = 8;      // columns per workgroup: 4 -> 0.127, 8 -> 0.072, 16 -> 0.078 ms (config 2)
        const bool seq_split = getenv("STTFE_KALDI_SPLIT_CMVN") != nullptr;     // A/B and test knob: the two-kernel form
        const size_t panel_bytes = (size_t)t_max * kG * sizeof(float);
        const bool f64_cmvn = (flags & STTFE_KALDI_F64) && !(flags & STTFE_KALDI_NO_CMVN);
        const bool lds_form = getenv("STTFE_KALDI_LDS_CMVN") != nullptr;        // A/B and test knob: the LDS-panel form
        // columns in registers: 8 columns x blocks of 8 rows up to 1024 frames, 4 columns x blocks of 16 rows up to 3072
        const bool reg_ok = f64_cmvn && !seq_split && !lds_form && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        auto launch_reg = [&](auto kern, int g) {
            hipLaunchKernelGGL(kern, dim3((batch + 7) / 8 * 8 * (n_mels / g)), dim3(64), 0, stream, out, lengths, batch,
                               n_mels, t_max, max_frames, attn_mask, attn_mask_bytes, mask_tbl, n_masks, tab, flags);
        };
        if (reg_ok && n_mels % 8 == 0 && t_max <= 8 * 128) {
            launch_reg(kaldi_cmvn_reg_kernel<8, 128>, 8);
        } else if (reg_ok && n_mels % 4 == 0 && t_max <= 16 * 128) {
            launch_reg(kaldi_cmvn_reg_kernel<4, 128>, 4);
        } else if (reg_ok && n_mels % 4 == 0 && t_max <= 16 * 192) {
            launch_reg(kaldi_cmvn_reg_kernel<4, 192>, 4);
        } else if (f64_cmvn && !seq_split && n_mels % kG == 0 && panel_bytes <= 64 * 1024 &&
            (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            hipLaunchKernelGGL(kaldi_cmvn_seq_kernel<kG>, dim3((batch + 7) / 8 * 8 * (n_mels / kG)), dim3(256), panel_bytes,
                               stream, out, lengths, batch, n_mels, t_max, max_frames, attn_mask, attn_mask_bytes,
                               mask_tbl, n_masks, tab, flags);
        } else {
            if (f64_cmvn)
                hipLaunchKernelGGL(kaldi_stats_seq_kernel, dim3((batch * n_mels + 63) / 64), dim3(64), 0, stream, out,
                                   lengths, reinterpret_cast<float*>(workspace), batch, n_mels, t_max, max_frames, flags);
            int cx = (t_max * n_mels + 255) / 256;
            if (cx > 64) cx = 64;
            hipLaunchKernelGGL(kaldi_cmvn_kernel, dim3(cx, batch), dim3(256), 0, stream, out, lengths, stats, n_mels,
                               t_max, max_frames, flags, attn_mask, attn_mask_bytes, mask_tbl, n_masks, tab);
        }
    }
    if (hip_failed(hipGetLastError(), "kaldi kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}
