// Whisper log-mel front-end for gfx950 (MI355X): framing + Hann + 400-point STFT + mel + log10,
// then the per-clip clamp(max - 8) / (x + 4) / 4 fix-up (+ SpecAugment masks, + 30 s zero padding).
//
// Replaces HF WhisperFeatureExtractor._torch_extract_fbank_features
// (HF:models/whisper/feature_extraction_whisper.py:135-168) as reached from the reference at
// src/data.py:338-343, inference.py:148-152 and src/processors.py:180-221.
//
// Kernel 1 (stft_mel): persistent 256-thread workgroups (4 waves, 253 VGPRs, 65 KB LDS -> two per CU, grid = 2 x CUs).
//   A workgroup stages window, twiddles, mel weights and the unit plan in LDS once and then takes work items --
//   (run of consecutive 48-frame tiles, clip) -- from a ticket counter in the workspace; ticket, length / offset and
//   the first PCM tile of the next item are requested one to two items ahead.  Per tile:
//   * every wave makes two passes of 3 frame pairs (12 frames).  Two real frames a, b are packed as
//     z = a + i b and transformed by ONE complex 400-point DFT, factored 20 x 20: lane (p, j) of a pair
//     runs a register-resident 20-point DFT over n1 of z[20 n1 + j] (twiddle-free 4 x 5 prime-factor form in the
//     packed-float32 forms of fft_pk.h), multiplies by W_400^(j k1) (the lane's twiddles stay in registers),
//     exchanges through a wave-private LDS slab ([pair][k1][n2], row stride 22 complex: b64 writes, ten
//     conflict-free b128 reads per lane) and runs the second 20-point DFT over n2.  Lane j then holds
//     Z[j + 20 k2]; the spectra of a and b are separated with one cross-lane exchange (lane j <-> lane 20 - j),
//     bins 0..200 kept.  The two passes are interleaved, so the arithmetic of one covers the LDS round trips of
//     the other; the PCM of the next tile is requested right before the mel phase of the current one.
//   * |A|^2, |B|^2 go to the wave's own 12 rows of the [frame][bin] power panel (row stride 204 floats), which
//     aliases the wave's own slab: no workgroup barrier between FFT and panel.
//   * barrier; banded mel projection on the matrix cores: exact-f32 v_mfma_f32_16x16x4_f32 tiles
//     (16 mel rows x 16 frames) over only the bins each 16-row band of the triangular bank touches
//     (60 k-steps for 80 mels, 68 for 128); operands as one b128 per lane and group of four k-steps.  The
//     (mel tile, frame tile) units of a wave come from a plan: for the 80- and 128-bin extractors a compile-time
//     copy of it (fully unrolled units, immediate-offset operand reads; taken for whole, unmasked tiles when the
//     tables' plan matches), otherwise the host-built descriptors of tables.cpp; barrier.
//   * epilogue: log10(max(mel, 1e-10)) -> (x + 4) / 4 stored unclamped with the SpecAugment masks
//     applied from per-tile / per-clip bit masks; per-clip max and per-tile min of log10 go to the
//     workspace with one atomic per wave.
//   Tiles that lie entirely in the padding are left to kernel 2 (they are constant).
//   Three modes (kMode*): HF semantics, and the two of the reference's TorchWhisperFeatureExtractor.
// Kernel 1', the 80- and 128-bin extractors in HF mode (stft_mel12, whisper_logmel12.inc): one 768-thread workgroup per CU, eight
//   transform waves + four mel waves handing the power panel over through LDS counters; same arithmetic, same bits.  It
//   computes batches of up to 3072 clips alone (uniform or ragged, masks, per-clip scale); larger batches launch both
//   kernels and the lengths on the device decide which one works.  This file's kernel 1 keeps the torch modes, other bin
//   counts and ragged batches beyond 3072 clips.
// Kernel 2 (clamp_fill): 16-32 workgroups per clip, each owning 4-5 interleaved mel rows.  The zero-padding region is filled
//   with the final constant (write only); of the computed tiles only those whose recorded minimum is below
//   max - 8 are re-read and clamped to (max - 8 + 4) / 4 (everything else is already final).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "../../include/sttfe.h"
#include "fft_core.h"
#include "fft_pk.h"
#include "tables.h"
#include "whisper_mel_bands.h"
#include "timing.h"

namespace sttfe {

namespace wh {
constexpr int kThreads = 256;
constexpr int kWaves = 4;
constexpr int kTileFrames = 48;
constexpr int kFramesPerWave = 12;            // two passes of 3 pairs
constexpr int kHop = 160;
constexpr int kNfft = 400;
constexpr int kMaxSamples = 480000;           // 30 s at 16 kHz (HF:...whisper.py:91)
constexpr int kVarMaxSamples = 1 << 28;       // variable-length mode: the clip's own length counts (no 30 s cut: the
                                              // reference crops AFTER the maximum, src/processors.py:226-239)
// Which frames count and how clips end (reference src/processors.py:180-239 differs from HF in both):
constexpr int kModeHF = 0;          // pad / truncate to 30 s, last frame dropped before the per-clip max (HF)
constexpr int kModeTorchPadded = 1; // pad / truncate to 30 s, per-clip max over all 3001 frames, then crop (:206-219)
constexpr int kModeTorchVarlen = 2; // no padding: reflect at the clip's own end, 1 + n / 160 frames, max over them,
                                    // crop to 3000, frames past the clip's own <- padding_value (:222-239)
constexpr int kMaxTiles = 63;                 // ceil(3000 / 48)
constexpr int kKeysPerClip = 64;              // workspace: [0] = max key, [1 + t] = ~min key of tile t
constexpr int kRowStride = 22;                // complex units, [k1][n2] rows: 176 B, so a row reads as 10 x b128
constexpr int kPairStride = 20 * kRowStride;  // complex units per frame pair (880 floats == 48 mod 64 banks)
constexpr int kPStride = 204;                 // floats per frame row of the power panel: 16-byte rows, 12 n mod 64 banks -> conflict-free b128 operand reads
constexpr int kSlabFloats = 2640;             // per wave: 3 pairs x 880 floats of exchange slab, aliased by 12 panel rows
static_assert(3 * kPairStride * 2 <= kSlabFloats, "exchange slab of three frame pairs");
static_assert(kFramesPerWave * kPStride <= kSlabFloats, "a wave's panel rows must fit inside its own slab");
static_assert((kSlabFloats - kFramesPerWave * kPStride) % 64 == 0, "panel rows keep one bank pattern across waves");
static_assert(kSlabFloats % 4 == 0, "16-byte aligned slabs");
constexpr int kWinOff = kWaves * kSlabFloats;         // float offsets of the tables inside the LDS array
constexpr int kTwStride = 44;                 // floats per lane row of the twiddle table ([j][k1] complex, padded: conflict-free b128)
constexpr int kTwOff = kWinOff + 400;
constexpr int kAfragOff = kTwOff + 20 * kTwStride;               // banded mel weights in MFMA A-operand order
constexpr int kMaxAfragFloats = 4608;                 // 53 (80 mels) / 56 (128 mels) k-steps x 64 lanes
constexpr int kDescOff = kAfragOff + kMaxAfragFloats;     // per wave: 8 x {kstart, nsteps, aoff, mt | ft << 8}, then the unit counts
constexpr int kTicketOff = kDescOff + kWaves * kMaxUnitsPerWave * 4 + kWaves;    // 3 words: work-item hand-over slots
constexpr int kMaskOff = kTicketOff + 4;              // the current clip's SpecAugment masks: STTFE_MAX_MASKS x {axis, start, end}
constexpr int kPlanOff = kMaskOff + 3 * STTFE_MAX_MASKS;      // ragged batches: exclusive prefix of the clips' run counts
constexpr int kMaxPlanClips = 3072;                   // (12 KB; larger batches keep the dense run x clip item space)
constexpr int kLdsFloats = kPlanOff + kMaxPlanClips + 1;   // ~77 KB -> 2 workgroups per CU
}  // namespace wh

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned float_to_key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // order-preserving float -> uint, never 0 for finite f
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

// Wave-wide minimum without the LDS pipe: DPP swizzles inside each row of 16 lanes (quad swaps, half-row and row
// mirrors), then the four row results through scalar registers.  (Six dependent ds_bpermute steps cost ~700 cycles.)
__device__ __forceinline__ float wave_min(float v) {
    auto step = [](float x, auto ctrl) {
        const int y = __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xf, 0xf, false);
        return fminf(x, __int_as_float(y));
    };
    v = step(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{});     // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});     // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}

// Bits [0, width) of the result mark the indices base + i covered by a mask of the given axis.
__device__ __forceinline__ uint64_t mask_bits(const int32_t* m, int n_masks, int axis, int base, int width) {
    uint64_t bits = 0;
    for (int i = 0; i < n_masks; ++i) {
        if (m[3 * i] != axis) continue;
        int s = m[3 * i + 1] - base, e = m[3 * i + 2] - base;
        s = s < 0 ? 0 : s;
        e = e > width ? width : e;
        if (e > s) bits |= ((e - s >= 64) ? ~0ull : ((1ull << (e - s)) - 1ull)) << s;
    }
    return bits;
}

#ifdef STTFE_STAMPS
// Diagnostic build only: per-wave cycle sums of each phase of the tile loop (never in the product build).
__device__ unsigned long long* g_stamp_buf = nullptr;
#define STAMP(i)                                                                 \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        unsigned long long t_;                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                       \
        stamp_sum[i] += t_ - stamp_prev;                                         \
        stamp_prev = t_;                                                         \
    } while (0)
#else
#define STAMP(i)
#endif

// ---- static mel plans of the 80- and 128-bin extractors ---------------------------------------------------------
// tables.cpp derives the banded plan from the mel bank at run time; for n_mels == 80 and 128 (the two Whisper
// configurations) the result is always the one below (the kernel compares the blob's plan with it once and falls back
// to the generic unit loop otherwise).
// With the plan known at compile time a unit's MFMA sequence is fully unrolled: operand reads are ds_read_b128 with
// immediate offsets off two per-lane base registers, the first steps take an inline zero accumulator, the stores use
// one scalar base per unit + four per-lane offsets computed once per kernel, and no descriptor is read per tile.
namespace wh {
template <int NM> struct StaticPlan;
template <> struct StaticPlan<80> {
    static constexpr int n = 5;
    static constexpr int kstart[5] = {0, 12, 28, 56, 104}, nsteps[5] = {4, 8, 8, 16, 24}, aoff[5] = {0, 256, 768, 1280, 2304};
};
template <> struct StaticPlan<128> {
    static constexpr int n = 8;
    static constexpr int kstart[8] = {0, 8, 16, 28, 40, 60, 88, 124}, nsteps[8] = {4, 4, 4, 4, 8, 12, 12, 20},
                         aoff[8] = {0, 256, 512, 768, 1024, 1536, 2304, 3072};
};
template <int NM>
__device__ __forceinline__ bool plan_matches(const MelPlan& p) {
    bool ok = p.n_mtiles == StaticPlan<NM>::n;
#pragma unroll
    for (int mt = 0; mt < StaticPlan<NM>::n; ++mt)
        ok = ok && p.kstart[mt] == StaticPlan<NM>::kstart[mt] && p.nsteps[mt] == StaticPlan<NM>::nsteps[mt] &&
             p.aoff[mt] == StaticPlan<NM>::aoff[mt];
    return ok;
}
}  // namespace wh

// One (mel tile MT, frame tile FT) unit of a whole 48-frame tile of an NM-bin extractor.
//   a_lane : this lane's A-operand base (mel weights, LDS), b_lane : its B-operand row base of frame tile FT (panel)
//   out_u  : scalar pointer out_clip + t0 + MT * 16 * n_frames + FT * 16;  o_r : per-lane store offsets, r = 0..3
//   kMask  : the clip has SpecAugment masks -- tbits = the tile's time-mask bits of frame tile FT (16), mbits = the
//            clip's mel-mask bits of mel tile MT (16), both wave-uniform: a unit no mask touches (most of them) takes
//            the plain stores after one scalar test, the others write 0.0 over the masked cells
template <int NM, int MT, bool kMask>
__device__ __forceinline__ void mel_unit_static(const float* __restrict__ a_lane, const float* __restrict__ b_lane,
                                           float* __restrict__ out_u, unsigned (&o_r)[4], float mel_floor,
                                           float y_bias, float& run_max, float& tile_min, unsigned tbits, unsigned mbits,
                                           int lane) {
    using P = wh::StaticPlan<NM>;
    constexpr int KS = P::kstart[MT], NG = P::nsteps[MT] / 4, AO = P::aoff[MT];
    const f32x4* __restrict__ a = reinterpret_cast<const f32x4*>(a_lane + AO);
    const f32x4* __restrict__ b = reinterpret_cast<const f32x4*>(b_lane + KS);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#if STTFE_FLOOR == 9 || STTFE_FLOOR == 11
        const f32x4 av = {mel_floor, y_bias, mel_floor, y_bias}, bv = {y_bias, mel_floor, y_bias, mel_floor};   // floor build: no operand reads
#else
        const f32x4 av = a[g * 64], bv = b[4 * g];
#endif
#if STTFE_FLOOR == 4 || STTFE_FLOOR == 6
        acc += av * bv;                                               // floor build: operand reads without the MFMAs
#else
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc2, 0, 0, 0);
#endif
    }
    acc += acc2;
    float mv[4], yv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        mv[r] = fmaxf(acc[r], mel_floor);
#if STTFE_FLOOR == 10 || STTFE_FLOOR == 11
        yv[r] = mv[r];                                                // floor build: no logarithm
#else
        yv[r] = fmaf(__log2f(mv[r]), 0.25f * 0.30102999566398120f, y_bias);
#endif
    }
#pragma unroll                                                        // opaque in place: the zero-extension then stays beside
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(o_r[r]));      // the store, which takes scalar base + 32-bit lane offset
#if STTFE_FLOOR == 7 || STTFE_FLOOR == 11
    run_max = fmaxf(run_max, (yv[0] + yv[1]) + (yv[2] + yv[3]));      // floor build: no feature stores
    return;
#endif
    if (kMask && (tbits | mbits) != 0u) {                             // wave-uniform
        const bool tmasked = (tbits >> (lane & 15)) & 1u;
        const unsigned mb = (mbits >> ((lane >> 4) * 4)) & 15u;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float*>(reinterpret_cast<char*>(out_u) + o_r[r]) = (tmasked || ((mb >> r) & 1u)) ? 0.0f : yv[r];
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(reinterpret_cast<char*>(out_u) + o_r[r]) = yv[r];
    }
    // max / min count the masked cells too: the masks are applied to the finished features (src/data.py:185-194)
#if STTFE_FLOOR != 10
    run_max = fmaxf(fmaxf(run_max, fmaxf(mv[0], mv[1])), fmaxf(mv[2], mv[3]));
    tile_min = fminf(fminf(tile_min, fminf(mv[0], mv[1])), fminf(mv[2], mv[3]));
#endif
}

// First frame (a multiple of 4) from which every cell of a clip is the padding constant.  Zero padding to 30 s: frame t
// reads samples [160 t - 200, 160 t + 200), so frames from ceil((n + 200) / 160) on see only zeros (clips within a
// second of 30 s are left to the tile rule: the reflection at the 30 s end reads backwards).  Kernel 1 still computes
// the cells of such frames inside its last tile (they come out as exactly the constant), but keeps them out of the
// tile's minimum -- otherwise the last tile of EVERY short clip is flagged for kernel 2's clamp -- and kernel 2 fills
// from here.  Variable-length mode keeps the tile rule (its padding is padding_value, written by kernel 1 up to the tile end).
__device__ __forceinline__ int whisper_const_frame(int n_valid, int n_frames, int tile_pad_frame, int mode) {
    if (mode == wh::kModeTorchVarlen || n_valid > wh::kMaxSamples - 16000) return tile_pad_frame;
    const int f = ((n_valid + wh::kNfft / 2 + wh::kHop - 1) / wh::kHop + 3) & ~3;
    return min(min(f, tile_pad_frame), n_frames);
}

// HF mode: first 48-frame tile that lies entirely in the zero padding to 30 s (as first_pad_tile_of in the kernels)
__device__ __forceinline__ int whisper_first_pad_tile_hf(int n_valid) {
    return (n_valid + wh::kNfft / 2 + wh::kHop * wh::kTileFrames - 1) / (wh::kHop * wh::kTileFrames);
}
// A batch is uniform when its shortest clip ends within a tile of its longest one.  Uniform batches take long runs
// (a third of the clip) -- and, for the calls whisper_stft_mel12_kernel serves, that kernel instead of this file's
// whisper_stft_mel_kernel: both kernels evaluate this same predicate on the same lengths, so exactly one of them works.
__device__ __forceinline__ bool whisper_batch_is_uniform(int min_len, int max_tiles) {
    return whisper_first_pad_tile_hf(min_len) + 1 >= max_tiles;
}

#ifndef STTFE_FLOOR
#define STTFE_FLOOR 0
#endif
#ifndef STTFE_WH_WAVES_PER_SIMD
#define STTFE_WH_WAVES_PER_SIMD 2
#endif
__global__ __launch_bounds__(wh::kThreads, STTFE_WH_WAVES_PER_SIMD) void whisper_stft_mel_kernel(
    const float* __restrict__ pcm, const int64_t* __restrict__ offsets,
    const int32_t* __restrict__ lengths, int n_mels, int n_frames,
    const WhisperTablesHeader* __restrict__ tab, float* __restrict__ out,
    unsigned* __restrict__ keys, const int32_t* __restrict__ mask_tbl, int n_masks, int tiles_per_block,
    int batch, unsigned* __restrict__ ticket, const float* __restrict__ clip_scale, int mode, float pad_value,
    int skip_uniform) {
    using namespace wh;
    __shared__ __attribute__((aligned(16))) float smem[kLdsFloats];

    // a tables blob built for another extractor (wrong magic / n_mels): nothing is computed, kernel 2 poisons the output
    if (tab->magic != kWhisperMagic || tab->n_mels != n_mels) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if defined(STTFE_PRIO)
    // Experiment: unequal wave priorities for the two workgroups of a CU, so that one of them
    // always wins the VALU and the other's transform falls into its mel phases.  1: by wave slot parity (HW_ID),
    // 2: second half of the grid, 3: odd workgroups
    {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const bool hi = STTFE_PRIO == 1 ? (hw & 1u) : (STTFE_PRIO == 2 ? blockIdx.x >= gridDim.x / 2 : (blockIdx.x & 1u));
        if (hi) __builtin_amdgcn_s_setprio(3);
    }
#endif
#ifdef STTFE_STAMPS
    unsigned long long stamp_entry;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_entry)::"memory");
#endif
    const int p = lane / 20;                 // pair within the wave (3 = idle lanes 60..63)
    const int j = lane - p * 20;
    const bool active = lane < 60;
    const int pa_ = active ? p : 0;

    // Work items are (tile run, clip) pairs, run-major: item = run * batch + clip, for the runs that the longest clip
    // of the batch reaches.  A workgroup stages the tables once and then takes items from a ticket counter in the
    // workspace (zeroed by the host): the first items are blockIdx.x and gridDim.x + blockIdx.x, every further one
    // 2 gridDim.x + atomicInc(ticket).
    // Thread 0 requests the ticket two items ahead and publishes it through LDS at the end of an item, so that the
    // atomic, the length / offset of the next item and its first PCM tile are all in flight under useful work.
    // Every workgroup leaves its loop after at most n_items tickets; all threads of a workgroup take the same path.
    // tiles a clip can have: those of the output -- except in variable-length mode, where a clip longer than the output
    // (more than 3000 frames) is transformed to its own end for the per-clip maximum and only stored up to n_frames
    // (batches of more than 16384 clips, whose lengths are not reduced below, stop at 63 tiles as before)
    const int n_tiles = mode == kModeTorchVarlen ? (kVarMaxSamples / kHop + kTileFrames) / kTileFrames
                                                 : (n_frames + kTileFrames - 1) / kTileFrames;
    const int len_cap = mode == kModeTorchVarlen ? kVarMaxSamples : kMaxSamples;
    auto clamp_len = [len_cap](int n) { return n < 0 ? 0 : (n > len_cap ? len_cap : n); };
    // first tile that lies entirely in the padding (those are constant: kernel 2 fills them): past the last frame that
    // touches a valid sample (zero padding to 30 s) or, in variable-length mode, past the clip's own 1 + n / 160 frames
    auto first_pad_tile_of = [mode](int n_valid) {
        return mode == kModeTorchVarlen ? (1 + n_valid / kHop + kTileFrames - 1) / kTileFrames
                                        : (n_valid + kNfft / 2 + kHop * kTileFrames - 1) / (kHop * kTileFrames);
    };
    // (atomicInc with an all-ones bound = "add 1, return the old value".  With atomicAdd the compiler's atomic optimiser
    // turns the one-lane add into a wave reduction whose result it broadcasts with v_readfirstlane RIGHT AFTER the
    // atomic: s_waitcnt vmcnt(0) at the top of every item, i.e. a full device-atomic round trip -- and every store and
    // prefetch in flight -- before wave 0 could go on.  The optimiser leaves inc alone.  Measured: -2.5 % on the float32
    // Kaldi kernel, whose items are short; neutral here, where the other resident workgroup covers wave 0's wait.)
    // The first TWO items of a workgroup are static (blockIdx.x and gridDim.x + blockIdx.x); tickets start at the third.
    // With a ticket for the second item every workgroup's first act was an atomic on the same word, needed before its
    // first tile: 512 serialised device-scope atomics, i.e. ~10 us before the last workgroup could start (12 us of
    // "start-up" per workgroup in tools/stamps.py).
    unsigned my_ticket = 0;
    // (requested first of all: the table loads are in flight under the reduction of the clip lengths below)
    static_assert(sizeof(WhisperTablesHeader) % 16 == 0, "the mel weights behind the header are read as 16-byte words");
    constexpr int kA = (kMaxAfragFloats / 4 + kThreads - 1) / kThreads;
    f32x4 av[kA];
    float wv[2], tv[4];
    const int na4 = min(tab->afrag_floats, kMaxAfragFloats) >> 2;
    {
        const f32x4* __restrict__ afrag4 = reinterpret_cast<const f32x4*>(tab + 1);
#pragma unroll
        for (int k = 0; k < kA; ++k) av[k] = afrag4[min(tid + k * kThreads, na4 - 1)];
#pragma unroll
        for (int k = 0; k < 2; ++k) wv[k] = tab->window[min(tid + k * kThreads, 399)];
#pragma unroll
        for (int k = 0; k < 4; ++k) tv[k] = tab->twiddle[min(tid + k * kThreads, 799)];
    }
    // longest and shortest clip of the batch (every workgroup computes the same values; skipped for very large batches)
    int max_len = kMaxSamples, min_len = 0, sum_tiles = 0;             // sum_tiles: computed tiles of the whole batch
    // this thread's clips tid * per + k (batches of up to kMaxPlanClips): all lengths requested at once and kept for the
    // item plan below -- as a load-use loop every trip was a round trip of its own
    constexpr int kPer = kMaxPlanClips / kThreads;
    const int per = (batch + kThreads - 1) / kThreads;
    int lens[kPer];
    if (batch <= kMaxPlanClips) {
#pragma unroll
        for (int k = 0; k < kPer; ++k) lens[k] = clamp_len(lengths[min(tid * per + k, batch - 1)]);
    }
    if (batch <= 16384) {
        int m = 0, mn = kMaxSamples, st = 0;
        if (batch <= kMaxPlanClips) {
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                if (k < per && tid * per + k < batch) {
                    m = max(m, lens[k]);
                    mn = min(mn, lens[k]);
                    st += min(n_tiles, first_pad_tile_of(lens[k]));
                }
            }
        } else {
            for (int i = tid; i < batch; i += kThreads) {
                const int l = clamp_len(lengths[i]);
                m = max(m, l);
                mn = min(mn, l);
                st += min(n_tiles, first_pad_tile_of(l));
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m = max(m, __shfl_xor(m, off, 64));
            mn = min(mn, __shfl_xor(mn, off, 64));
            st += __shfl_xor(st, off, 64);
        }
        if (lane == 0) {                                           // the slabs are zeroed after the barrier below
            reinterpret_cast<int*>(smem)[wave] = m;
            reinterpret_cast<int*>(smem)[kWaves + wave] = mn;
            reinterpret_cast<int*>(smem)[2 * kWaves + wave] = st;
        }
        __syncthreads();
        max_len = 0;
        min_len = kMaxSamples;
        for (int w = 0; w < kWaves; ++w) {
            max_len = max(max_len, reinterpret_cast<const int*>(smem)[w]);
            min_len = min(min_len, reinterpret_cast<const int*>(smem)[kWaves + w]);
            sum_tiles += reinterpret_cast<const int*>(smem)[2 * kWaves + w];
        }
        __syncthreads();
    }
    // Run length (tiles per item).  With persistent workgroups the run length does not change the time per tile
    // (7 and 21 tiles both 1.03 ms for 1024 full-length clips once the clocks have settled); short runs keep the
    // workgroups balanced when clip lengths differ (11-tile runs cost +40 % on Common-Voice-length clips): a third
    // of the clip for (near-)uniform batches; otherwise about eight items per workgroup (runs of 3 tiles for batches
    // of 512-1024 Common-Voice-length clips: 0.136 ms against 0.149 ms with a fifth of the longest clip, config 3).
    const int max_tiles = min(n_tiles, first_pad_tile_of(max_len));
    // (HF mode only, set by the host for the calls whisper_stft_mel12_kernel serves: that kernel has computed this batch)
    if (skip_uniform && whisper_batch_is_uniform(min_len, max_tiles)) return;
    if (tiles_per_block <= 0) {
        const bool uniform = first_pad_tile_of(min_len) + 1 >= max_tiles;
        // (a uniform batch too small to give every workgroup a third of a clip -- a serving batch of a few 30 s chunks --
        // takes shorter runs, one item per workgroup: one clip was three items on three CUs, 115 us per call)
        tiles_per_block = uniform ? min((max_tiles + 2) / 3, (batch * max_tiles + (int)gridDim.x - 1) / (int)gridDim.x)
                                  : (sum_tiles > 0 ? sum_tiles / (8 * (int)gridDim.x) : (max_tiles + 4) / 5);
        tiles_per_block = max(3, min(21, tiles_per_block));
    }
    tiles_per_block = __builtin_amdgcn_readfirstlane(tiles_per_block);
    const int n_runs = (max_tiles + tiles_per_block - 1) / tiles_per_block;
    // Item space.  Dense: item = run * batch + clip for the runs the LONGEST clip reaches -- every item exists for
    // uniform batches, but in a ragged batch half of them are empty, and an empty item still costs a ticket round trip
    // and a barrier (11 % + 5 % of the kernel on config 4, tools/stamps.py).  Ragged batches of up to kMaxPlanClips
    // clips are therefore compacted: every workgroup builds the exclusive prefix of the clips' own run counts in LDS
    // (clip-major items, plan[batch] of them) and finds an item's clip by bisection.
    auto runs_of = [&](int n_valid) {
        return (min(n_tiles, first_pad_tile_of(n_valid)) + tiles_per_block - 1) / tiles_per_block;
    };
    int* plan = reinterpret_cast<int*>(smem + kPlanOff);
    const bool planned = __builtin_amdgcn_readfirstlane(
        (batch <= kMaxPlanClips && batch <= 16384 && runs_of(min_len) != n_runs) ? 1 : 0);
    int n_items = n_runs * batch;
    if (planned) {
        int before[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int c = tid * per + k;
            before[k] = sum;
            if (k < per && c < batch) sum += runs_of(lens[k]);
        }
        int incl = sum;                                                // inclusive scan over the wave, then over the waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) reinterpret_cast<int*>(smem)[wave] = incl;
        __syncthreads();
        int base = incl - sum;
        for (int w = 0; w < wave; ++w) base += reinterpret_cast<const int*>(smem)[w];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int c = tid * per + k;
            if (k < per && c < batch) plan[c] = base + before[k];
        }
        if (tid == kThreads - 1) plan[batch] = base + sum;
        __syncthreads();
        n_items = plan[batch];
    }
    // item -> (clip, run)
    auto decode = [&](int it, int& clip_out, int& run_out) {
        if (planned) {
            // largest c with plan[c] <= it (plan is strictly increasing: every clip has a run), two 64-way steps: the
            // lanes sample plan at 64 points / read one block of it, and a ballot counts the entries <= it
            // (a bisection's 12 dependent LDS reads were 1.7 k cycles per item)
            const int stride = (batch + 63) >> 6;
            const int c1 = lane * stride;
            const bool le1 = c1 < batch && plan[min(c1, batch)] <= it;
            const int block = __popcll(__ballot(le1)) - 1;                  // >= 0: plan[0] = 0 <= it
            const int c2 = block * stride + lane;
            const bool le2 = lane < stride && c2 < batch && plan[min(c2, batch)] <= it;
            const int c = block * stride + __popcll(__ballot(le2)) - 1;
            clip_out = __builtin_amdgcn_readfirstlane(c);
            run_out = __builtin_amdgcn_readfirstlane(it - plan[c]);
        } else {
            run_out = __builtin_amdgcn_readfirstlane(it / batch);
            clip_out = __builtin_amdgcn_readfirstlane(it - run_out * batch);
        }
    };
    int item = blockIdx.x, next_item = 0;
    if (item >= n_items) return;
    int clip_cur, run_cur, clip_nx = 0, run_nx = 0;
    decode(item, clip_cur, run_cur);
    // length / offset of the current item and of the next one (requested one item ahead)
    int len_cur = lengths[clip_cur];
    int64_t off_cur = offsets[clip_cur];
    int len_next = 0;
    int64_t off_next = 0;
    unsigned iter = 0;

    // 28 samples per lane and pass: frame a = rows 0..19, frame b = rows 8..27 of x[base + 20 m]
    auto load_pass = [&](const float* __restrict__ x, int n_valid, int t0, int pass, float (&r)[28]) {
#if STTFE_FLOOR == 1
        // floor build, no HBM: the registers keep their values, nudged so that no transform can be hoisted out of the loop
#pragma unroll
        for (int m = 0; m < 28; ++m) r[m] += 1.0e-3f;
        return;
#endif
        const int f0 = t0 + wave * kFramesPerWave + 6 * pass;          // first frame of the pass (wave-uniform)
        const int first = kHop * f0 - kNfft / 2;
        const int base = first + 2 * kHop * pa_ + j;
        if (first >= 0 && first + 5 * kHop + kNfft <= n_valid) {
#pragma unroll
            for (int m = 0; m < 28; ++m) r[m] = x[base + 20 * m];
        } else if (mode != kModeTorchVarlen && first >= n_valid && first + 5 * kHop + kNfft <= kMaxSamples) {
            // the whole pass lies in the zero padding to 30 s (about half of the passes of a short clip's last tile)
#pragma unroll
            for (int m = 0; m < 28; ++m) r[m] = 0.0f;
        } else {
#pragma unroll
            for (int m = 0; m < 28; ++m) {
                int idx = base + 20 * m;
                const int lim = mode == kModeTorchVarlen ? n_valid : kMaxSamples;
                idx = idx < 0 ? -idx : idx;                                       // reflect padding (center=True) ...
                idx = idx >= lim ? 2 * (lim - 1) - idx : idx;                     // ... of the 30 s buffer / of the clip
                r[m] = (idx >= 0 && idx < n_valid) ? x[idx] : 0.0f;               // zero padding to 30 s
            }
        }
    };
    float raw0[28], raw1[28];                      // the two passes of the NEXT tile, loaded a whole tile ahead
#if STTFE_FLOOR
    // Floor builds (tools/ubench/whisper_floor.py): the same kernel with phases left out; all but
    // the shipped build (0) give wrong features and exist for timing only.
    //   1: the transform alone -- no PCM loads, no power panel, no barriers, no mel projection, no stores
    //   2: PCM loads + transform (no panel, barriers, mel, stores)
    //   3: everything but the transform (the power values are copies of the samples)
    //   4: the whole kernel with the mel projection's MFMAs replaced by one packed multiply-add per operand pair
    //   5: the whole kernel without the two workgroup barriers of the tile loop;  6: both 4 and 5
    //   7: the whole kernel without the feature stores;  8: without the mel phase (operand reads, MFMAs, epilogue, stores)
    //   9: the mel units' operands from registers instead of LDS;  10: no logarithm, no min / max keys;  11: 7 + 9 + no logarithm
    f2 floor_sink = {0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < 28; ++m) {
        raw0[m] = 1.0e-2f * (float)(lane + m);
        raw1[m] = 2.0e-2f * (float)(lane - m);
    }
#endif
    bool raw_ready = false;                        // raw0 / raw1 already hold the first tile of the current item

    // zero the slabs once (pad cells are read by zero-weight MFMA steps and must be finite), stage tables.
    // Every global load of the staging is requested before the first LDS store: as copy loops (load, store, load, ...)
    // the 18 + 4 + 2 dependent round trips per thread were ~15 us of every launch (a quarter of a 512-clip config-3 launch).
    {
        for (int i = tid; i < kWinOff; i += kThreads) smem[i] = 0.f;
        // window as [j][n1] and twiddles W400^(j k1) as [j][k1]: each lane reads its 20 values as b128 words
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = tid + k * kThreads;
            if (i < 400) smem[kWinOff + (i % 20) * 20 + i / 20] = wv[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + k * kThreads;
            const int c = i >> 1, k1 = c / 20, jj = c - 20 * k1;
            if (i < 800) smem[kTwOff + jj * kTwStride + 2 * k1 + (i & 1)] = tv[k];
        }
#pragma unroll
        for (int k = 0; k < kA; ++k) {
            const int i = tid + k * kThreads;
            if (i < na4) reinterpret_cast<f32x4*>(smem + kAfragOff)[i] = av[k];
        }
        int* desc = reinterpret_cast<int*>(smem + kDescOff);
        if (tid < kWaves * kMaxUnitsPerWave)
            reinterpret_cast<int4v*>(desc)[tid] = reinterpret_cast<const int4v*>(&tab->plan.desc[0][0][0])[tid];
        if (tid < kWaves) desc[4 * kWaves * kMaxUnitsPerWave + tid] = tab->plan.wave_nunits[tid];
        if (tid == 0) reinterpret_cast<unsigned*>(smem + kTicketOff)[0] = blockIdx.x + gridDim.x;
    }
    __syncthreads();
    next_item = (int)min(reinterpret_cast<const unsigned*>(smem + kTicketOff)[0], (unsigned)n_items);
    if (next_item < n_items) decode(next_item, clip_nx, run_nx);

    const float* afrag = smem + kAfragOff;
    const int4v* desc = reinterpret_cast<const int4v*>(smem + kDescOff) + wave * kMaxUnitsPerWave;
    const int nu = __builtin_amdgcn_readfirstlane(
        reinterpret_cast<const int*>(smem + kDescOff)[4 * kWaves * kMaxUnitsPerWave + wave]);
    const int partner = active ? (p * 20 + ((20 - j) % 20)) : lane;
    float* my_rows = smem + wave * kSlabFloats;
    f2* slab_w = reinterpret_cast<f2*>(my_rows) + pa_ * kPairStride + j;                        // stage-1 writes: [k1][n2 = j]
    const f32x4* slab_r = reinterpret_cast<const f32x4*>(my_rows + 2 * (pa_ * kPairStride + j * kRowStride));   // row k1 = j
    const f32x4* win4 = reinterpret_cast<const f32x4*>(smem + kWinOff + 20 * j);
    const f32x4* tw4 = reinterpret_cast<const f32x4*>(smem + kTwOff + kTwStride * j);


    // The twiddles of this lane stay in registers for the whole kernel (40 VGPRs).
    f32x4 t4[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) t4[q] = tw4[q];

    // The FFT of one pass (3 frame pairs) in four pieces, so that the tile loop can interleave the two passes and
    // put arithmetic of one pass under the LDS latencies of the other.  Complex values live in even-aligned
    // register pairs and go through the packed-f32 forms of fft_pk.h.
    //   s1_math  : window, 20-point DFTs over n1, twiddles
    //   s1_write : [k1][n2] rows into the wave's exchange slab
    //   s2_read  : row k1 = j back as 10 x b128 (issued early, consumed after other work)
    //   s2_math  : 20-point DFTs over n2: y[k2] = Z[j + 20 k2]
    //   sep_*    : partner lane's Z[400 - k] by ds_bpermute, then 4 |A|^2, 4 |B|^2 for bins j + 20 k2
    auto s1_math = [&](const float (&raw)[28], f2 (&y)[20]) {
        f2 v[20];
        f32x4 w4[5];                                               // window of this lane: 5 b128 reads per pass
#pragma unroll
        for (int q = 0; q < 5; ++q) w4[q] = win4[q];
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            const float w = w4[i >> 2][i & 3];
            v[i] = f2{raw[i] * w, raw[i + 8] * w};
        }
        dft20_pk(v, y);
#pragma unroll
        for (int k1 = 1; k1 < 20; ++k1) {
            const f2 tw = {t4[k1 >> 1][2 * (k1 & 1)], t4[k1 >> 1][2 * (k1 & 1) + 1]};
            y[k1] = cmul_pk(y[k1], tw);
        }
    };
    auto s1_write = [&](const f2 (&y)[20]) {
        wave_lds_sync();      // after every earlier read of the slab / panel rows by this wave
        if (active) {
#pragma unroll
            for (int k1 = 0; k1 < 20; ++k1) slab_w[k1 * kRowStride] = y[k1];
        }
        wave_lds_sync();
    };
    auto s2_read = [&](f32x4 (&r4)[10]) {
        // the idle lanes 60..63 read rows 0..3 of pair 0 too (same addresses as lanes 0..3: a broadcast, no bank
        // conflict); predicating them off cost 40 v_mov per pass to give their registers a defined value
#pragma unroll
        for (int q = 0; q < 10; ++q) r4[q] = slab_r[q];
    };
    auto s2_math = [&](const f32x4 (&r4)[10], f2 (&y)[20]) {
        f2 v[20];
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            v[2 * q] = f2{r4[q][0], r4[q][1]};
            v[2 * q + 1] = f2{r4[q][2], r4[q][3]};
        }
        dft20_pk(v, y);
    };
    auto sep_fetch = [&](const f2 (&y)[20], f2 (&q)[10]) {
#pragma unroll
        for (int k2 = 0; k2 < 10; ++k2)
            q[k2] = f2{__shfl(y[19 - k2].x, partner, 64), __shfl(y[19 - k2].y, partner, 64)};
    };
    auto sep_math = [&](const f2 (&y)[20], const f2 (&q)[10], f2 (&pw)[11]) {
#pragma unroll
        for (int k2 = 0; k2 < 10; ++k2) {
            const f2 qq = (j == 0) ? ((k2 == 0) ? y[0] : y[20 - k2]) : q[k2];      // lane 0: own Z[400 - 20 k2]
            pw[k2] = power_pair(y[k2] + qq, y[k2] - qq);          // (4 |A[k]|^2, 4 |B[k]|^2)
        }
        pw[10] = y[10] * y[10] * 4.0f;                            // bin 200 (lane j == 0 only)
    };

    auto store_rows = [&](int pass, const f2 (&pw)[11]) {
        if (!active) return;
        float* row = my_rows + (6 * pass + 2 * p) * kPStride + j;
#pragma unroll
        for (int k2 = 0; k2 < 10; ++k2) {
            row[20 * k2] = pw[k2].x;
            row[kPStride + 20 * k2] = pw[k2].y;
        }
        if (j == 0) {
            row[200] = pw[10].x;
            row[kPStride + 200] = pw[10].y;
        }
    };

    // LDS float offset of this lane's B-operand row (frame ft * 16 + (lane & 15), k = lane >> 4) per frame tile
    auto b_row = [&](int ft) {
        const int fl = ft * 16 + (lane & 15), fw = fl / kFramesPerWave;
        return fw * kSlabFloats + (fl - fw * kFramesPerWave) * kPStride + 4 * (lane >> 4);
    };
    const int brow0 = b_row(0), brow1 = b_row(1), brow2 = b_row(2);
    // static plan (mel_unit_static): usable when the blob's plan is one of those compiled in
    const int static_plan = __builtin_amdgcn_readfirstlane(
        n_mels == 80 && plan_matches<80>(tab->plan) ? 80 : (n_mels == 128 && plan_matches<128>(tab->plan) ? 128 : 0));
    const float* a_lane = smem + kAfragOff + 4 * lane;
    unsigned o_r[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o_r[r] = 4u * (unsigned)(((lane >> 4) * 4 + r) * n_frames + (lane & 15));   // BYTES: a 32-bit
    // lane offset beside the unit's scalar base is the store's own addressing mode (no 64-bit add per store)
#ifdef STTFE_STAMPS
    // slots 0-9: phases of the tile loop; 10: kernel entry -> item loop; 11: item switches (end of an item's last tile
    // -> start of the next item's first tile, empty items included); 12: after the last item; 13 / 14: entry / exit time;
    // 16: item tail + ticket hand-over barrier; 17: item decode
    unsigned long long stamp_sum[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = stamp_entry;
    STAMP(10);
#endif

    while (item < n_items) {
      const int run = run_cur, clip = clip_cur;
      // (lengths / offsets are workgroup-uniform: move them to scalar registers once they have arrived)
      const int n_valid = clamp_len(__builtin_amdgcn_readfirstlane(len_cur));
      const int64_t off_x = ((int64_t)__builtin_amdgcn_readfirstlane((int)(off_cur >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)off_cur);
      const float* __restrict__ x = pcm + off_x;
      // Per-clip amplitude scale s (peak normalisation folded in, sttfe_whisper_logmel_scaled): the features of s x are
      // log10(max(s^2 mel, 1e-10)) = log10(max(mel, 1e-10 / s^2)) + 2 log10(s) -- a floor and a bias of the epilogue,
      // nothing per sample.  Maxima / minima (keys) stay in the unscaled domain; kernel 2 applies the same bias.
      float mel_floor = 1e-10f, y_bias = 1.0f;
      if (clip_scale) {
          const float sc = clip_scale[clip];                       // workgroup-uniform address: a scalar load
          mel_floor = 1e-10f / (sc * sc);
          y_bias = 1.0f + 0.5f * 0.30102999566398120f * __log2f(sc);
      }
      if (tid == 0 && next_item < n_items) my_ticket = atomicInc(ticket, 0xffffffffu);     // -> the item after the next
      if (next_item < n_items) {                   // lands while this item is processed
          len_next = lengths[clip_nx];
          off_next = offsets[clip_nx];
      }
      const int tile_begin = run * tiles_per_block;
      const int tile_end = min(min(tile_begin + tiles_per_block, n_tiles), first_pad_tile_of(n_valid));
      // frames that count for the per-clip maximum (and hold data rather than padding_value)
      const int key_frames = mode == kModeTorchVarlen ? 1 + n_valid / kHop : (mode == kModeTorchPadded ? n_frames + 1 : n_frames);
      // frames from here on are the padding constant (kernel 2 fills them): not part of a tile's minimum
      const int const_frame = whisper_const_frame(n_valid, n_frames, min(n_frames, min(n_tiles, first_pad_tile_of(n_valid)) * kTileFrames), mode);
      if (tile_begin < tile_end) {
      float* __restrict__ out_clip = out + (size_t)clip * n_mels * n_frames;
      // SpecAugment: mel-axis masks as two 64-bit words for the clip; time-axis masks per tile
      // The clip's mask table goes to LDS once per item (read per tile from global memory it cost one HBM round
      // trip per tile); it is first read after the "panel complete" barrier of the item's first tile.  (Wave 0 waits
      // for this load at the top of every item; requesting it an item ahead was measured neutral.)
      const int nm = mask_tbl ? n_masks : 0;
      int32_t* clip_masks = reinterpret_cast<int32_t*>(smem + kMaskOff);
      if (tid < 3 * nm) clip_masks[tid] = mask_tbl[(size_t)clip * n_masks * 3 + tid];
      uint64_t melmask_lo = 0, melmask_hi = 0;
      if (!raw_ready) {
          load_pass(x, n_valid, tile_begin * kTileFrames, 0, raw0);
          load_pass(x, n_valid, tile_begin * kTileFrames, 1, raw1);
      }
      raw_ready = false;
      float run_max = 0.0f;                         // mel energies (>= 1e-10 once any cell was produced)
      for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int t0 = tile * kTileFrames;
#ifdef STTFE_STAMPS
        if (tile == tile_begin) STAMP(11);
#endif
        {
            f2 ya[20], yb[20], pw0[11], pw1[11], q0[10], q1[10];
            f32x4 ra[10], rb[10];
#if STTFE_FLOOR == 3
#pragma unroll
            for (int k2 = 0; k2 < 11; ++k2) {
                pw0[k2] = f2{raw0[k2], raw0[k2 + 8]};
                pw1[k2] = f2{raw1[k2], raw1[k2 + 8]};
            }
            wave_lds_sync();
            store_rows(0, pw0);
            store_rows(1, pw1);
#else
            s1_math(raw0, ya);
            s1_write(ya);
            s2_read(ra);                       // pass 0 rows in flight ...
            s1_math(raw1, yb);                 // ... under the first-stage arithmetic of pass 1
            STAMP(0);
            s1_write(yb);                      // LDS runs a wave's operations in order: after the reads above
            s2_read(rb);
            s2_math(ra, ya);                   // under the pass-1 reads
            sep_fetch(ya, q0);
            STAMP(2);
            s2_math(rb, yb);                   // under the pass-0 partner exchange
            sep_fetch(yb, q1);
            sep_math(ya, q0, pw0);
            STAMP(3);
            sep_math(yb, q1, pw1);
#if STTFE_FLOOR == 1 || STTFE_FLOOR == 2
#pragma unroll
            for (int k2 = 0; k2 < 11; ++k2) floor_sink += pw0[k2] + pw1[k2];
#else
            wave_lds_sync();                   // panel rows alias the slab: after its last read
            STAMP(4);
            store_rows(0, pw0);
            store_rows(1, pw1);
#endif
#endif
        }
        // next tile's PCM -- of this item, or the first tile of the workgroup's next item: issued now (few live
        // registers from here on), lands during the mel projection
        if (tile + 1 < tile_end) {
            load_pass(x, n_valid, t0 + kTileFrames, 0, raw0);
            load_pass(x, n_valid, t0 + kTileFrames, 1, raw1);
        } else if (next_item < n_items) {
            // (the empty asm makes these values opaque here: otherwise the compiler computes the 56 load indices and
            // bounds predicates of the next item at the top of this item and spills them across the whole tile loop)
            int nv = clamp_len(__builtin_amdgcn_readfirstlane(len_next)), tb = __builtin_amdgcn_readfirstlane(run_nx * tiles_per_block);
            asm volatile("" : "+v"(nv), "+v"(tb));
            nv = __builtin_amdgcn_readfirstlane(nv);
            tb = __builtin_amdgcn_readfirstlane(tb);
            const int64_t off_n = ((int64_t)__builtin_amdgcn_readfirstlane((int)(off_next >> 32)) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)off_next);
            if (tb < min(n_tiles, first_pad_tile_of(nv))) {
                load_pass(pcm + off_n, nv, tb * kTileFrames, 0, raw0);
                load_pass(pcm + off_n, nv, tb * kTileFrames, 1, raw1);
                raw_ready = true;
            }
        }
#if STTFE_FLOOR == 1 || STTFE_FLOOR == 2
        continue;
#endif
        STAMP(5);
#if STTFE_FLOOR == 5 || STTFE_FLOOR == 6
        wave_lds_sync();                                           // floor build: no workgroup barriers in the tile loop
#else
        __syncthreads();                                           // panel complete
#endif
        STAMP(6);

        // ---------------- banded mel projection on the matrix cores + log epilogue --------------
        const uint64_t tmask = mask_bits(clip_masks, nm, STTFE_AXIS_TIME, t0, kTileFrames);
        if (tile == tile_begin) {
            melmask_lo = mask_bits(clip_masks, nm, STTFE_AXIS_FREQ, 0, 64);
            melmask_hi = mask_bits(clip_masks, nm, STTFE_AXIS_FREQ, 64, 64);
        }
        // Operands come as one b128 per lane and group of four k-steps: step i of group g multiplies bins
        // kstart + 16 g + 4 (lane >> 4) + i (tables.cpp lays the mel weights out to match).  Two interleaved
        // accumulation chains hide the 40-cycle dependent latency of the 16x16x4 MFMA; the next group's
        // operands -- across unit boundaries too -- are read while the current four steps multiply.
        // (Running a unit's epilogue under the first MFMAs of the next unit was measured slower: 1.15 vs 1.08 ms.)
        float tile_min = 3.0e38f;
#if STTFE_FLOOR == 8
        if (false) {
#else
        if (static_plan != 0 && t0 + kTileFrames <= min(min(n_frames, key_frames), const_frame)) {
#endif
            // whole tile of an 80- or 128-bin extractor: the wave's units from the static plan (the same assignment
            // as tables.cpp's: 48 / 44 / 44 / 44 k-steps per wave for 80 bins, 52 / 52 / 52 / 48 for 128)
            float* __restrict__ ob = out_clip + t0;
            const float* b0 = smem + brow0;
            const float* b1 = smem + brow1;
            const float* b2 = smem + brow2;
            const size_t nf16 = (size_t)16 * n_frames;
#define STTFE_U(NM, MT, FT) \
    mel_unit_static<NM, MT, STTFE_MASKED>(a_lane, b##FT, ob + MT * nf16 + FT * 16, o_r, mel_floor, y_bias, run_max, tile_min, \
                                          (unsigned)(tmask >> (FT * 16)) & 0xffffu,                                     \
                                          (unsigned)((MT < 4 ? melmask_lo : melmask_hi) >> ((MT & 3) * 16)) & 0xffffu, lane)
            if (nm == 0) {
#define STTFE_MASKED false
            if (static_plan == 80) {
                if (wave == 0) {
                    STTFE_U(80, 4, 0); STTFE_U(80, 3, 2); STTFE_U(80, 2, 2);
                } else if (wave == 1) {
                    STTFE_U(80, 4, 1); STTFE_U(80, 1, 0); STTFE_U(80, 1, 1); STTFE_U(80, 0, 0);
                } else if (wave == 2) {
                    STTFE_U(80, 4, 2); STTFE_U(80, 2, 0); STTFE_U(80, 2, 1); STTFE_U(80, 0, 1);
                } else {
                    STTFE_U(80, 3, 0); STTFE_U(80, 3, 1); STTFE_U(80, 1, 2); STTFE_U(80, 0, 2);
                }
            } else {
                if (wave == 0) {
                    STTFE_U(128, 7, 0); STTFE_U(128, 5, 1); STTFE_U(128, 4, 0);
                    STTFE_U(128, 1, 0); STTFE_U(128, 1, 1); STTFE_U(128, 1, 2);
                } else if (wave == 1) {
                    STTFE_U(128, 7, 1); STTFE_U(128, 6, 1); STTFE_U(128, 4, 1);
                    STTFE_U(128, 2, 0); STTFE_U(128, 2, 1); STTFE_U(128, 2, 2);
                } else if (wave == 2) {
                    STTFE_U(128, 7, 2); STTFE_U(128, 5, 2); STTFE_U(128, 4, 2);
                    STTFE_U(128, 3, 0); STTFE_U(128, 3, 1); STTFE_U(128, 3, 2);
                } else {
                    STTFE_U(128, 5, 0); STTFE_U(128, 6, 0); STTFE_U(128, 6, 2);
                    STTFE_U(128, 0, 0); STTFE_U(128, 0, 1); STTFE_U(128, 0, 2);
                }
            }
#undef STTFE_MASKED
            } else {
#define STTFE_MASKED true
            if (static_plan == 80) {
                if (wave == 0) {
                    STTFE_U(80, 4, 0); STTFE_U(80, 3, 2); STTFE_U(80, 2, 2);
                } else if (wave == 1) {
                    STTFE_U(80, 4, 1); STTFE_U(80, 1, 0); STTFE_U(80, 1, 1); STTFE_U(80, 0, 0);
                } else if (wave == 2) {
                    STTFE_U(80, 4, 2); STTFE_U(80, 2, 0); STTFE_U(80, 2, 1); STTFE_U(80, 0, 1);
                } else {
                    STTFE_U(80, 3, 0); STTFE_U(80, 3, 1); STTFE_U(80, 1, 2); STTFE_U(80, 0, 2);
                }
            } else {
                if (wave == 0) {
                    STTFE_U(128, 7, 0); STTFE_U(128, 5, 1); STTFE_U(128, 4, 0);
                    STTFE_U(128, 1, 0); STTFE_U(128, 1, 1); STTFE_U(128, 1, 2);
                } else if (wave == 1) {
                    STTFE_U(128, 7, 1); STTFE_U(128, 6, 1); STTFE_U(128, 4, 1);
                    STTFE_U(128, 2, 0); STTFE_U(128, 2, 1); STTFE_U(128, 2, 2);
                } else if (wave == 2) {
                    STTFE_U(128, 7, 2); STTFE_U(128, 5, 2); STTFE_U(128, 4, 2);
                    STTFE_U(128, 3, 0); STTFE_U(128, 3, 1); STTFE_U(128, 3, 2);
                } else {
                    STTFE_U(128, 5, 0); STTFE_U(128, 6, 0); STTFE_U(128, 6, 2);
                    STTFE_U(128, 0, 0); STTFE_U(128, 0, 1); STTFE_U(128, 0, 2);
                }
            }
#undef STTFE_MASKED
            }
#undef STTFE_U
        } else {
#if STTFE_FLOOR == 8
        const int nu = 0;                                           // floor build: no mel projection at all
#endif
        int4v d = desc[0];                                          // {kstart, nsteps, aoff, mt | ft << 8}
        int ks = __builtin_amdgcn_readfirstlane(d.x), ns = __builtin_amdgcn_readfirstlane(d.y);
        int ao = __builtin_amdgcn_readfirstlane(d.z), code = __builtin_amdgcn_readfirstlane(d.w);
        d = desc[1];
        const f32x4* a_ptr = reinterpret_cast<const f32x4*>(afrag + ao) + lane;
        const f32x4* b_ptr = reinterpret_cast<const f32x4*>(smem + ((code >> 8) == 0 ? brow0 : ((code >> 8) == 1 ? brow1 : brow2)) + ks);
        f32x4 av = a_ptr[0], bv = b_ptr[0];
        for (int u = 0; u < nu; ++u) {
            const int mt = code & 0xff, ft = code >> 8, ng = ns >> 2;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
            for (int g = 1; g < ng; ++g) {
                const f32x4 an = a_ptr[g * 64], bn = b_ptr[4 * g];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc2, 0, 0, 0);
                av = an;
                bv = bn;
            }
            f32x4 an = av, bn = bv;
            if (u + 1 < nu) {                                       // first group of the next unit
                ks = __builtin_amdgcn_readfirstlane(d.x);
                ns = __builtin_amdgcn_readfirstlane(d.y);
                ao = __builtin_amdgcn_readfirstlane(d.z);
                code = __builtin_amdgcn_readfirstlane(d.w);
                d = desc[min(u + 2, kMaxUnitsPerWave - 1)];
                a_ptr = reinterpret_cast<const f32x4*>(afrag + ao) + lane;
                b_ptr = reinterpret_cast<const f32x4*>(smem + ((code >> 8) == 0 ? brow0 : ((code >> 8) == 1 ? brow1 : brow2)) + ks);
                an = a_ptr[0];
                bn = b_ptr[0];
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], acc2, 0, 0, 0);
            av = an;
            bv = bn;
            acc += acc2;
            STAMP(9);
            // (Swapping the operand roles so that a lane holds 4 consecutive frames and stores 16 bytes was
            // measured 2 % slower than these four dword stores.)
            const int fl = ft * 16 + (lane & 15);                  // frame within the tile
            const int mel0 = mt * 16 + (lane >> 4) * 4;
            const unsigned o0 = (unsigned)(mel0 * n_frames + t0 + fl);
            // max / min are tracked on the mel energies themselves (log is monotonic); y = (log10(m) + 4) / 4
            float mv[4], yv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mv[r] = fmaxf(acc[r], mel_floor);
                yv[r] = fmaf(__log2f(mv[r]), 0.25f * 0.30102999566398120f, y_bias);
            }
            const bool whole = (mt * 16 + 16 <= n_mels) && (t0 + ft * 16 + 16 <= min(min(n_frames, key_frames), const_frame));    // wave-uniform
            if (whole && nm == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    run_max = fmaxf(run_max, mv[r]);
                    tile_min = fminf(tile_min, mv[r]);
                    out_clip[o0 + (unsigned)(r * n_frames)] = yv[r];
                }
            } else {
                const bool tmasked = (tmask >> fl) & 1ull;
                const unsigned mbits = (unsigned)((mel0 < 64 ? melmask_lo >> mel0 : melmask_hi >> (mel0 - 64)) & 15ull);
                const bool counted = t0 + fl < key_frames, stored = t0 + fl < n_frames, varies = t0 + fl < const_frame;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (mel0 + r < n_mels) {
                        if (counted) {
                            run_max = fmaxf(run_max, mv[r]);
                            if (varies) tile_min = fminf(tile_min, mv[r]);
                        }
                        if (stored)
                            out_clip[o0 + (unsigned)(r * n_frames)] =
                                (tmasked || ((mbits >> r) & 1u)) ? 0.0f : (counted ? yv[r] : pad_value);
                    }
                }
            }
            STAMP(1);
        }
        }   // generic unit loop
        // per-tile minimum (kernel 2 re-reads only the tiles that hold something below max - 8)
        tile_min = wave_min(tile_min);
        if (lane == 0 && tile_min < 1.0e38f && tile < kMaxTiles)       // (tiles behind the output have no min key: not stored)
            atomicMax(&keys[(size_t)clip * kKeysPerClip + 1 + tile], ~float_to_key(__log2f(tile_min) * 0.30102999566398120f));
        STAMP(7);
#if STTFE_FLOOR != 5 && STTFE_FLOOR != 6
        __syncthreads();      // panel reads done before the next tile's slab writes
#endif
        STAMP(8);
      }   // tiles of the item
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) run_max = fmaxf(run_max, __shfl_xor(run_max, off, 64));
      if (lane == 0 && run_max > 0.0f)
          atomicMax(&keys[(size_t)clip * kKeysPerClip], float_to_key(__log2f(run_max) * 0.30102999566398120f));
      }   // item has tiles
      // publish the ticket taken at the top of this item (two alternating LDS slots: a slot is rewritten only after
      // the barrier of the following item, which every thread reaches after this read)
      unsigned* slot = reinterpret_cast<unsigned*>(smem + kTicketOff) + 1 + (iter & 1u);
      if (tid == 0) *slot = (next_item < n_items) ? my_ticket + 2 * gridDim.x : (unsigned)n_items;
      __syncthreads();
      const int after_next = (int)min(*slot, (unsigned)n_items);
#ifdef STTFE_STAMPS
      STAMP(16);      // diagnostic: end of the item's last tile -> ticket hand-over barrier passed
#endif
      ++iter;
      item = next_item;
      clip_cur = clip_nx;
      run_cur = run_nx;
      next_item = after_next;
      if (next_item < n_items) decode(next_item, clip_nx, run_nx);
#ifdef STTFE_STAMPS
      STAMP(17);      // diagnostic: item decode
#endif
      len_cur = len_next;
      off_cur = off_next;
    }
#if STTFE_FLOOR
    if (floor_sink.x == 1234.5f) out[tid] = floor_sink.y;         // keeps the floor builds' arithmetic alive
#endif
#ifdef STTFE_STAMPS
    if (g_stamp_buf && lane == 0) {
        STAMP(12);
        stamp_sum[13] = stamp_entry;
        stamp_sum[14] = stamp_prev;
        unsigned long long* dst = g_stamp_buf + ((size_t)blockIdx.x * kWaves + wave) * 20;
        for (int i = 0; i < 20; ++i) dst[i] = stamp_sum[i];
    }
#endif
}

// Clamp / fill pass.  grid.x = row split, grid.y = clips: workgroup x owns mel rows x, x + gridDim.x, ... of its clip.
//   fill : the clip's zero-padding region [first padding frame, n_frames) gets the final constant with full-row
//          vector stores (kernel 1 never touches it);
//   clamp: of the computed tiles only those whose recorded minimum is below max - 8 (a few per cent of the tiles of
//          ordinary audio; none for most clips) are re-read and clamped to (max - 8 + 4) / 4.
// SpecAugment masks -> 0.0 on everything written.
__global__ __launch_bounds__(256) void whisper_clamp_fill_kernel(
    float* __restrict__ out, const int32_t* __restrict__ lengths, const unsigned* __restrict__ keys,
    int n_mels, int n_frames, const int32_t* __restrict__ mask_tbl, int n_masks,
    const float* __restrict__ clip_scale, int mode, float pad_value, const WhisperTablesHeader* __restrict__ tab) {
    using namespace wh;
    const int clip = blockIdx.y;
    const int row0 = blockIdx.x, row_step = gridDim.x;
    if (tab->magic != kWhisperMagic || tab->n_mels != n_mels) {
        // STTFE_E_TABLES cannot be returned without synchronising: the whole output becomes NaN instead (loud, and
        // never uninitialised memory)
        for (int mel = row0; mel < n_mels; mel += row_step)
            for (int t = threadIdx.x; t < n_frames; t += blockDim.x)
                out[((size_t)clip * n_mels + mel) * n_frames + t] = __builtin_nanf("");
        return;
    }
    // Everything the workgroup needs from memory is requested in ONE batch (length, max key, the per-tile min keys,
    // the clip's mask table -> LDS): with four dependent round trips per workgroup this kernel was latency-bound
    // (0.10 ms for 32 k workgroups that find nothing to clamp).
    __shared__ unsigned long long s_flags;
    __shared__ int32_t s_masks[3 * STTFE_MAX_MASKS];
    const int nm = mask_tbl ? n_masks : 0;
    int n_valid = lengths[clip];
    const unsigned kmax = keys[(size_t)clip * kKeysPerClip];
    unsigned kmin_raw = 0xffffffffu;
    if (threadIdx.x < kKeysPerClip - 1) kmin_raw = keys[(size_t)clip * kKeysPerClip + 1 + threadIdx.x];
    if ((int)threadIdx.x < 3 * nm) s_masks[threadIdx.x] = mask_tbl[(size_t)clip * n_masks * 3 + threadIdx.x];
    const int len_cap = mode == kModeTorchVarlen ? kVarMaxSamples : kMaxSamples;
    n_valid = n_valid < 0 ? 0 : (n_valid > len_cap ? len_cap : n_valid);
    const int n_tiles = (n_frames + kTileFrames - 1) / kTileFrames;
    const int first_pad_tile = min(n_tiles, mode == kModeTorchVarlen
                                                ? (1 + n_valid / kHop + kTileFrames - 1) / kTileFrames
                                                : (n_valid + kNfft / 2 + kHop * kTileFrames - 1) / (kHop * kTileFrames));
    // first frame of the constant region (a multiple of 4)
    const int pad_frame = whisper_const_frame(n_valid, n_frames, min(n_frames, first_pad_tile * kTileFrames), mode);
    const int key_frames = mode == kModeTorchVarlen ? 1 + n_valid / kHop : n_frames;   // frames that hold data
    // scaled clips (see kernel 1): keys are log10 of the unscaled energies, stored values carry the bias c / 4 with
    // c = 2 log10(s); the zero padding is log10(1e-10) = -10 in the scaled domain, i.e. -10 - c in the keys' domain
    const float c_lg = clip_scale ? 2.0f * 0.30102999566398120f * __log2f(clip_scale[clip]) : 0.0f;
    const float lmax = kmax ? key_to_float(kmax) : -10.0f - c_lg; // no computed tile at all: the floor itself
    const float floor_lg = lmax - 8.0f;
    const float floor_y = (floor_lg + 4.0f) * 0.25f + 0.25f * c_lg;
    // zero padding to 30 s: log10(1e-10) clamped like everything else; variable-length mode: the literal padding_value
    const float pad_y = mode == kModeTorchVarlen ? pad_value : fmaxf((-10.0f + 4.0f) * 0.25f, floor_y);

    // tiles of the clip with something below the floor (n_tiles <= 63: one 64-bit word)
    if (threadIdx.x < 64) {
        const bool flag = (int)threadIdx.x < first_pad_tile && key_to_float(~kmin_raw) < floor_lg;
        const unsigned long long bal = __ballot(flag);
        if (threadIdx.x == 0) s_flags = bal;
    }
    __syncthreads();
    const unsigned long long flags = s_flags;
    if (pad_frame >= n_frames && flags == 0ull) return;           // the usual full-length clip: nothing to do

    const int32_t* cm = s_masks;
    const uint64_t melmask_lo = mask_bits(cm, nm, STTFE_AXIS_FREQ, 0, 64);
    const uint64_t melmask_hi = mask_bits(cm, nm, STTFE_AXIS_FREQ, 64, 64);
    int ts[STTFE_MAX_MASKS], te[STTFE_MAX_MASKS], nt = 0;          // time-axis masks of the clip
    for (int i = 0; i < nm; ++i)
        if (cm[3 * i] == STTFE_AXIS_TIME && cm[3 * i + 2] > cm[3 * i + 1]) {
            ts[nt] = cm[3 * i + 1];
            te[nt] = cm[3 * i + 2];
            ++nt;
        }
    float* __restrict__ o = out + (size_t)clip * n_mels * n_frames;
    // the workgroup's rows: one block of consecutive mel rows (so that the line a row's padding shares with the start
    // of the NEXT row belongs to the same workgroup, see the fill)
    const int rows_per = (n_mels + row_step - 1) / row_step;
    const int mel_lo = row0 * rows_per;
    const int rows = max(0, min(rows_per, n_mels - mel_lo));

    // ---- fill.  A row is n_frames floats = 12 000 B long, not a multiple of the 128-byte line, so the padding segment
    // of every row begins and ends inside a line, and a partly written line costs the memory side a read-modify-write:
    // with plain segment stores those two lines per row held the whole pass at 4.3 TB/s (tools/ubench/fill_rate.hip:
    // 5.5 TB/s when they are simply left out).  Lanes are therefore mapped to ABSOLUTE 16-byte slots -- eight lanes per
    // line -- and the lanes of a first / last line that lie outside the segment read what is there (the row's own last
    // computed frames; the first frames of the workgroup's next row) and write it back, so that the line goes out
    // whole: 4.9 TB/s.  Only the last row of a workgroup keeps a partial last line (it belongs to another workgroup).
    if (pad_frame < n_frames && rows > 0) {
        const int vq = (((n_frames - pad_frame) * 4 + 127) / 128 + 1) * 8;      // 16-byte slots per row, whole lines
        const int n_slots = rows * vq;
        int r = threadIdx.x / vq, v = threadIdx.x - r * vq;
        const int dr = blockDim.x / vq, dv = blockDim.x - dr * vq;
        const uintptr_t clip_base = reinterpret_cast<uintptr_t>(o);
        for (int item = threadIdx.x; item < n_slots; item += blockDim.x) {
            const int mel = mel_lo + r;
            const uintptr_t start = clip_base + ((size_t)mel * n_frames + pad_frame) * 4;
            const uintptr_t end = clip_base + ((size_t)mel * n_frames + n_frames) * 4;
            const uintptr_t a = (start & ~(uintptr_t)127) + 16u * (uintptr_t)v;
            if (a >= start && a < end) {
                const bool mmasked = ((mel < 64 ? melmask_lo >> mel : melmask_hi >> (mel - 64)) & 1ull);
                const float c = mmasked ? 0.0f : pad_y;
                *reinterpret_cast<f32x4*>(a) = f32x4{c, c, c, c};
            } else if ((a < start && a >= clip_base + (size_t)mel * n_frames * 4) ||
                       (a >= end && a < ((end + 127) & ~(uintptr_t)127) && r + 1 < rows &&
                        a < end + (size_t)pad_frame * 4)) {
                // (the tail of a row's last line is kept only where it holds COMPUTED frames of this workgroup's next
                // row: with short clips or short rows -- pad_frame * 4 < 128 bytes, compact n_frames < 32 -- the line
                // reaches that row's own fill segment, which its fill lane writes, or rows of another workgroup)
                const f32x4 keep = *reinterpret_cast<const f32x4*>(a);          // computed cells of this workgroup's rows
                *reinterpret_cast<f32x4*>(a) = keep;
            }
            r += dr;
            v += dv;
            if (v >= vq) {
                v -= vq;
                ++r;
            }
        }
        if (nt != 0) {                                             // then the few time-masked columns are zeroed
            __syncthreads();
            for (int i = 0; i < nt; ++i) {
                const int a = max(ts[i], pad_frame), w = min(te[i], n_frames) - a;
                if (w <= 0) continue;
                for (int e = threadIdx.x; e < rows * w; e += blockDim.x) {
                    const int r = e / w, c = e - r * w;
                    o[(size_t)(mel_lo + r) * n_frames + a + c] = 0.0f;
                }
            }
        }
    }
    if (flags == 0ull) return;

    // ---- clamp: (flagged tile, row, 4-frame quad) items are dealt round-robin to the threads; a thread loads up to
    // kBatch items before its first store (the compiler cannot move a load above a store to the same array, and
    // one HBM round trip per item would make the pass latency-bound).
    __shared__ unsigned char s_tiles[64];
    const int n_flagged = __popcll(flags);
    if (threadIdx.x < 64 && ((flags >> threadIdx.x) & 1ull))
        s_tiles[__popcll(flags & ((1ull << threadIdx.x) - 1ull))] = (unsigned char)threadIdx.x;
    __syncthreads();
    constexpr int kBatch = 4;
    constexpr int kQuadsPerTile = kTileFrames / 4;                 // 12
    const int per_tile = rows * kQuadsPerTile, n_items = n_flagged * per_tile;
    for (int base = threadIdx.x; base < n_items; base += kBatch * blockDim.x) {
        int fr0[kBatch], melv[kBatch];
        f32x4 v[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int item = base + i * blockDim.x;
            fr0[i] = -1;
            if (item < n_items) {
                const int ti = item / per_tile, rem = item - ti * per_tile;
                const int r = rem / kQuadsPerTile, qt = rem - r * kQuadsPerTile;
                const int fr = s_tiles[ti] * kTileFrames + 4 * qt;
                melv[i] = mel_lo + r;
                if (fr < pad_frame) {                              // pad_frame is a multiple of 4
                    fr0[i] = fr;
                    v[i] = *reinterpret_cast<const f32x4*>(o + (size_t)melv[i] * n_frames + fr);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            if (fr0[i] < 0) continue;
            const int mel = melv[i];
            const bool mmasked = ((mel < 64 ? melmask_lo >> mel : melmask_hi >> (mel - 64)) & 1ull);
            f32x4 w;
            bool changed = false;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int fr = fr0[i] + c;
                float y = fr < key_frames ? fmaxf(v[i][c], floor_y) : v[i][c];   // padding_value cells stay as they are
                bool masked = mmasked;                             // masked cells of a computed tile already hold 0.0
                for (int m = 0; m < nt; ++m) masked |= (fr >= ts[m] && fr < te[m]);
                if (masked) y = 0.0f;
                changed |= (y != v[i][c]);
                w[c] = y;
            }
            if (changed) *reinterpret_cast<f32x4*>(o + (size_t)mel * n_frames + fr0[i]) = w;
        }
    }
}

#include "whisper_logmel12.inc"
}  // namespace sttfe

using namespace sttfe;

#ifdef STTFE_STAMPS
extern "C" int sttfe_debug_set_stamp_buffer(void* dev_buf) {
    unsigned long long* p = reinterpret_cast<unsigned long long*>(dev_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -4;
}
#endif

extern "C" int sttfe_whisper_workspace_bytes(int batch, size_t* bytes) {
    if (batch < 0 || !bytes) return STTFE_E_INVALID_ARG;
    *bytes = (size_t)batch * wh::kKeysPerClip * sizeof(unsigned) + 64;      // keys + the work-item ticket counter
    return STTFE_OK;
}

// Consecutive 48-frame tiles one workgroup of kernel 1 walks (tuning knob; 0 = automatic: long runs
// when the batch alone fills the chip, shorter ones for small batches).
static std::atomic<int> g_whisper_tiles_per_block{0};       // read once per call: calls on other threads see old or new
extern "C" int sttfe_set_whisper_tiles_per_block(int tiles) {
    if (tiles < 0 || tiles > wh::kMaxTiles) return STTFE_E_INVALID_ARG;
    g_whisper_tiles_per_block.store(tiles, std::memory_order_relaxed);
    return STTFE_OK;
}

namespace sttfe {
// Compute units of the CURRENT device, looked up once per device (the library may serve several GPUs from one process).
int current_device_cus() {
    static std::atomic<int> cus_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int cus = cus_of[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    return cus;
}
}  // namespace sttfe

// The 12-wave kernel for uniform batches (whisper_logmel12.inc): on by default; sttfe_set_whisper_uniform_kernel(0) or
// STTFE_WH_UNIFORM_KERNEL=0 sends every call to whisper_stft_mel_kernel (A/B runs, parity tests).  The diagnostic builds
// (phase stamps, floor builds) measure whisper_stft_mel_kernel and start with it off.
static std::atomic<int> g_whisper_uniform_kernel{-1};       // -1: not set through the API
extern "C" int sttfe_set_whisper_uniform_kernel(int on) {
    if (on != 0 && on != 1) return STTFE_E_INVALID_ARG;
    g_whisper_uniform_kernel.store(on, std::memory_order_relaxed);
    return STTFE_OK;
}
static bool whisper_uniform_kernel_enabled() {
    const int api = g_whisper_uniform_kernel.load(std::memory_order_relaxed);
    if (api >= 0) return api != 0;
    static const int env = [] {
        const char* e = getenv("STTFE_WH_UNIFORM_KERNEL");
#if defined(STTFE_STAMPS) || STTFE_FLOOR
        return e ? (atoi(e) != 0) : 0;
#else
        return e ? (atoi(e) != 0) : 1;
#endif
    }();
    return env != 0;
}
// The kernel needs 147 KB of dynamic LDS: raised once per device (0 = not tried, 1 = ready, 2 = refused -> never used).
static bool whisper_uniform_kernel_ready() {
    static std::atomic<int> state[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    int st = state[dev].load(std::memory_order_acquire);
    if (st == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(whisper_stft_mel12_kernel<80>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wh12::kLdsFloats * sizeof(float)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(whisper_stft_mel12_kernel<128>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wh12::kLdsFloats * sizeof(float)));
        if (e != hipSuccess) (void)hipGetLastError();
        st = e == hipSuccess ? 1 : 2;
        state[dev].store(st, std::memory_order_release);
    }
    return st == 1;
}

static int whisper_logmel_impl(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                               int batch, int n_mels, int n_frames, const void* dev_tables,
                               float* out, const int32_t* mask_tbl, int n_masks, const float* clip_scale,
                               void* workspace, size_t workspace_bytes, void* stream_, int mode = wh::kModeHF,
                               float pad_value = 0.0f) {
    if (batch < 0 || n_mels < 1 || n_mels > 128 || n_frames < 1 || (n_frames & 3) || n_masks < 0 || n_masks > STTFE_MAX_MASKS)
        return STTFE_E_INVALID_ARG;
    if (batch == 0) return STTFE_OK;
    if (!pcm || !offsets || !lengths || !dev_tables || !out || !workspace) return STTFE_E_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(out) & 15) return STTFE_E_INVALID_ARG;      // rows are written as 16-byte vectors
    if ((size_t)wh::kHop * (size_t)n_frames > (size_t)wh::kMaxSamples || batch > 65535) return STTFE_E_UNSUPPORTED;
    size_t need = 0;
    sttfe_whisper_workspace_bytes(batch, &need);
    if (workspace_bytes < need) return STTFE_E_WORKSPACE;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    unsigned* keys = reinterpret_cast<unsigned*>(workspace);
    (void)hipGetLastError();      // drop any stale error left by earlier, unrelated HIP calls of this thread
    if (hip_failed(hipMemsetAsync(keys, 0, need, stream), "hipMemsetAsync(keys)")) return STTFE_E_LAUNCH;

    const int n_tiles = (n_frames + wh::kTileFrames - 1) / wh::kTileFrames;
    static const int tpb_env = [] {
        const char* e = getenv("STTFE_WH_TILES_PER_BLOCK");       // tuning knob, as sttfe_set_whisper_tiles_per_block
        const int v = e ? atoi(e) : 0;
        return v < 0 || v > wh::kMaxTiles ? 0 : v;
    }();
    const int tpb_knob = g_whisper_tiles_per_block.load(std::memory_order_relaxed);
    int tiles_per_block = tpb_knob ? tpb_knob : tpb_env;
    // Kernel 1 runs as persistent workgroups (two per CU) that walk (tile run, clip) items; short runs keep the
    // workgroups balanced when clip lengths differ, and the tables are staged once per workgroup, not per run.
    // tiles_per_block == 0: the kernel picks the run length itself from the clip lengths it finds on the device
    if ((long)n_tiles * batch > 0x3fffffffL) return STTFE_E_UNSUPPORTED;
    const int resident = 2 * current_device_cus();
    const long most_items = (long)batch * ((n_tiles + 2) / 3);      // fewer workgroups than that cannot all find work
    const int grid = most_items < resident ? (int)most_items : resident;
    const auto* tab = reinterpret_cast<const WhisperTablesHeader*>(dev_tables);
    // Calls of the 80- and 128-bin extractors in HF mode (with or without SpecAugment masks or a per-clip scale) are computed by
    // whisper_stft_mel12_kernel (whisper_logmel12.inc).  Batches of up to wh12::kPlanClips clips: by that kernel ALONE,
    // uniform or ragged (it builds the ragged batch's item plan in its LDS) -- one STFT launch.  Larger batches: only if
    // uniform, which is known on the device alone, so such a call launches BOTH kernels and each returns at once if the
    // batch is the other's (the same predicate on the same lengths: whisper_batch_is_uniform).
    // STTFE_WH12_RAGGED=0: ragged batches of any size stay with whisper_stft_mel_kernel (A/B and test knob).
    static const bool ragged12 = [] {
        const char* e = getenv("STTFE_WH12_RAGGED");
        return !(e && atoi(e) == 0);
    }();
    // STTFE_WH12_MFMA=1: the mel waves project with MFMAs (mel_unit12) instead of the band-sparse packed-VALU form (A/B knob;
    // the same bits either way)
    static const int mfma_flag = [] {
        const char* e = getenv("STTFE_WH12_MFMA");
        return (e && atoi(e) != 0) ? 2 : 0;
    }();
    const bool try_uniform = whisper_uniform_kernel_enabled() && (n_mels == 80 || n_mels == 128) && mode == wh::kModeHF &&
                             batch <= 16384 && whisper_uniform_kernel_ready();
    // (one clip is a uniform batch whatever its length)
    const bool sure_uniform = try_uniform && (batch == 1 || (ragged12 && batch <= wh12::kPlanClips));
    if (try_uniform) {
        ScopedTiming t(kTimeWhisperStftMel12, stream);
        const int cus = current_device_cus();
        const long tiles12 = (long)batch * n_tiles;                   // (the kernel walks single tiles when the batch is small)
        const int grid12 = tiles12 < cus ? (int)tiles12 : cus;
        if (n_mels == 80)
            hipLaunchKernelGGL(whisper_stft_mel12_kernel<80>, dim3(grid12), dim3(wh12::kThreads),
                               (size_t)wh12::kLdsFloats * sizeof(float), stream, pcm, offsets, lengths, n_frames, tab, out, keys,
                               batch, tiles_per_block, (sure_uniform ? 0 : 1) | mfma_flag, mask_tbl, n_masks, clip_scale);
        else
            hipLaunchKernelGGL(whisper_stft_mel12_kernel<128>, dim3(grid12), dim3(wh12::kThreads),
                               (size_t)wh12::kLdsFloats * sizeof(float), stream, pcm, offsets, lengths, n_frames, tab, out, keys,
                               batch, tiles_per_block, (sure_uniform ? 0 : 1) | mfma_flag, mask_tbl, n_masks, clip_scale);
    }
    if (!sure_uniform) {
        ScopedTiming t(kTimeWhisperStftMel, stream);
#ifdef STTFE_STAMPS
        // diagnostic build: extra dynamic LDS forces one workgroup per CU (phase times without a co-resident workgroup)
        static const int dyn_lds = getenv("STTFE_WH_DYN_LDS") ? atoi(getenv("STTFE_WH_DYN_LDS")) : 0;
#else
        const int dyn_lds = 0;
#endif
        hipLaunchKernelGGL(whisper_stft_mel_kernel, dim3(grid), dim3(wh::kThreads), dyn_lds, stream,
                           pcm, offsets, lengths, n_mels, n_frames, tab, out, keys, mask_tbl, n_masks, tiles_per_block,
                           batch, keys + (size_t)batch * wh::kKeysPerClip, clip_scale, mode, pad_value, try_uniform ? 1 : 0);
    }
    {
        ScopedTiming t(kTimeWhisperClamp, stream);
        // 4-5 mel rows per workgroup measured best for the fill (4.3 TB/s) without slowing the full-length case
        static const int row_split_env = [] {
            const char* e = getenv("STTFE_ROW_SPLIT");            // tuning knob (0 / unset = automatic)
            const int v = e ? atoi(e) : 0;
            return v < 0 ? 0 : (v > 128 ? 128 : v);
        }();
        const int row_split = row_split_env ? row_split_env : (n_mels > 96 ? (n_mels + 3) / 4 : (n_mels + 4) / 5);
        hipLaunchKernelGGL(whisper_clamp_fill_kernel, dim3(row_split, batch), dim3(256), 0, stream, out, lengths, keys,
                           n_mels, n_frames, mask_tbl, n_masks, clip_scale, mode, pad_value, tab);
    }
    if (hip_failed(hipGetLastError(), "whisper kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}

extern "C" int sttfe_whisper_logmel(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                    int batch, int n_mels, int n_frames, const void* dev_tables,
                                    float* out, const int32_t* mask_tbl, int n_masks,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    return whisper_logmel_impl(pcm, offsets, lengths, batch, n_mels, n_frames, dev_tables, out, mask_tbl, n_masks,
                               nullptr, workspace, workspace_bytes, stream);
}

// TorchWhisperFeatureExtractor semantics (reference src/processors.py:180-239), see include/sttfe.h
extern "C" int sttfe_whisper_logmel_torch(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                          int batch, int n_mels, int n_frames, const void* dev_tables,
                                          float* out, const int32_t* mask_tbl, int n_masks, int variable_length,
                                          float padding_value, void* workspace, size_t workspace_bytes, void* stream) {
    if (variable_length != 0 && variable_length != 1) return STTFE_E_INVALID_ARG;
    if (!variable_length && n_frames != wh::kMaxSamples / wh::kHop) return STTFE_E_INVALID_ARG;   // 3001 frames -> 3000
    return whisper_logmel_impl(pcm, offsets, lengths, batch, n_mels, n_frames, dev_tables, out, mask_tbl, n_masks,
                               nullptr, workspace, workspace_bytes, stream,
                               variable_length ? wh::kModeTorchVarlen : wh::kModeTorchPadded, padding_value);
}

// ---- the entry point that allocates nothing (no counterpart in the reference; include/sttfe.h) ----
// The per-call workspace of sttfe_whisper_logmel (per-clip max / per-tile min keys + the work-item ticket counter)
// lives in a file-scope device array instead of a caller-provided buffer.  Every GPU the library runs on holds its
// own copy of the array at its own address until the process ends; the address is looked up once per device.
namespace sttfe {
namespace wh {
constexpr int kStaticMaxBatch = STTFE_WHISPER_STATIC_MAX_BATCH;
constexpr size_t kStaticWords = (size_t)kStaticMaxBatch * kKeysPerClip + 16;     // keys + the ticket counter (64 B)
constexpr int kMaxDevices = 64;
}  // namespace wh
static __device__ __align__(256) unsigned scratch[wh::kStaticWords];
static std::mutex g_scratch_mu;
static unsigned* g_scratch_ptr[wh::kMaxDevices] = {};     // device address of `scratch` on each device (never the name)
}  // namespace sttfe

extern "C" int sttfe_whisper_logmel_static(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                           int batch, int n_mels, int n_frames, const void* dev_tables,
                                           float* out, const int32_t* mask_tbl, int n_masks, void* stream) {
    if (batch < 0) return STTFE_E_INVALID_ARG;
    if (batch > wh::kStaticMaxBatch) {
        std::snprintf(g_last_detail, sizeof(g_last_detail),
                      "sttfe_whisper_logmel_static: batch %d exceeds the static workspace (%d clips)", batch,
                      wh::kStaticMaxBatch);
        return STTFE_E_UNSUPPORTED;
    }
    if (batch == 0) return STTFE_OK;
    int dev = -1;
    if (hip_failed(hipGetDevice(&dev), "hipGetDevice")) return STTFE_E_LAUNCH;
    if (dev < 0 || dev >= wh::kMaxDevices) return STTFE_E_UNSUPPORTED;
    unsigned* ws = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[dev]) {                   // first call on this device (which is current): resolve its copy
            void* p = nullptr;
            if (hip_failed(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)") || !p)
                return STTFE_E_LAUNCH;
            g_scratch_ptr[dev] = reinterpret_cast<unsigned*>(p);
        }
        ws = g_scratch_ptr[dev];
    }
    // whisper_logmel_impl zeroes the part of the workspace this batch uses (hipMemsetAsync on `stream`) before the
    // kernels read it, so whatever the previous call left behind is cleared; it calls no allocating routine.
    size_t need = 0;
    sttfe_whisper_workspace_bytes(batch, &need);
    return whisper_logmel_impl(pcm, offsets, lengths, batch, n_mels, n_frames, dev_tables, out, mask_tbl, n_masks,
                               nullptr, ws, need, stream);
}

extern "C" int sttfe_whisper_logmel_scaled(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                           int batch, int n_mels, int n_frames, const void* dev_tables,
                                           float* out, const int32_t* mask_tbl, int n_masks,
                                           const float* clip_scale, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    if (batch > 0 && !clip_scale) return STTFE_E_INVALID_ARG;
    return whisper_logmel_impl(pcm, offsets, lengths, batch, n_mels, n_frames, dev_tables, out, mask_tbl, n_masks,
                               clip_scale, workspace, workspace_bytes, stream);
}
