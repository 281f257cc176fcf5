// Layout of the constant "tables" blobs shared by the host builders (tables.cpp) and the kernels.
#pragma once
#include <stdint.h>

namespace sttfe {

constexpr int kMaxMelTiles = 8;          // 128 mel bins / 16 rows per MFMA tile
constexpr int kMaxUnitsPerWave = 8;
constexpr int kWavesPerBlock = 8;
constexpr int kFrameTiles = 3;           // 48 frames per block iteration / 16 columns per MFMA tile

constexpr uint32_t kWhisperMagic = 0x57484d31u;   // "WHM1"
constexpr uint32_t kKaldiMagic = 0x4b414c31u;     // "KAL1"
constexpr uint32_t kResampleMagic = 0x52534d31u;  // "RSM1"

// Banded mel projection on 16x16x4 f32 MFMA tiles.  Mel tile mt covers mel rows [16 mt, 16 mt + 16)
// and FFT bins [kstart, kstart + 4 nsteps).  afrag holds, for step s and lane l, the A operand
// W[16 mt + (l & 15)][kstart + 4 s + (l >> 4)] (already multiplied by any constant scale).
struct MelPlan {
    int32_t n_mels;
    int32_t n_mtiles;
    int32_t n_bins;
    int32_t reserved;
    int32_t kstart[kMaxMelTiles];
    int32_t nsteps[kMaxMelTiles];
    int32_t aoff[kMaxMelTiles];                       // float offset into afrag
    int32_t wave_nunits[kWavesPerBlock];
    int32_t wave_units[kWavesPerBlock][kMaxUnitsPerWave];   // mt | (ft << 8)
    // the same plan as ready-made unit descriptors {kstart, nsteps, aoff, mt | ft << 8} (zeros past wave_nunits):
    // a kernel stages them with one 16-byte load per unit instead of two dependent table look-ups
    int32_t desc[kWavesPerBlock][kMaxUnitsPerWave][4];
};
static_assert(sizeof(MelPlan) % 16 == 0, "descriptors are read as 16-byte words");

struct KaldiTablesHeader {
    uint32_t magic;
    int32_t n_mels;
    int32_t afrag_floats;
    int32_t reserved;
    MelPlan plan;               // 4 waves, 2 frame tiles (32-frame workgroup tile)
    float window[400];          // Povey window, symmetric hanning(400)^0.85
    float twiddle[1024];        // [k1][n2] complex: W_512^(n2 k1), k1 < 16, n2 < 32
    // followed by: float afrag[afrag_floats]
};

struct WhisperTablesHeader {
    uint32_t magic;
    int32_t n_mels;
    int32_t afrag_floats;
    int32_t reserved;           // bits 0, 1: whisper_mel_bands.h describes this bank / the weight-pair stream follows the A fragments
    MelPlan plan;
    float window[400];          // periodic Hann
    float twiddle[800];         // [k1][j] complex: W_400^(j k1) = (cos, -sin)(2 pi j k1 / 400)
    // followed by: float afrag[afrag_floats]; then, if reserved & 2, the weight pairs of every mel in the order of the
    // band-sparse projection's chain (2 * MelBand::woff[n_mels] floats; whisper_logmel12.inc reads them with scalar loads)
};

// Polyphase windowed-sinc resampler (torchaudio.transforms.Resample defaults).  Output sample
// i * new_rate + p = sum_k taps[p][k] * padded[i * orig_rate + k], padded = input with `width` zeros in front.
struct ResamplePlanHeader {
    uint32_t magic;
    int32_t orig_rate;          // rates divided by their gcd
    int32_t new_rate;
    int32_t width;
    int32_t n_taps;             // 2 width + orig_rate
    int32_t taps_row;           // floats per phase row: n_taps rounded up to a multiple of orig_rate (zero filled)
    int32_t reserved[2];
    // followed by: int32 lo[new_rate], int32 hi[new_rate] (non-zero tap span [lo, hi) of each phase),
    //              float taps[new_rate][taps_row]
};

}  // namespace sttfe
