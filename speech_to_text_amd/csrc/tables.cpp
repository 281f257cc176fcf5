// Host-side builders for the constant tables (mel banks, windows, twiddles, MFMA mel plans).
// Pure C++ double-precision math; no GPU calls.  The filter-bank definitions follow the formulas
// of the Hugging Face extractors the reference calls (HF:audio_utils.py:448-517,541-560,638-729);
// tests/test_boundary_cpu.py checks the banks and the resampler taps against the NumPy oracle.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/sttfe.h"
#include "tables.h"
#include "whisper_mel_bands.h"

namespace sttfe {

static double hz_to_mel_slaney(double f) {
    if (f >= 1000.0) return 15.0 + std::log(f / 1000.0) * (27.0 / std::log(6.4));
    return 3.0 * f / 200.0;
}
static double mel_to_hz_slaney(double m) {
    if (m >= 15.0) return 1000.0 * std::exp((std::log(6.4) / 27.0) * (m - 15.0));
    return 200.0 * m / 3.0;
}
static double hz_to_mel_kaldi(double f) { return 1127.0 * std::log(1.0 + f / 700.0); }

// Triangular bank (n_mels x n_bins), given the coordinates of the FFT bins and of the n_mels + 2
// filter edge points in the same unit (Hz for Whisper, mel for Kaldi).
static void triangular_bank(const std::vector<double>& bin_pos, const std::vector<double>& pts,
                            int n_mels, std::vector<double>& w) {
    const int n_bins = (int)bin_pos.size();
    w.assign((size_t)n_mels * n_bins, 0.0);
    for (int m = 0; m < n_mels; ++m) {
        const double lo = pts[m], ce = pts[m + 1], hi = pts[m + 2];
        for (int k = 0; k < n_bins; ++k) {
            const double down = -(lo - bin_pos[k]) / (ce - lo);   // rising edge
            const double up = (hi - bin_pos[k]) / (hi - ce);      // falling edge
            w[(size_t)m * n_bins + k] = std::max(0.0, std::min(down, up));
        }
    }
}

void whisper_mel_bank(int n_mels, std::vector<float>& dense) {
    const int n_bins = 201;
    std::vector<double> pts(n_mels + 2), bins(n_bins), w;
    const double m0 = hz_to_mel_slaney(0.0), m1 = hz_to_mel_slaney(8000.0);
    for (int i = 0; i < n_mels + 2; ++i)
        pts[i] = mel_to_hz_slaney(m0 + (m1 - m0) * (double)i / (double)(n_mels + 1));
    for (int k = 0; k < n_bins; ++k) bins[k] = 8000.0 * (double)k / (double)(n_bins - 1);
    triangular_bank(bins, pts, n_mels, w);
    dense.resize(w.size());
    for (int m = 0; m < n_mels; ++m) {
        const double enorm = 2.0 / (pts[m + 2] - pts[m]);          // slaney area normalisation
        for (int k = 0; k < n_bins; ++k)
            dense[(size_t)m * n_bins + k] = (float)(w[(size_t)m * n_bins + k] * enorm);
    }
}

static void kaldi_mel_bank_f64(int n_mels, std::vector<double>& w) {
    const int n_bins = 257;
    std::vector<double> pts(n_mels + 2), bins(n_bins);
    const double m0 = hz_to_mel_kaldi(20.0), m1 = hz_to_mel_kaldi(8000.0);
    for (int i = 0; i < n_mels + 2; ++i) pts[i] = m0 + (m1 - m0) * (double)i / (double)(n_mels + 1);
    const double bin_width = 16000.0 / ((n_bins - 1) * 2);
    for (int k = 0; k < n_bins; ++k) bins[k] = hz_to_mel_kaldi(bin_width * k);
    triangular_bank(bins, pts, n_mels, w);
}

void kaldi_mel_bank(int n_mels, std::vector<float>& dense) {
    std::vector<double> w;
    kaldi_mel_bank_f64(n_mels, w);
    dense.resize(w.size());
    for (size_t i = 0; i < w.size(); ++i) dense[i] = (float)w[i];
}

// Build the banded MFMA plan + A fragments from a dense (n_mels x n_bins) bank times `scale`.
void build_mel_plan(const std::vector<float>& dense, int n_mels, int n_bins, float scale,
                    int n_waves, int n_ftiles, int step_multiple, int row_floats, MelPlan& plan,
                    std::vector<float>& afrag) {
    std::memset(&plan, 0, sizeof(plan));
    plan.n_mels = n_mels;
    plan.n_bins = n_bins;
    plan.n_mtiles = (n_mels + 15) / 16;
    afrag.clear();
    for (int mt = 0; mt < plan.n_mtiles; ++mt) {
        int lo = n_bins, hi = -1;
        for (int r = 0; r < 16; ++r) {
            const int m = mt * 16 + r;
            if (m >= n_mels) break;
            for (int k = 0; k < n_bins; ++k)
                if (dense[(size_t)m * n_bins + k] != 0.0f) {
                    lo = std::min(lo, k);
                    hi = std::max(hi, k);
                }
        }
        int nsteps = 0, kstart = 0;
        if (hi >= lo) {
            if (row_floats > 0) lo &= ~3;             // grouped layout: 16-byte aligned operand reads
            nsteps = (hi - lo + 1 + 3) / 4;
            // zero-weight steps pad the band to a multiple of step_multiple (the kernel batches its loads)
            nsteps = (nsteps + step_multiple - 1) / step_multiple * step_multiple;
            // never read past the row: the panel row holds n_bins values + finite pad cells
            const int row_len = row_floats > 0 ? row_floats : n_bins + 1;
            kstart = std::max(0, std::min(lo, row_len - 4 * nsteps));
        }
        plan.kstart[mt] = kstart;
        plan.nsteps[mt] = nsteps;
        plan.aoff[mt] = (int32_t)afrag.size();
        if (row_floats > 0) {
            // grouped: [group of 4 steps][lane][step in group]; step i of group g covers bins
            // kstart + 16 g + 4 (lane >> 4) + i, so a lane's four A values and four B values are one b128 each
            for (int g = 0; g < nsteps / 4; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int i = 0; i < 4; ++i) {
                        const int m = mt * 16 + (l & 15);
                        const int k = kstart + 16 * g + 4 * (l >> 4) + i;
                        float v = 0.0f;
                        if (m < n_mels && k < n_bins) v = dense[(size_t)m * n_bins + k] * scale;
                        afrag.push_back(v);
                    }
            continue;
        }
        for (int s = 0; s < nsteps; ++s)
            for (int l = 0; l < 64; ++l) {
                const int m = mt * 16 + (l & 15);
                const int k = kstart + 4 * s + (l >> 4);
                float v = 0.0f;
                if (m < n_mels && k < n_bins) v = dense[(size_t)m * n_bins + k] * scale;
                afrag.push_back(v);
            }
    }
    // Longest-processing-time assignment of (mel tile, frame tile) units to the 8 waves.
    struct Unit { int mt, ft, cost; };
    std::vector<Unit> units;
    for (int ft = 0; ft < n_ftiles; ++ft)
        for (int mt = 0; mt < plan.n_mtiles; ++mt) units.push_back({mt, ft, plan.nsteps[mt] + 3});
    std::stable_sort(units.begin(), units.end(), [](const Unit& a, const Unit& b) { return a.cost > b.cost; });
    int load[kWavesPerBlock] = {0};
    for (const Unit& u : units) {
        int best = 0;
        for (int w = 1; w < n_waves; ++w)
            if (load[w] < load[best]) best = w;
        // n_mtiles * kFrameTiles = 24 units over 8 waves: LPT never needs more than kMaxUnitsPerWave
        plan.wave_units[best][plan.wave_nunits[best]++] = u.mt | (u.ft << 8);
        load[best] += u.cost;
    }
    for (int w = 0; w < n_waves; ++w)
        for (int u = 0; u < plan.wave_nunits[w]; ++u) {
            const int code = plan.wave_units[w][u], mt = code & 0xff;
            plan.desc[w][u][0] = plan.kstart[mt];
            plan.desc[w][u][1] = plan.nsteps[mt];
            plan.desc[w][u][2] = plan.aoff[mt];
            plan.desc[w][u][3] = code;
        }
}

}  // namespace sttfe

using namespace sttfe;

extern "C" int sttfe_whisper_mel_bank(int n_mels, float* host_dense) {
    if (n_mels < 1 || n_mels > 128 || !host_dense) return STTFE_E_INVALID_ARG;
    std::vector<float> dense;
    whisper_mel_bank(n_mels, dense);
    std::memcpy(host_dense, dense.data(), dense.size() * sizeof(float));
    return STTFE_OK;
}

// True when the generated per-mel band extents (whisper_mel_bands.h) are exactly the bank's: first / last non-zero bin of
// every mel row.  The 12-wave kernel's band-sparse projection skips every product outside them.
template <int NM>
static bool bands_match(const std::vector<float>& dense) {
    for (int m = 0; m < NM; ++m)
        for (int k = 0; k < 201; ++k) {
            const bool inside = k >= wh::MelBand<NM>::lo[m] && k <= wh::MelBand<NM>::hi[m];
            const bool edge = k == wh::MelBand<NM>::lo[m] || k == wh::MelBand<NM>::hi[m];
            const float w = dense[(size_t)m * 201 + k];
            if ((!inside && w != 0.0f) || (edge && w == 0.0f)) return false;
        }
    return true;
}
// The weight pairs (W[m][k], W[m][k + 1]) * scale of every mel in the order in which the band-sparse projection's chain
// visits them -- the MFMA form's accumulation order (mel_sparse_dot in whisper_logmel12.inc walks the same loops) -- as one
// contiguous stream behind the A fragments: the mel waves read it with scalar loads, MelBand::woff[m] is mel m's first pair.
template <int NM>
static void pair_stream(const std::vector<float>& dense, float scale, std::vector<float>& out) {
    using B = wh::MelBand<NM>;
    for (int m = 0; m < NM; ++m) {
        const int ks = B::kstart[m / 16], ng = B::nsteps[m / 16] / 4, lo = B::lo[m], hi = B::hi[m];
        int n = 0;
        for (int g = 0; g < ng; ++g) {
            const int k0 = ks + 16 * g;
            if (k0 + 15 < lo || k0 > hi) continue;
            for (int h = 0; h < 2; ++h)
                for (int kq = 0; kq < 4; ++kq) {
                    const int k = k0 + 4 * kq + 2 * h;
                    if (k + 1 < lo || k > hi) continue;
                    for (int i = 0; i < 2; ++i) out.push_back(k + i < 201 ? dense[(size_t)m * 201 + k + i] * scale : 0.0f);
                    ++n;
                }
        }
        if (n != B::woff[m + 1] - B::woff[m]) {   // stale header: no stream, the kernel keeps the MFMA form
            out.clear();
            return;
        }
    }
}
static bool whisper_bands_match(const std::vector<float>& dense, int n_mels) {
    return n_mels == 80 ? bands_match<80>(dense) : (n_mels == 128 ? bands_match<128>(dense) : false);
}
template <int NM>
static bool plan_is(const MelPlan& p) {
    for (int mt = 0; mt < NM / 16; ++mt)
        if (p.kstart[mt] != wh::MelBand<NM>::kstart[mt] || p.nsteps[mt] != wh::MelBand<NM>::nsteps[mt]) return false;
    return true;
}
static bool plan_equals_bands(const MelPlan& p, int n_mels) {
    return n_mels == 80 ? plan_is<80>(p) : (n_mels == 128 ? plan_is<128>(p) : false);
}

static int whisper_tables(int n_mels, std::vector<unsigned char>& blob) {
    if (n_mels < 1 || n_mels > 128) return STTFE_E_INVALID_ARG;
    std::vector<float> dense, afrag;
    whisper_mel_bank(n_mels, dense);
    WhisperTablesHeader h;
    std::memset(&h, 0, sizeof(h));
    // 0.25 = the 1/4 of the two-real-frames-in-one-complex-FFT separation, folded into the bank
    // (a power of two: bit-identical to scaling the power spectrum itself).
    build_mel_plan(dense, n_mels, 201, 0.25f, 4, kFrameTiles, 4, 204, h.plan, afrag);   // 4 waves x 3 frame tiles, grouped b128 layout, 204-float panel rows
    h.magic = kWhisperMagic;
    h.n_mels = n_mels;
    h.afrag_floats = (int32_t)afrag.size();
    // bits 0 + 1: whisper_mel_bands.h describes this bank and the pair stream follows the A fragments (band-sparse projection)
    std::vector<float> stream;
    if (whisper_bands_match(dense, n_mels) && (int)plan_equals_bands(h.plan, n_mels)) {
        if (n_mels == 80) pair_stream<80>(dense, 0.25f, stream);
        else pair_stream<128>(dense, 0.25f, stream);
    }
    h.reserved = stream.empty() ? 0 : 3;   // bit 1: the stream is there (bit 0 alone: a blob of the build that read the weights from LDS)
    for (int i = 0; i < 400; ++i) h.window[i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / 400.0));
    for (int k1 = 0; k1 < 20; ++k1)
        for (int j = 0; j < 20; ++j) {
            const int m = (j * k1) % 400;
            h.twiddle[2 * (20 * k1 + j)] = (float)std::cos(2.0 * M_PI * m / 400.0);
            h.twiddle[2 * (20 * k1 + j) + 1] = (float)(-std::sin(2.0 * M_PI * m / 400.0));
        }
    blob.resize(sizeof(h) + (afrag.size() + stream.size()) * sizeof(float));
    std::memcpy(blob.data(), &h, sizeof(h));
    std::memcpy(blob.data() + sizeof(h), afrag.data(), afrag.size() * sizeof(float));
    if (!stream.empty()) std::memcpy(blob.data() + sizeof(h) + afrag.size() * sizeof(float), stream.data(), stream.size() * sizeof(float));
    return STTFE_OK;
}

extern "C" int sttfe_whisper_tables_bytes(int n_mels, size_t* bytes) {
    if (!bytes) return STTFE_E_INVALID_ARG;
    std::vector<unsigned char> blob;
    int rc = whisper_tables(n_mels, blob);
    if (rc != STTFE_OK) return rc;
    *bytes = blob.size();
    return STTFE_OK;
}

extern "C" int sttfe_whisper_tables_build(int n_mels, void* host_blob, size_t bytes) {
    if (!host_blob) return STTFE_E_INVALID_ARG;
    std::vector<unsigned char> blob;
    int rc = whisper_tables(n_mels, blob);
    if (rc != STTFE_OK) return rc;
    if (bytes < blob.size()) return STTFE_E_TABLES;
    std::memcpy(host_blob, blob.data(), blob.size());
    return STTFE_OK;
}

// ------------------------------------------------------------------ Kaldi fbank tables --------
extern "C" int sttfe_kaldi_mel_bank(int n_mels, float* host_dense) {
    if (n_mels < 1 || n_mels > 128 || !host_dense) return STTFE_E_INVALID_ARG;
    std::vector<float> dense;
    kaldi_mel_bank(n_mels, dense);
    std::memcpy(host_dense, dense.data(), dense.size() * sizeof(float));
    return STTFE_OK;
}

static int kaldi_tables(int n_mels, std::vector<unsigned char>& blob) {
    if (n_mels < 1 || n_mels > 128) return STTFE_E_INVALID_ARG;
    std::vector<float> dense, afrag;
    kaldi_mel_bank(n_mels, dense);
    KaldiTablesHeader h;
    std::memset(&h, 0, sizeof(h));
    // 2^28 = (2^15)^2 [the x * 2**15 of HF speech_to_text.py:112 applied to the power spectrum] * 1/4
    // [two-real-frames separation]; a power of two, so folding it into the bank is exact.
    build_mel_plan(dense, n_mels, 257, 268435456.0f, 4, 2, 4, 0, h.plan, afrag);   // 8 waves x 4 frame tiles, bands padded to 4 steps
    h.magic = kKaldiMagic;
    h.n_mels = n_mels;
    h.afrag_floats = (int32_t)afrag.size();
    for (int i = 0; i < 400; ++i)
        h.window[i] = (float)std::pow(0.5 - 0.5 * std::cos(2.0 * M_PI * i / 399.0), 0.85);
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < 32; ++n2) {
            const int m = (n2 * k1) % 512;
            h.twiddle[2 * (32 * k1 + n2)] = (float)std::cos(2.0 * M_PI * m / 512.0);
            h.twiddle[2 * (32 * k1 + n2) + 1] = (float)(-std::sin(2.0 * M_PI * m / 512.0));
        }
    // float64 section (STTFE_KALDI_F64): Povey window and W_512^m = (cos, -sin)(2 pi m / 512), 8-byte aligned behind afrag
    const size_t f64_off = (sizeof(h) + afrag.size() * sizeof(float) + 7) & ~(size_t)7;
    std::vector<double> t64(400 + 2 * 512);
    for (int i = 0; i < 400; ++i) t64[i] = std::pow(0.5 - 0.5 * std::cos(2.0 * M_PI * i / 399.0), 0.85);
    for (int m = 0; m < 512; ++m) {
        // exact values on the axes and diagonals; elsewhere long-double evaluation rounded once
        const long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)m / 512.0L;
        t64[400 + 2 * m] = (double)cosl(a);
        t64[400 + 2 * m + 1] = (double)(-sinl(a));
    }
    t64[400 + 2 * 128] = 0.0;       // cos(pi / 2)
    t64[400 + 2 * 384] = 0.0;       // cos(3 pi / 2)
    t64[400 + 2 * 256 + 1] = 0.0;   // -sin(pi)
    t64[400 + 1] = 0.0;             // -sin(0): +0.0
    // the float64 mel bank as per-bin bands: {first bin, bins, offset of the first weight, 0} per mel bin (two doubles'
    // worth of int32), then the weights themselves times 2^28 (see above; exact)
    {
        std::vector<double> w64;
        kaldi_mel_bank_f64(n_mels, w64);
        std::vector<int32_t> band(4 * (size_t)n_mels, 0);
        std::vector<double> wts;
        for (int m = 0; m < n_mels; ++m) {
            int lo = 257, hi = -1;
            for (int k = 0; k < 257; ++k)
                if (w64[(size_t)m * 257 + k] != 0.0) {
                    lo = std::min(lo, k);
                    hi = std::max(hi, k);
                }
            band[4 * m] = hi >= lo ? lo : 0;
            band[4 * m + 1] = hi >= lo ? hi - lo + 1 : 0;
            band[4 * m + 2] = (int32_t)wts.size();
            for (int k = lo; k <= hi; ++k) wts.push_back(w64[(size_t)m * 257 + k] * 268435456.0);
        }
        h.reserved = (int32_t)wts.size();          // number of float64 band weights (<= 2 per FFT bin)
        const size_t base = t64.size();
        t64.resize(base + 2 * (size_t)n_mels + wts.size());
        std::memcpy(&t64[base], band.data(), band.size() * sizeof(int32_t));
        std::memcpy(&t64[base + 2 * (size_t)n_mels], wts.data(), wts.size() * sizeof(double));
    }
    blob.assign(f64_off + t64.size() * sizeof(double), 0);
    std::memcpy(blob.data(), &h, sizeof(h));
    std::memcpy(blob.data() + sizeof(h), afrag.data(), afrag.size() * sizeof(float));
    std::memcpy(blob.data() + f64_off, t64.data(), t64.size() * sizeof(double));
    return STTFE_OK;
}

extern "C" int sttfe_kaldi_tables_bytes(int n_mels, size_t* bytes) {
    if (!bytes) return STTFE_E_INVALID_ARG;
    std::vector<unsigned char> blob;
    int rc = kaldi_tables(n_mels, blob);
    if (rc != STTFE_OK) return rc;
    *bytes = blob.size();
    return STTFE_OK;
}

extern "C" int sttfe_kaldi_tables_build(int n_mels, void* host_blob, size_t bytes) {
    if (!host_blob) return STTFE_E_INVALID_ARG;
    std::vector<unsigned char> blob;
    int rc = kaldi_tables(n_mels, blob);
    if (rc != STTFE_OK) return rc;
    if (bytes < blob.size()) return STTFE_E_TABLES;
    std::memcpy(host_blob, blob.data(), blob.size());
    return STTFE_OK;
}

// ------------------------------------------------------------------ resampler plan ------------
// torchaudio.functional._get_sinc_resample_kernel with the transform's defaults (sinc_interp_hann,
// lowpass_filter_width 6, rolloff 0.99, kernel built in float64 and cast to float32), as restated in
// SURVEY.md section 8a row R.  The reference calls it through torchaudio.transforms.Resample
// (src/data.py:70-72, inference.py:111-113).
static int resample_plan(int orig_sr, int new_sr, std::vector<unsigned char>& blob) {
    if (orig_sr < 1 || new_sr < 1) return STTFE_E_INVALID_ARG;
    int a = orig_sr, b = new_sr;
    while (b) { const int t = a % b; a = b; b = t; }
    const int orig = orig_sr / a, nw = new_sr / a;
    if (orig > (1 << 22) || nw > (1 << 22)) return STTFE_E_UNSUPPORTED;
    const double lowpass_width = 6.0, rolloff = 0.99;
    const double base = (double)std::min(orig, nw) * rolloff;
    const int width = (int)std::ceil(lowpass_width * orig / base);
    const int n_taps = 2 * width + orig;
    const int row = (n_taps + orig - 1) / orig * orig;
    ResamplePlanHeader h;
    std::memset(&h, 0, sizeof(h));
    h.magic = kResampleMagic;
    h.orig_rate = orig;
    h.new_rate = nw;
    h.width = width;
    h.n_taps = n_taps;
    h.taps_row = row;
    if (orig > 1024 || nw > 1024) {
        // Reduced rates beyond 1024 (torchaudio accepts any: speed factors such as 0.97 give 97:100 -- fine -- but
        // int(0.966625 * 16000) = 15466 gives 7733:8000, a 8000 x 7747 table of 248 MB): no table.  The plan is its
        // header alone (taps_row = 0 marks it) and resample_onthefly_kernel evaluates the dozen non-zero taps of an
        // output sample as the table builder below does.
        h.taps_row = 0;
        blob.resize(sizeof(h));
        std::memcpy(blob.data(), &h, sizeof(h));
        return STTFE_OK;
    }
    std::vector<int32_t> lo(nw), hi(nw);
    std::vector<float> taps((size_t)nw * row, 0.0f);
    const double scale = base / orig;
    for (int p = 0; p < nw; ++p) {
        // torch: arange(0, -new, -1) / new is a float32 division, promoted to float64 by the add
        const double phase = (double)((float)(-p) / (float)nw);
        int l = n_taps, r = 0;
        for (int k = 0; k < n_taps; ++k) {
            double t = (phase + (double)(k - width) / (double)orig) * base;
            t = std::max(-lowpass_width, std::min(lowpass_width, t));
            const double c = std::cos(t * M_PI / lowpass_width / 2.0);
            const double window = c * c;
            t *= M_PI;
            const double sinc = (t == 0.0) ? 1.0 : std::sin(t) / t;
            const float v = (float)(sinc * window * scale);
            taps[(size_t)p * row + k] = v;
            if (v != 0.0f) {
                l = std::min(l, k);
                r = std::max(r, k + 1);
            }
        }
        lo[p] = std::min(l, r);
        hi[p] = r;
    }
    blob.resize(sizeof(h) + 2 * (size_t)nw * sizeof(int32_t) + taps.size() * sizeof(float));
    unsigned char* q = blob.data();
    std::memcpy(q, &h, sizeof(h));
    q += sizeof(h);
    std::memcpy(q, lo.data(), (size_t)nw * sizeof(int32_t));
    q += (size_t)nw * sizeof(int32_t);
    std::memcpy(q, hi.data(), (size_t)nw * sizeof(int32_t));
    q += (size_t)nw * sizeof(int32_t);
    std::memcpy(q, taps.data(), taps.size() * sizeof(float));
    return STTFE_OK;
}

extern "C" int sttfe_resample_plan_bytes(int orig_sr, int new_sr, size_t* bytes) {
    if (!bytes) return STTFE_E_INVALID_ARG;
    std::vector<unsigned char> blob;
    int rc = resample_plan(orig_sr, new_sr, blob);
    if (rc != STTFE_OK) return rc;
    *bytes = blob.size();
    return STTFE_OK;
}

extern "C" int sttfe_resample_plan_build(int orig_sr, int new_sr, void* host_blob, size_t bytes) {
    if (!host_blob) return STTFE_E_INVALID_ARG;
    std::vector<unsigned char> blob;
    int rc = resample_plan(orig_sr, new_sr, blob);
    if (rc != STTFE_OK) return rc;
    if (bytes < blob.size()) return STTFE_E_TABLES;
    std::memcpy(host_blob, blob.data(), blob.size());
    return STTFE_OK;
}

// Geometry + float32 taps of the plan (for inspection / tests): taps_out may be NULL.
extern "C" int sttfe_resample_plan_info(const void* host_blob, int* orig_rate, int* new_rate, int* width,
                                        int* n_taps, int* taps_row) {
    if (!host_blob) return STTFE_E_INVALID_ARG;
    const auto* h = reinterpret_cast<const ResamplePlanHeader*>(host_blob);
    if (h->magic != kResampleMagic) return STTFE_E_TABLES;
    if (orig_rate) *orig_rate = h->orig_rate;
    if (new_rate) *new_rate = h->new_rate;
    if (width) *width = h->width;
    if (n_taps) *n_taps = h->n_taps;
    if (taps_row) *taps_row = h->taps_row;
    return STTFE_OK;
}

extern "C" int sttfe_resample_out_length(int orig_sr, int new_sr, int64_t n_in, int64_t* n_out) {
    if (orig_sr < 1 || new_sr < 1 || n_in < 0 || !n_out) return STTFE_E_INVALID_ARG;
    int a = orig_sr, b = new_sr;
    while (b) { const int t = a % b; a = b; b = t; }
    const int64_t orig = orig_sr / a, nw = new_sr / a;
    *n_out = (nw * n_in + orig - 1) / orig;          // ceil(new * n / orig)
    return STTFE_OK;
}
