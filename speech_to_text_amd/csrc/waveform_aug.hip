// Waveform-domain augmentations of the reference's training path, batched and in place on packed PCM.
//
// Reference: AudioAugmentationCPU.apply_time_domain_augmentations (src/data.py:152-183), called per clip
// at src/data.py:328-329 between the preprocessor and the feature extractor.  Per clip, in this order:
//   noise  : audio + torch.randn_like(audio) * noise_factor                          (src/data.py:155-157)
//   speed  : torchaudio SpeedPerturbation = a rational resample -> done by resample.hip, not here (:160-161)
//   pitch  : torchaudio PitchShift -> not built               (:164-167)
//   volume : torchaudio Vol(gain, "amplitude"): clamp(audio * gain, -1, 1)            (:170-174)
//   fade   : torchaudio Fade(n, n, "linear"): fade_in * fade_out * audio with
//            fade_in = cat(linspace(0, 1, n), ones), fade_out = cat(ones, 1 - linspace(0, 1, n))   (:177-181)
// All random draws (gates, gain, the noise itself when bit parity is wanted) are made on the host in the
// reference's order (speech_to_text_amd/augment.py); the kernel only applies the per-clip parameters
//   params[b] = {flags, float bits of noise_factor, float bits of gain, fade samples}.
// `stages` selects which of a clip's flagged stages this call applies, so that the caller can run the noise
// stage before a speed resample and the volume / fade stages after it.
// Arithmetic is float32 with separately rounded products and sums (no FMA contraction), as torch computes it;
// the linspace ramps are start + step * i below the midpoint and end - step * (n - 1 - i) above it (torch's
// scalar formula; its vectorised CPU path may differ from that in the last bit).
// One pass over the data: 4 n bytes read + 4 n written (+ 4 n noise), 8 samples per thread and trip, all
// loads issued before the first store.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sttfe.h"
#include "timing.h"

// torch rounds every product and sum separately; hipcc's default (-ffp-contract=fast-honor-pragmas) would fuse
// x + n * sigma into one FMA -- also through __fmul_rn / __fadd_rn, which are plain * and + in the HIP headers and
// are not covered by a pragma placed here; hence plain operators below, under this pragma.
#pragma clang fp contract(off)

namespace sttfe {

__global__ __launch_bounds__(256) void waveform_augment_kernel(
    float* __restrict__ pcm, const int64_t* __restrict__ offsets, const int32_t* __restrict__ lengths,
    const int32_t* __restrict__ params, const float* __restrict__ noise,
    const int64_t* __restrict__ noise_offsets, int stages) {
    const int clip = blockIdx.y;
    const int n = lengths[clip];
    const int flags = params[4 * clip] & stages;
    if (n <= 0 || flags == 0) return;
    const float sigma = __int_as_float(params[4 * clip + 1]);
    const float gain = __int_as_float(params[4 * clip + 2]);
    const int fade_n = params[4 * clip + 3];
    const bool do_noise = (flags & STTFE_AUG_NOISE) && noise != nullptr;
    const bool do_gain = flags & STTFE_AUG_GAIN;
    const bool do_fade = (flags & STTFE_AUG_FADE) && fade_n > 0 && fade_n <= n;
    if (!do_noise && !do_gain && !do_fade) return;
    float* __restrict__ x = pcm + offsets[clip];
    const float* __restrict__ z = do_noise ? noise + noise_offsets[clip] : nullptr;
    const float step = fade_n > 1 ? __fdiv_rn(1.0f, (float)(fade_n - 1)) : 0.0f;     // linspace step, float32
    const int half = fade_n / 2;
    auto ramp = [&](int i) {                    // torch.linspace(0, 1, fade_n)[i]
        if (fade_n == 1) return 0.0f;
        return i < half ? step * (float)i : 1.0f - step * (float)(fade_n - 1 - i);
    };
    const int span = gridDim.x * blockDim.x * 8;
    for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8; i0 < n; i0 += span) {
        float v[8], w[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            v[c] = (i0 + c < n) ? x[i0 + c] : 0.f;
            w[c] = (do_noise && i0 + c < n) ? z[i0 + c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int i = i0 + c;
            float y = v[c];
            if (do_noise) y = y + w[c] * sigma;
            if (do_gain) y = fminf(fmaxf(y * gain, -1.0f), 1.0f);
            if (do_fade) {
                const float fin = i < fade_n ? ramp(i) : 1.0f;
                const int k = i - (n - fade_n);
                const float fout = k >= 0 ? 1.0f - ramp(k) : 1.0f;
                y = (fin * fout) * y;
            }
            v[c] = y;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (i0 + c < n) x[i0 + c] = v[c];
    }
}

}  // namespace sttfe

using namespace sttfe;

extern "C" int sttfe_waveform_augment(float* pcm, const int64_t* offsets, const int32_t* lengths, int batch,
                                      int64_t max_length, const int32_t* params, const float* noise,
                                      const int64_t* noise_offsets, int stages, void* stream_) {
    if (batch < 0 || max_length < 0) return STTFE_E_INVALID_ARG;
    if (batch == 0 || max_length == 0 || (stages & (STTFE_AUG_NOISE | STTFE_AUG_GAIN | STTFE_AUG_FADE)) == 0)
        return STTFE_OK;
    if (!pcm || !offsets || !lengths || !params) return STTFE_E_INVALID_ARG;
    if ((noise == nullptr) != (noise_offsets == nullptr)) return STTFE_E_INVALID_ARG;
    if (batch > 65535) return STTFE_E_UNSUPPORTED;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    (void)hipGetLastError();
    {
        ScopedTiming t(kTimeWaveformAug, stream);
        int bx = (int)((max_length + 256 * 8 - 1) / (256 * 8));      // one 8-sample trip per thread
        bx = bx < 1 ? 1 : (bx > 4096 ? 4096 : bx);
        hipLaunchKernelGGL(waveform_augment_kernel, dim3(bx, batch), dim3(256), 0, stream, pcm, offsets, lengths,
                           params, noise, noise_offsets, stages);
    }
    if (hip_failed(hipGetLastError(), "waveform augment kernel launch")) return STTFE_E_LAUNCH;
    return STTFE_OK;
}
