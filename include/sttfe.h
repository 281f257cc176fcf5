/*
 * sttfe.h -- C ABI of libsttfe.so, the MI355X (gfx950) audio front-end.
 *
 * Drop-in boundary for the feature-extraction path of Alexey511/speech_to_text.  The reference is
 * pure Python, so there is no existing FFI to match; each entry point below replaces the Python
 * call cited next to it (paths relative to the reference repo; "HF:" = the transformers package
 * the reference pins in requirements.txt:11).  A maintainer binds these with ctypes -- see
 * INTEGRATION.md for the stub.
 *
 * Conventions
 *   - every function returns 0 (STTFE_OK) or a negative STTFE_E_* code; sttfe_strerror() names it;
 *   - "device" pointers are HIP device memory owned by the caller (e.g. torch tensors); the
 *     library never allocates, frees or synchronises; all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - "host" pointers are ordinary host memory;
 *   - tables: constant data a kernel needs (window, twiddles, banded mel weights) is produced by a
 *     *_tables_build() call into a HOST buffer; the caller copies that blob to the device once and
 *     passes the device copy to the kernels;
 *   - PCM is float32, one clip after another in one buffer; clip b starts at pcm[offsets[b]] and
 *     has lengths[b] valid samples (offsets / lengths are DEVICE arrays: no host sync needed when
 *     an earlier kernel produced them).
 */
#ifndef STTFE_H
#define STTFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STTFE_OK 0
#define STTFE_E_INVALID_ARG (-1)   /* NULL pointer, negative size, unsupported n_mels ... */
#define STTFE_E_WORKSPACE (-2)     /* workspace too small */
#define STTFE_E_TABLES (-3)        /* tables blob does not match the call (host-side checks: *_tables_build with a
                                      short buffer, resample plans).  A DEVICE tables blob built for another extractor
                                      (wrong kind or n_mels) cannot be reported without synchronising: the feature
                                      kernels check its header themselves and make every output element NaN. */
#define STTFE_E_LAUNCH (-4)        /* HIP reported a launch error */
#define STTFE_E_UNSUPPORTED (-5)   /* sampling-rate ratio / size outside what the kernels handle */

#define STTFE_AXIS_FREQ 0
#define STTFE_AXIS_TIME 1
#define STTFE_MAX_MASKS 16         /* SpecAugment masks per clip (num_time_masks + num_freq_masks, src/config.py:43-44);
                                      more than that: STTFE_E_INVALID_ARG */

int sttfe_version(void);
const char* sttfe_strerror(int code);
/* HIP error text behind the calling thread's last STTFE_E_LAUNCH ("" if none). */
const char* sttfe_last_error_detail(void);

/* ---------------------------------------------------------------- Whisper log-mel (row W) ----
 * Replaces processor.feature_extractor(audio_np, sampling_rate=16000, return_tensors="pt")
 * for WhisperProcessor: src/data.py:338-343, inference.py:148-152, and the batched
 * TorchWhisperFeatureExtractor.__call__ of src/processors.py:148-246.  Arithmetic =
 * HF:models/whisper/feature_extraction_whisper.py:135-168,300-307.
 */

/* Size of / fill the host tables blob for an n_mels-bin extractor (n_mels in [1,128]). */
int sttfe_whisper_tables_bytes(int n_mels, size_t* bytes);
int sttfe_whisper_tables_build(int n_mels, void* host_blob, size_t bytes);
/* The dense (n_mels x 201) float32 mel bank the tables were derived from (for inspection/tests). */
int sttfe_whisper_mel_bank(int n_mels, float* host_dense);

/* Bytes of device workspace needed for a batch (per-clip max key + per-tile min keys + one work-item ticket
 * counter; the entry point zeroes it itself, contents need not survive between calls). */
int sttfe_whisper_workspace_bytes(int batch, size_t* bytes);

/*
 * pcm/offsets/lengths : device; clips at 16 kHz; samples past lengths[b] (up to 480000) count as 0,
 *                       samples past 480000 are ignored (HF pad/truncate to 30 s).
 * out                 : device float32 [batch][n_mels][n_frames], n_frames = 3000 for HF parity.
 *                       Compact output (callers that keep "frames + padding constant", e.g. a feature cache): any
 *                       multiple of 4 with 160 * n_frames >= lengths[b] + 200 + 4 * 160 for every clip gives columns
 *                       bit-identical to the first n_frames of the 3000-frame result, the last 4 of them padding for
 *                       every clip (out[b][0][n_frames - 1] is clip b's padding constant); the 30 s padding behind them
 *                       is then neither computed nor written.  The caller guarantees the bound; it is not checked
 *                       here (the lengths are device data).  With a longer clip the columns written are still those of
 *                       the 3000-frame result -- the kernel keeps reading the clip's samples and reflects at 480000,
 *                       not at 160 * n_frames -- but the last columns are then data, not padding, and
 *                       out[b][0][n_frames - 1] is not the padding constant (host callers that know the lengths check:
 *                       HipWhisperFeatureExtractor.extract(host_lengths=), FeatureCacheWriter.add_batch).
 * mask_tbl            : device int32 [batch][n_masks][3] = {axis, start, end} or NULL; elements
 *                       with start <= index < end along the axis are set to 0.0 AFTER the
 *                       log/clamp/scale (SpecAugment, src/data.py:185-194); rows with end<=start
 *                       are no-ops.
 */
int sttfe_whisper_logmel(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                         int batch, int n_mels, int n_frames, const void* dev_tables,
                         float* out, const int32_t* mask_tbl, int n_masks,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same features for clip_scale[b] * clip b (clip_scale: device float32 [batch], > 0), without touching the
 * samples: rows P2 + W fused.  Replaces the pair x / (max|x| + 1e-8) (src/data.py:78-81) -> feature extractor
 * (src/data.py:338-343) when the normalised waveform itself is not needed: with s = 1 / (peak + 1e-8) from
 * sttfe_resample_peak's workspace, log10(max(s^2 mel, 1e-10)) = log10(max(mel, 1e-10 / s^2)) + 2 log10(s) is a
 * per-clip floor and bias in the kernel's epilogue.  Differs from normalising first by the float32 rounding of
 * s * x per sample (about 1e-7 in the normalised log-mel). */
int sttfe_whisper_logmel_scaled(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                int batch, int n_mels, int n_frames, const void* dev_tables,
                                float* out, const int32_t* mask_tbl, int n_masks, const float* clip_scale,
                                void* workspace, size_t workspace_bytes, void* stream);

/* The same extractor with the semantics of the reference's own TorchWhisperFeatureExtractor (src/processors.py:148-246;
 * HF mel bank), which differ from HF's in which frames exist and which count for the clamp:
 *   variable_length = 0 (padding=True, src/processors.py:180-221): pad / truncate to 30 s, per-clip max over all 3001
 *       STFT frames, THEN crop to 3000 (HF drops the last frame before the max); n_frames must be 3000;
 *   variable_length = 1 (padding=False, :222-239): no padding -- reflect at the clip's own end, 1 + lengths[b] / 160
 *       frames, per-clip max over them, cropped to 3000; out is [batch][n_mels][n_frames] with n_frames >= the longest
 *       clip's (cropped) frame count and a multiple of 4; frames past a clip's own are padding_value.  Clips need more
 *       than 200 samples (torch.stft's reflect padding raises below that; here such clips give unspecified values).
 *       Clips with more than n_frames frames are transformed to their own end: the frames behind the output count for
 *       the maximum and are not stored -- the reference crops after the maximum (src/processors.py:226-239).  (Calls
 *       with more than 16 384 clips look at most 3024 frames far.) */
int sttfe_whisper_logmel_torch(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                               int batch, int n_mels, int n_frames, const void* dev_tables,
                               float* out, const int32_t* mask_tbl, int n_masks, int variable_length,
                               float padding_value, void* workspace, size_t workspace_bytes, void* stream);

/* sttfe_whisper_logmel for callers that cannot allocate.  No counterpart in the reference: it is for code under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc, which cannot size and provide the
 * `workspace` argument above.  Same inputs, outputs and results as sttfe_whisper_logmel; the workspace (per-clip max
 * key, per-tile min keys, work-item ticket counter) is a device array inside the library:
 * 4 * (64 * STTFE_WHISPER_STATIC_MAX_BATCH + 16) bytes (1 MiB + 64 B) on every GPU the library runs on, held until the
 * process ends.  batch > STTFE_WHISPER_STATIC_MAX_BATCH returns STTFE_E_UNSUPPORTED (sttfe_last_error_detail() says
 * why).  The call allocates and frees nothing (the array's device address is looked up once per device, on the first
 * call made with that device current: hipGetSymbolAddress, which may load the code object and is not legal under
 * stream capture -- make ONE ordinary warm-up call per device (any batch >= 1) before the first captured or
 * allocation-forbidden one; every later call on that device only launches).  All calls on one GPU share that GPU's array: the entry point clears the part
 * it uses on `stream` before its kernels run, and two calls that overlap on one GPU (different streams, or different
 * host threads on different streams) race -- serialise them, or use sttfe_whisper_logmel with separate workspaces. */
#define STTFE_WHISPER_STATIC_MAX_BATCH 4096
int sttfe_whisper_logmel_static(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                                int batch, int n_mels, int n_frames, const void* dev_tables,
                                float* out, const int32_t* mask_tbl, int n_masks, void* stream);

/* ------------------------------------------- Speech2Text Kaldi fbank + utterance CMVN (row K) ----
 * Replaces processor.feature_extractor(audio_np, sampling_rate=16000, return_tensors="pt",
 * truncation=..., max_length=...) for Speech2TextProcessor: src/data.py:346-358, plus the time
 * padding / attention mask of the collator, src/data.py:468-487.  Arithmetic =
 * HF:models/speech_to_text/feature_extraction_speech_to_text.py:104-163 and HF:audio_utils.py:953-1017.
 */
#define STTFE_KALDI_NO_CMVN 1      /* flags: store ln(mel) without the (x - mean) / std step */
#define STTFE_KALDI_F64 2          /* flags: DC removal, pre-emphasis, window and the 512-point DFT in float64, spectra
                                      rounded to complex64 afterwards -- the arithmetic of HF's NumPy path
                                      (HF:audio_utils.py:957-993), max-abs <= 1e-4 against it on every cell.  Without the
                                      flag those stages run in float32, as torchaudio's Kaldi fbank does
                                      (HF:...speech_to_text.py:113-121): cells 55+ dB below their frame's peak bin then
                                      carry the float32 STFT's noise floor. */

#define STTFE_KALDI_NO_MEAN 4      /* flags: utterance CMVN without the mean step (normalize_means=False): x / std(x) */
#define STTFE_KALDI_NO_VAR 8       /* flags: ... without the variance step (normalize_vars=False): x - mean(x)
                                      (HF:...speech_to_text.py:141-163; both set = STTFE_KALDI_NO_CMVN) */

int sttfe_kaldi_tables_bytes(int n_mels, size_t* bytes);
int sttfe_kaldi_tables_build(int n_mels, void* host_blob, size_t bytes);
int sttfe_kaldi_mel_bank(int n_mels, float* host_dense /* [n_mels][257] */);
/* workspace = per-clip per-bin float64 sums for CMVN + the kernel's work-item ticket counter; the entry
 * point zeroes it itself on `stream`. */
int sttfe_kaldi_workspace_bytes(int batch, int n_mels, size_t* bytes);

/*
 * Frames of clip b: T_b = min(1 + (lengths[b] - 400) / 160, max_frames if > 0, t_max); 0 if shorter
 * than 400 samples.
 * out        : device float32 [batch][t_max][n_mels] (time-major, what Speech2Text consumes);
 *              rows t >= T_b are 0.0 (padding_value).
 * attn_mask  : device [batch][t_max] of int32 (attn_mask_bytes = 4, the HF extractor's dtype) or
 *              int64 (8, the collator's torch.long, src/data.py:484); NULL to skip.
 * mask_tbl   : as for sttfe_whisper_logmel; axis STTFE_AXIS_FREQ indexes the mel bin,
 *              STTFE_AXIS_TIME the frame; applied after CMVN.
 */
int sttfe_kaldi_fbank_cmvn(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                           int batch, int n_mels, int t_max, int max_frames, int flags,
                           const void* dev_tables, float* out, void* attn_mask, int attn_mask_bytes,
                           const int32_t* mask_tbl, int n_masks,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------ mono mix, resample, peak normalise, silence trim (rows P1 R P2 P3) ----
 * Replaces AudioPreprocessor.preprocess, src/data.py:54-108 (same steps in inference.py:100-132):
 * torch.mean over channels (:66-67), torchaudio.transforms.Resample(orig, 16000) (:70-72),
 * x / (max|x| + 1e-8) (:78-81), trim to the first..last sample with |x| > threshold * max|x| (:84-91).
 * The duration gate (:95-98) stays on the host: it only needs the lengths array.
 *
 * Plan = the polyphase windowed-sinc kernel of torchaudio's Resample defaults (sinc_interp_hann,
 * lowpass_filter_width 6, rolloff 0.99) for one (orig_sr, new_sr) pair, built on the HOST; the caller
 * copies the blob to the device once and passes BOTH copies (the library reads the header from the
 * host copy, the kernels read the device copy).  Rates are reduced by their gcd.  Reduced rates above 1024 (a speed
 * factor such as 0.9667: 7733 : 8000, whose table would be 248 MB; torchaudio accepts any, src/data.py:144-150) get a
 * plan WITHOUT a table -- the 32-byte header alone, taps_row == 0 -- and run through a kernel that evaluates the
 * non-zero taps of every output sample in float64 as the table builder does (several times slower per sample than the
 * tabled kernels); reduced rates above 2^22 return STTFE_E_UNSUPPORTED.
 */
int sttfe_resample_plan_bytes(int orig_sr, int new_sr, size_t* bytes);
int sttfe_resample_plan_build(int orig_sr, int new_sr, void* host_blob, size_t bytes);
int sttfe_resample_plan_info(const void* host_blob, int* orig_rate, int* new_rate, int* width,
                             int* n_taps, int* taps_row);
/* Zero-copy ingest: the device address of PAGE-LOCKED host memory (what a DataLoader with pin_memory=True hands over,
 * reference src/config.py:77 / train.py:1457-1464), for use as `pcm_in` of sttfe_resample_peak / _s16: the kernel then
 * pulls the clips' VALID samples over PCIe itself -- no staging copy of the padded (batch, n_max) tensor, padding never
 * crosses the bus.  STTFE_E_UNSUPPORTED when the pointer is not mapped for the current device (pageable memory): copy
 * it yourself then.  The caller keeps the host memory alive and unmodified until the work queued on `stream` is done. */
int sttfe_host_mapped_pointer(const void* host_ptr, void** device_ptr);
/* ceil(new * n_in / orig): the output length torchaudio crops to. */
int sttfe_resample_out_length(int orig_sr, int new_sr, int64_t n_in, int64_t* n_out);

/* Device workspace shared by the two calls below: per-clip peak keys + 64 + 64 candidate slots per clip for the trim
 * bounds (516 bytes per clip). */
int sttfe_preprocess_workspace_bytes(int batch, size_t* bytes);

/*
 * pcm_in       : device float32; clip b = `channels` planar rows of in_lengths[b] samples starting at
 *                pcm_in[in_offsets[b]] (channels > 1 are averaged to mono).
 * max_in_length: host upper bound of in_lengths[] (sizes the grid).
 * dev_plan / host_plan : both NULL = no rate change (mono mix / copy / peak only; pcm_out may alias
 *                pcm_in when channels == 1).
 * pcm_out      : device float32; clip b is written at pcm_out[out_offsets[b]], out_lengths[b] =
 *                ceil(new * in_lengths[b] / orig) samples (out_offsets is chosen by the caller with
 *                sttfe_resample_out_length; 16-byte aligned starts get the vector-store path).
 * workspace    : receives max|out| of every clip (consumed by sttfe_normalize_trim).
 */
int sttfe_resample_peak(const float* pcm_in, const int64_t* in_offsets, const int32_t* in_lengths,
                        int batch, int channels, int64_t max_in_length,
                        const void* dev_plan, const void* host_plan,
                        float* pcm_out, const int64_t* out_offsets, int32_t* out_lengths,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The same call for int16 PCM as decoders deliver it (sample / 32768, the normalisation torchaudio.load applies at
 * reference src/data.py:313; exact in float32): pcm_in holds int16 samples, offsets and lengths count samples.
 * Half the bytes over PCIe and out of HBM, no conversion pass (decoded-PCM ingest).
 * pcm_out may not alias pcm_in. */
int sttfe_resample_peak_s16(const int16_t* pcm_in, const int64_t* in_offsets, const int32_t* in_lengths,
                            int batch, int channels, int64_t max_in_length,
                            const void* dev_plan, const void* host_plan,
                            float* pcm_out, const int64_t* out_offsets, int32_t* out_lengths,
                            void* workspace, size_t workspace_bytes, void* stream);

/*
 * In place on the output of sttfe_resample_peak (same workspace, same stream):
 *   normalize : pcm <- pcm / (peak + 1e-8) for clips with peak > 0 (IEEE float32 division);
 *   trim      : offsets[b] += first, lengths[b] = last + 1 - first, where first / last are the first and
 *               last sample with |x| > silence_threshold * max|x| (after normalisation when enabled);
 *               clips with nothing above the threshold are left unchanged.  No samples move.
 * max_length : host upper bound of lengths[].
 */
int sttfe_normalize_trim(float* pcm, int64_t* offsets, int32_t* lengths, int batch, int64_t max_length,
                         int normalize, int trim, float silence_threshold,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------- waveform augmentations --------
 * Replaces: AudioAugmentationCPU.apply_time_domain_augmentations (reference src/data.py:152-183, called per clip at
 * src/data.py:328-329): Gaussian noise (:155-157), torchaudio Vol "amplitude" gain with its clamp to [-1, 1]
 * (:170-174) and torchaudio Fade, linear, same length in and out (:177-181), applied in place to packed PCM in the
 * reference's order.  Speed perturbation (:160-161) is a rational resample: sttfe_resample_peak with a
 * (round(factor * rate) : rate) plan; pitch shift (:164-167) is sttfe_pitch_shift below.
 *   params  : device int32 [batch][4] = {STTFE_AUG_* flags of the clip, float bits of noise_factor,
 *             float bits of the amplitude gain, fade samples}; the host draws the gates and values in the
 *             reference's RNG order (speech_to_text_amd/augment.py).
 *   noise   : device float32 standard-normal samples, clip b at noise[noise_offsets[b] ...] (lengths[b] of them),
 *             or NULL (then no noise is added); torch.randn drawn on the host reproduces the reference bit for bit.
 *   stages  : which flagged stages this call applies (noise before a speed resample, gain and fade after it).
 *   max_length : longest clip of the batch (grid sizing). */
#define STTFE_AUG_NOISE 1
#define STTFE_AUG_GAIN 2
#define STTFE_AUG_FADE 4
int sttfe_waveform_augment(float* pcm, const int64_t* offsets, const int32_t* lengths, int batch, int64_t max_length,
                           const int32_t* params, const float* noise, const int64_t* noise_offsets, int stages,
                           void* stream);

/* ------------------------------------------------------------- pitch shift -------------------
 * Replaces: self.pitch_shift(audio) of AudioAugmentationCPU (reference src/data.py:163-167; the transform is built at
 * src/data.py:133-136 as torchaudio.transforms.PitchShift(sample_rate, n_steps=0) and its n_steps attribute is set per
 * clip to random.randint(-2, 2)).  torchaudio 2.5.1 arithmetic: stft (512 / hop 128 / periodic hann / reflect) ->
 * phase vocoder at rate 2^(-n_steps/12) -> istft to round(n / rate) samples -> sinc resample int(sr / rate) -> sr ->
 * crop or zero-pad to n samples; in place on packed mono PCM, clip lengths unchanged.
 *   mode STTFE_PITCH_AS_CONSTRUCTED : what the reference's transform object computes -- it fixes the resample ratio in
 *        its constructor (n_steps = 0 -> 1:1), so the resample is skipped: a phase-vocoder tempo change cropped / padded
 *        back to the clip's length;
 *   mode STTFE_PITCH_FUNCTIONAL     : torchaudio.functional.pitch_shift with the clip's n_steps (resample included).
 * sttfe_pitch_shift_plan (host only): from the HOST copy of the clip lengths and n_steps[b] (STTFE_PITCH_NONE = leave
 * the clip alone; -48..48 otherwise) fills layout[batch][8] (int64, opaque) and the workspace size.  Clips of <= 256
 * samples cannot be shifted (torch.stft's reflect padding raises there): STTFE_E_INVALID_ARG.
 * sttfe_pitch_shift: `layout` is that HOST array (every extent is re-derived from it and checked against
 * workspace_bytes before anything is launched; it is not read after the call returns), `dev_layout` the caller's
 * DEVICE copy of the same bytes, which the kernels read; `lengths` are the device lengths -- a clip whose device
 * length differs from the planned one is left untouched.  workspace must be 16-byte aligned.  About 6 MB of
 * workspace per 10 s clip.  Like every entry point: no allocation, no copy, no synchronisation inside. */
#define STTFE_PITCH_NONE (-2147483647 - 1)
#define STTFE_PITCH_AS_CONSTRUCTED 0
#define STTFE_PITCH_FUNCTIONAL 1
int sttfe_pitch_shift_plan(const int32_t* lengths, const int32_t* n_steps, int batch, int sample_rate, int mode,
                           int64_t* layout, size_t* workspace_bytes);
int sttfe_pitch_shift(float* pcm, const int64_t* offsets, const int32_t* lengths, int batch, const int64_t* layout,
                      const int64_t* dev_layout, int mode, void* workspace, size_t workspace_bytes, void* stream);

/* Consecutive 48-frame tiles per work item of the STFT kernels (0 = automatic, the default: the kernel derives it
 * from the clip lengths it finds on the device -- general kernel: a third of the clip for uniform batches, about eight
 * items per workgroup otherwise, within [3, 21]; 12-wave kernel: about eight (uniform) or sixteen (ragged) items per
 * workgroup, within [1, 21]).  Tuning knob, no counterpart in the reference. */
int sttfe_set_whisper_tiles_per_block(int tiles);

/* Calls of the 80- and 128-bin extractors in HF mode (sttfe_whisper_logmel, _scaled, _static; with or without masks) are
 * computed by a second STFT kernel (one 12-wave workgroup per CU with transform and mel waves,
 * csrc/whisper_logmel12.inc; bit-identical features).  on = 1 (default): batches of up to 3072 clips take that kernel
 * alone, uniform or ragged; larger batches launch both STFT kernels and the lengths on the device decide which of them
 * works (the 12-wave kernel if every clip ends within a 48-frame tile of the longest one).  on = 0: every call takes the
 * general kernel.  Also env STTFE_WH_UNIFORM_KERNEL=0/1 (the API call wins); env STTFE_WH12_RAGGED=0 keeps ragged
 * batches of any size on the general kernel.  Tuning / A-B knobs, no counterpart in the reference. */
int sttfe_set_whisper_uniform_kernel(int on);

/* ------------------------------------------------------------- measurement support -----------
 * No counterpart in the reference.  Optional per-kernel timing with HIP events recorded on the
 * launch stream around every kernel launch (used by bench.py for the roofline figure).
 * sttfe_timing_enable(capacity) allocates `capacity` event pairs (0 disables and frees them);
 * sttfe_timing_query(name, ...) returns the summed duration and launch count of one kernel
 * ("whisper_stft_mel", "whisper_clamp_fill", "kaldi_fbank", "kaldi_cmvn", "resample_peak",
 * "normalize_trim", "waveform_augment", "pitch_shift", "whisper_stft_mel12" -- the 12-wave STFT kernel of uniform
 * batches, launched beside "whisper_stft_mel" by the calls that qualify for it: the one the batch is not for returns at
 * once) since the last enable/reset; it synchronises on the recorded events.
 */
int sttfe_timing_enable(int capacity);
int sttfe_timing_reset(void);
/* Time only every stride-th launch of each kernel (default 1).  An event pair drains the queue around the launch it
 * brackets (about 35 us per pair on MI355X), so a benchmark that also reports throughput over the same region
 * samples the launches instead of timing all of them. */
int sttfe_timing_set_stride(int stride);
/* Launches of the named kernel since the last enable / reset / set_stride, whether timed or skipped by the stride. */
int sttfe_timing_seen(const char* name, int* launches);
int sttfe_timing_query(const char* name, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* STTFE_H */
