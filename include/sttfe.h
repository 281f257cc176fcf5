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
#define STTFE_E_TABLES (-3)        /* tables blob does not match the call */
#define STTFE_E_LAUNCH (-4)        /* HIP reported a launch error */
#define STTFE_E_UNSUPPORTED (-5)   /* sampling-rate ratio / size outside what the kernels handle */

#define STTFE_AXIS_FREQ 0
#define STTFE_AXIS_TIME 1

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

/* Bytes of device workspace needed for a batch (per-clip max keys). */
int sttfe_whisper_workspace_bytes(int batch, size_t* bytes);

/*
 * pcm/offsets/lengths : device; clips at 16 kHz; samples past lengths[b] (up to 480000) count as 0,
 *                       samples past 480000 are ignored (HF pad/truncate to 30 s).
 * out                 : device float32 [batch][n_mels][n_frames], n_frames = 3000 for HF parity.
 * mask_tbl            : device int32 [batch][n_masks][3] = {axis, start, end} or NULL; elements
 *                       with start <= index < end along the axis are set to 0.0 AFTER the
 *                       log/clamp/scale (SpecAugment, src/data.py:185-194); rows with end<=start
 *                       are no-ops.
 */
int sttfe_whisper_logmel(const float* pcm, const int64_t* offsets, const int32_t* lengths,
                         int batch, int n_mels, int n_frames, const void* dev_tables,
                         float* out, const int32_t* mask_tbl, int n_masks,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------- Speech2Text Kaldi fbank + utterance CMVN (row K) ----
 * Replaces processor.feature_extractor(audio_np, sampling_rate=16000, return_tensors="pt",
 * truncation=..., max_length=...) for Speech2TextProcessor: src/data.py:346-358, plus the time
 * padding / attention mask of the collator, src/data.py:468-487.  Arithmetic =
 * HF:models/speech_to_text/feature_extraction_speech_to_text.py:104-163 and HF:audio_utils.py:953-1017.
 */
#define STTFE_KALDI_NO_CMVN 1      /* flags: store ln(mel) without the (x - mean) / std step */

int sttfe_kaldi_tables_bytes(int n_mels, size_t* bytes);
int sttfe_kaldi_tables_build(int n_mels, void* host_blob, size_t bytes);
int sttfe_kaldi_mel_bank(int n_mels, float* host_dense /* [n_mels][257] */);
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

/* Clips per launch pair of sttfe_whisper_logmel (default 96): kernel 1's output for a chunk should
 * still be in the 256 MiB Infinity Cache when the clamp kernel re-reads it.  Tuning knob. */
int sttfe_set_whisper_chunk(int clips);

/* ------------------------------------------------------------- measurement support -----------
 * No counterpart in the reference.  Optional per-kernel timing with HIP events recorded on the
 * launch stream around every kernel launch (used by bench.py for the roofline figure).
 * sttfe_timing_enable(capacity) allocates `capacity` event pairs (0 disables and frees them);
 * sttfe_timing_query(name, ...) returns the summed duration and launch count of one kernel
 * ("whisper_stft_mel", "whisper_clamp_mask", "kaldi_fbank", "kaldi_cmvn", "resample_peak",
 * "normalize_trim") since the last enable/reset; it synchronises on the recorded events.
 */
int sttfe_timing_enable(int capacity);
int sttfe_timing_reset(void);
int sttfe_timing_query(const char* name, double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* STTFE_H */
