// Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
// roofline figure).  Disabled by default: the hot path then records nothing.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>

namespace sttfe {

extern thread_local char g_last_detail[256];
// Records `what: <hip error string>` and returns true when e is an error.
inline bool hip_failed(hipError_t e, const char* what) {
    if (e == hipSuccess) return false;
    std::snprintf(g_last_detail, sizeof(g_last_detail), "%s: %s", what, hipGetErrorString(e));
    return true;
}

enum TimedKernel { kTimeWhisperStftMel = 0, kTimeWhisperClamp, kTimeKaldiFbank, kTimeKaldiCmvn,
                   kTimeResample, kTimeNormTrim, kTimeWaveformAug, kTimePitchShift, kTimeWhisperStftMel12, kTimedKernelCount };

bool timing_enabled();
// Record a start/stop event pair around one launch; returns a slot (or -1 when disabled / full).
int timing_begin(TimedKernel k, hipStream_t stream);
void timing_end(int slot, hipStream_t stream);

struct ScopedTiming {
    int slot;
    hipStream_t stream;
    ScopedTiming(TimedKernel k, hipStream_t s) : slot(timing_begin(k, s)), stream(s) {}
    ~ScopedTiming() { timing_end(slot, stream); }
};

}  // namespace sttfe
