#include "timing.h"

#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/sttfe.h"

namespace sttfe {

namespace {
struct Slot {
    hipEvent_t start, stop;
    int kernel;
    bool used;
};
std::mutex g_mu;
std::vector<Slot> g_slots;
size_t g_next = 0;
bool g_enabled = false;
// Record only every g_stride-th launch of each kernel: an event pair drains the queue around the launch it brackets
// (rocprofv3 trace of bench.py: a step of the headline workload takes 1.035 ms without events and 1.109 ms with the
// four events of its two kernels), so timing every launch would cost the timed region 7 % of its throughput.
int g_stride = 1;
unsigned g_seen[kTimedKernelCount] = {};
const char* const kNames[kTimedKernelCount] = {"whisper_stft_mel", "whisper_clamp_fill", "kaldi_fbank",
                                               "kaldi_cmvn", "resample_peak", "normalize_trim", "waveform_augment", "pitch_shift",
                                               "whisper_stft_mel12"};
}  // namespace

bool timing_enabled() { return g_enabled; }

int timing_begin(TimedKernel k, hipStream_t stream) {
    if (!g_enabled) return -1;
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_seen[k]++ % (unsigned)g_stride != 0) return -1;
    if (g_next >= g_slots.size()) return -1;
    Slot& s = g_slots[g_next];
    s.kernel = (int)k;
    s.used = true;
    (void)hipEventRecord(s.start, stream);
    return (int)g_next++;
}

void timing_end(int slot, hipStream_t stream) {
    if (slot < 0) return;
    (void)hipEventRecord(g_slots[slot].stop, stream);
}

}  // namespace sttfe

using namespace sttfe;

// Enable (capacity > 0: allocates that many event pairs, clears earlier records) or disable (0).
extern "C" int sttfe_timing_enable(int capacity) {
    std::lock_guard<std::mutex> lock(g_mu);
    for (Slot& s : g_slots) {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    g_slots.clear();
    g_next = 0;
    std::memset(g_seen, 0, sizeof(g_seen));
    g_enabled = capacity > 0;
    for (int i = 0; i < capacity; ++i) {
        Slot s;
        s.used = false;
        s.kernel = -1;
        if (hipEventCreateWithFlags(&s.start, hipEventDisableSystemFence) != hipSuccess ||
            hipEventCreateWithFlags(&s.stop, hipEventDisableSystemFence) != hipSuccess)
            return STTFE_E_LAUNCH;
        g_slots.push_back(s);
    }
    return STTFE_OK;
}

// Forget recorded launches but keep the events.
extern "C" int sttfe_timing_reset(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_next = 0;
    std::memset(g_seen, 0, sizeof(g_seen));
    return STTFE_OK;
}

// Launches of the named kernel since the last enable / reset / set_stride, timed or not.
extern "C" int sttfe_timing_seen(const char* name, int* launches) {
    if (!name || !launches) return STTFE_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_mu);
    for (int i = 0; i < kTimedKernelCount; ++i)
        if (std::strcmp(name, kNames[i]) == 0) {
            *launches = (int)g_seen[i];
            return STTFE_OK;
        }
    return STTFE_E_INVALID_ARG;
}

// Time every stride-th launch of each kernel (1 = every launch, the default).
extern "C" int sttfe_timing_set_stride(int stride) {
    if (stride < 1) return STTFE_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_mu);
    g_stride = stride;
    std::memset(g_seen, 0, sizeof(g_seen));
    return STTFE_OK;
}

// Sum of durations (ms) and number of recorded launches of the named kernel.  Synchronises on the events.
extern "C" int sttfe_timing_query(const char* name, double* total_ms, int* launches) {
    if (!name || !total_ms || !launches) return STTFE_E_INVALID_ARG;
    int k = -1;
    for (int i = 0; i < kTimedKernelCount; ++i)
        if (std::strcmp(name, kNames[i]) == 0) k = i;
    if (k < 0) return STTFE_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(g_mu);
    double total = 0.0;
    int n = 0;
    for (size_t i = 0; i < g_next; ++i) {
        const Slot& s = g_slots[i];
        if (s.kernel != k) continue;
        if (hipEventSynchronize(s.stop) != hipSuccess) return STTFE_E_LAUNCH;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) != hipSuccess) return STTFE_E_LAUNCH;
        total += ms;
        ++n;
    }
    *total_ms = total;
    *launches = n;
    return STTFE_OK;
}
