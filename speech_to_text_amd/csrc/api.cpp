// Version / error-string entry points of libsttfe.so (include/sttfe.h).
#include <hip/hip_runtime.h>

#include "../../include/sttfe.h"

extern "C" int sttfe_version(void) { return 100; }   // 0.1.0

extern "C" const char* sttfe_strerror(int code) {
    switch (code) {
        case STTFE_OK: return "ok";
        case STTFE_E_INVALID_ARG: return "invalid argument";
        case STTFE_E_WORKSPACE: return "workspace too small";
        case STTFE_E_TABLES: return "tables blob does not match the call";
        case STTFE_E_LAUNCH: return "HIP launch error";
        case STTFE_E_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown sttfe error";
    }
}

// Detail of the last STTFE_E_LAUNCH (the HIP error string), per thread.
namespace sttfe { thread_local char g_last_detail[256] = ""; }
extern "C" const char* sttfe_last_error_detail(void) { return sttfe::g_last_detail; }

// Device address of page-locked host memory (a DataLoader batch with pin_memory=True, src/config.py:77,
// train.py:1457-1464), or STTFE_E_UNSUPPORTED when the pointer is not mapped into the device's address space.
extern "C" int sttfe_host_mapped_pointer(const void* host_ptr, void** device_ptr) {
    if (!host_ptr || !device_ptr) return STTFE_E_INVALID_ARG;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(host_ptr), 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        return STTFE_E_UNSUPPORTED;
    }
    *device_ptr = d;
    return STTFE_OK;
}
