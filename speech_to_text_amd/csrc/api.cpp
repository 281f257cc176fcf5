// Version / error-string entry points of libsttfe.so (include/sttfe.h).
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
