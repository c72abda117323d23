// abi.hip -- version / error reporting entry points of liblhvi.so.
#include "common.hpp"

namespace lhvi { thread_local int g_last_hip_error = 0; }

extern "C" {

int lhvi_version(void) { return LHVI_ABI_VERSION; }

const char* lhvi_strerror(int code) {
    switch (code) {
        case LHVI_OK: return "ok";
        case LHVI_E_ARG: return "invalid argument";
        case LHVI_E_LAUNCH: return "HIP launch failed";
        case LHVI_E_UNSUPPORTED: return "unsupported configuration";
        case LHVI_E_NODEVICE: return "no HIP device";
        default: return "unknown error";
    }
}

int lhvi_last_hip_error(void) { return lhvi::g_last_hip_error; }

int lhvi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"
