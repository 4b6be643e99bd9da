// Library-level entry points of the C ABI (version, error strings).
#include "common.hpp"

extern "C" int effi_version(void) { return 100; }   // 0.1.0

extern "C" const char* effi_error_string(int code) {
    switch (code) {
        case EFFI_OK: return "ok";
        case EFFI_ERR_BADARG: return "bad argument (null pointer, non-positive size, or inconsistent shapes)";
        case EFFI_ERR_UNSUPPORTED: return "shape / channel count not instantiated in this build";
        case EFFI_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
        default: return "unknown error code";
    }
}
