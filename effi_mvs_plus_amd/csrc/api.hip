// Library-level entry points of the C ABI (version, error strings).
#include "common.hpp"

extern "C" int effi_version(void) { return 200; }   // 0.2.0

extern "C" const char* effi_error_string(int code) {
    switch (code) {
        case EFFI_OK: return "ok";
        case EFFI_ERR_BADARG: return "bad argument (null pointer, non-positive size, or inconsistent shapes)";
        case EFFI_ERR_UNSUPPORTED: return "shape / channel count not instantiated in this build";
        case EFFI_ERR_LAUNCH: return "kernel launch failed (hipPeekAtLastError)";
        case EFFI_ERR_WORKSPACE: return "no workspace registered for the current device (effi_set_workspace)";
        default: return "unknown error code";
    }
}

// Per-device zero page, owned by the caller (effi_set_workspace): one slot per device ordinal.
static const void* g_workspace[EFFI_MAX_DEVICES];

extern "C" long effi_workspace_bytes(void) { return 256; }

extern "C" int effi_set_workspace(int device, void* workspace, long bytes) {
    if (device < 0 || device >= EFFI_MAX_DEVICES) return EFFI_ERR_BADARG;
    if (workspace != nullptr && bytes < effi_workspace_bytes()) return EFFI_ERR_BADARG;
    __atomic_store_n(&g_workspace[device], (const void*)workspace, __ATOMIC_RELEASE);
    return EFFI_OK;
}

extern "C" const void* effi_get_workspace(int device) {
    if (device < 0 || device >= EFFI_MAX_DEVICES) return nullptr;
    return __atomic_load_n(&g_workspace[device], __ATOMIC_ACQUIRE);
}

// zero page of the current device, or nullptr
const float* effi_zero_page() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    return (const float*)effi_get_workspace(dev);
}
