// Library-level entry points of the C ABI (version, error strings, workspace registry, tuning options).
#include "common.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>

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

// ---- tuning / A-B options (common.hpp: EffiOption) ------------------------------------------------------------------------
namespace {
const char* const kOptName[EFFI_OPT_COUNT] = {"warp_lds_kb", "dyn_form", "dyn_setup_exact", "dyn_xchg", "pixnet_mfma", "force_mr", "mr4_min",
                                              "mr4_nt2_max", "mr2_min", "wide_tiles", "roll_mr", "roll_zt", "roll_rp", "deconv_mr", "sr_waves", "enc_gen_mr3", "c3_lean"};
long g_opt[EFFI_OPT_COUNT];
std::once_flag g_opt_once;
void load_options() {
    for (int i = 0; i < EFFI_OPT_COUNT; ++i) {
        char env[64] = "EFFI_";
        size_t n = strlen(env);
        for (const char* c = kOptName[i]; *c && n + 1 < sizeof(env); ++c) env[n++] = (*c >= 'a' && *c <= 'z') ? (char)(*c - 32) : *c;
        env[n] = 0;
        const char* v = getenv(env);            // the ONLY getenv of the library, once per process
        g_opt[i] = (v && *v) ? atol(v) : EFFI_OPT_UNSET;
    }
}
int find_option(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < EFFI_OPT_COUNT; ++i)
        if (strcmp(name, kOptName[i]) == 0) return i;
    return -1;
}
}  // namespace

long effi_option(int id) {
    std::call_once(g_opt_once, load_options);
    return __atomic_load_n(&g_opt[id], __ATOMIC_RELAXED);
}

extern "C" int effi_set_option(const char* name, long value) {
    const int id = find_option(name);
    if (id < 0) return EFFI_ERR_BADARG;
    std::call_once(g_opt_once, load_options);
    __atomic_store_n(&g_opt[id], value, __ATOMIC_RELAXED);
    return EFFI_OK;
}

extern "C" long effi_get_option(const char* name) {
    const int id = find_option(name);
    return id < 0 ? EFFI_OPT_UNSET : effi_option(id);
}

extern "C" long effi_option_unset(void) { return EFFI_OPT_UNSET; }
