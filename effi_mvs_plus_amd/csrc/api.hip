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
                                              "mr4_nt2_max", "mr2_min", "wide_tiles", "roll_mr", "roll_zt", "roll_rp", "deconv_mr", "sr_waves", "enc_gen_mr3", "c3_lean", "dyn_win"};
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

// ---- diagnostics: fill the LDS of every CU with a pattern (tools/stress_c8_corun.py: what a kernel reads from LDS without having written
// it is whatever the previous workgroup on that CU left there; after this launch that is `pattern`) ----
namespace {
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned pattern, unsigned* sink) {
    __shared__ unsigned buf[16 * 1024];                               // 64 KB: two or three workgroups per CU cover its 160 KB over a few launches
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) buf[i] = pattern;
    __syncthreads();
    if (buf[(threadIdx.x * 61) & 16383] != pattern) sink[0] = 1;      // keeps the stores alive
}
}  // namespace

extern "C" int effi_debug_poison_lds(unsigned pattern, void* sink4, effi_stream_t stream) {
    if (!sink4) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(lds_poison_kernel, dim3(2048), dim3(256), 0, effi_s(stream), pattern, reinterpret_cast<unsigned*>(sink4));
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

// ---- diagnostics: probe of a write-after-read on a packed fp32 instruction's source (tools/probe_pk_war.py) ---------------------------
// Round 4: the compiler's packed form of conv3d_c8to1_kernel (v_pk_fma_f32 on register pairs assembled with v_mov_b32, the NEXT
// instruction already overwriting the low register of the pair) returned wrong low halves in lanes 48-63 -- only while another queue
// ran matrix-core GEMMs on the same CUs.  The probe repeats exactly that pair of instructions: acc += (x, y) * w, then x's register is
// overwritten with a different value and restored.  VARIANT 0: back to back; 1: one s_nop between; 2: the overwrite is another
// v_pk instruction (v_pk_mov_b32); 3: plain v_fma_f32 instead of the packed one.  out[2 tid], out[2 tid + 1] = the accumulators.
namespace {
template <int VARIANT>
__global__ __launch_bounds__(256) void pk_war_probe_kernel(float* __restrict__ out, int reps) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const float x = 1.0f + (float)(tid & 15) * 0.0625f, y = 2.0f, bad = 1000.0f, w = 1.0f;      // sums of up to 2^17 terms stay exact
    float r0, r1;
    asm volatile(
        "v_mov_b32 v10, %2\n"
        "v_mov_b32 v11, %3\n"
        "v_mov_b32 v12, 0\n"
        "v_mov_b32 v13, 0\n"
        "v_mov_b32 v14, %5\n"
        "v_mov_b32 v15, %5\n"
        "s_nop 7\n"
        "s_mov_b32 s20, %6\n"
        "1:\n"
        ".rept 16\n"
        ".if %c7 == 3\n"
        "v_fma_f32 v12, v10, v14, v12\n"
        "v_fma_f32 v13, v11, v14, v13\n"
        ".else\n"
        "v_pk_fma_f32 v[12:13], v[10:11], v[14:15], v[12:13] op_sel_hi:[1,0,1]\n"
        ".endif\n"
        ".if %c7 == 1\n"
        "s_nop 0\n"
        ".endif\n"
        ".if %c7 == 2\n"
        "v_pk_mov_b32 v[10:11], v[16:17], v[10:11] op_sel:[0,1]\n"
        ".else\n"
        "v_mov_b32 v10, %4\n"
        ".endif\n"
        "v_mov_b32 v10, %2\n"
        ".endr\n"
        "s_sub_u32 s20, s20, 1\n"
        "s_cmp_lg_u32 s20, 0\n"
        "s_cbranch_scc1 1b\n"
        "s_nop 7\n"
        "v_mov_b32 %0, v12\n"
        "v_mov_b32 %1, v13\n"
        : "=v"(r0), "=v"(r1)
        : "v"(x), "v"(y), "v"(bad), "v"(w), "s"(reps), "n"(VARIANT)
        : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "s20", "scc");
    out[2 * tid] = r0;
    out[2 * tid + 1] = r1;
}
}  // namespace

extern "C" int effi_debug_pk_war_probe(float* out, int blocks, int reps, int variant, effi_stream_t stream) {
    if (!out || blocks < 1 || reps < 1) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    switch (variant) {
        case 0: hipLaunchKernelGGL(pk_war_probe_kernel<0>, dim3(blocks), dim3(256), 0, st, out, reps); break;
        case 1: hipLaunchKernelGGL(pk_war_probe_kernel<1>, dim3(blocks), dim3(256), 0, st, out, reps); break;
        case 3: hipLaunchKernelGGL(pk_war_probe_kernel<3>, dim3(blocks), dim3(256), 0, st, out, reps); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
