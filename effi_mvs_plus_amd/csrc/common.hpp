// Shared device helpers for the gfx950 kernels.  Wave = 64 lanes; no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/effi_mvs_hip.h"

// Launch status of THIS call: every entry point starts by taking (and thereby clearing) whatever non-sticky error an earlier
// call of the host framework or of this library left pending (effi_s(), below, which every entry uses to unwrap its stream), so
// a failed launch is reported once, by the entry that failed, and not by every later call.  The check itself peeks: the error
// stays readable for the host framework (hipGetLastError) after the entry has returned EFFI_ERR_LAUNCH.
#define EFFI_LAUNCH_CHECK()                                   \
    do {                                                      \
        if (hipPeekAtLastError() != hipSuccess) return EFFI_ERR_LAUNCH; \
    } while (0)

// zero page of the current device (caller-registered, api.hip); nullptr when none is registered
const float* effi_zero_page();

// Tuning / A-B switches of the library (api.hip): a table of integers, initialised ONCE from the environment (variable EFFI_<NAME>)
// and changed afterwards only through effi_set_option -- no entry point calls getenv.
enum EffiOption {
    EFFI_OPT_WARP_LDS_KB,        // stage-1 warp kernel: LDS window in KB (unset = 72), 0 = window kernel on global loads, -1 = direct-gather kernel
    EFFI_OPT_DYN_FORM,           // stage-2/3 warp kernel: 1 = channel-split lanes instead of hypothesis-per-lane
    EFFI_OPT_DYN_SETUP_EXACT,    // stage-2/3 warp kernel: 1 = IEEE divisions in the projection
    EFFI_OPT_DYN_XCHG,           // diagnostic builds (-DEFFI_DIAG_LANE_EXCHANGE) only: 1 = DPP, 2 = ds_bpermute exchange of set-ups
    EFFI_OPT_PIXNET_MFMA,        // view-weight net: 0 = vector-ALU kernel
    EFFI_OPT_FORCE_MR,           // split-precision 3x3 convolutions: rows per wave (1, 2, 4) instead of the rule
    EFFI_OPT_MR4_MIN, EFFI_OPT_MR4_NT2_MAX, EFFI_OPT_MR2_MIN,     // thresholds of that rule (workgroup counts)
    EFFI_OPT_WIDE_TILES,         // 0 / 1: 4 x 64 tiles never / always
    EFFI_OPT_ROLL_MR, EFFI_OPT_ROLL_ZT, EFFI_OPT_ROLL_RP,         // rolling 3-D convolution: rows per wave, planes per workgroup, row-pair operand
    EFFI_OPT_DECONV_MR,          // transposed 3-D convolution: rows per wave
    EFFI_OPT_SR_WAVES,           // split-resident 3x3 convolutions: unset = 512-thread workgroups for 1 row per wave x 6 N-tiles; 8 = wherever the rule picks 1 or 2 rows per wave; 4 = never
    EFFI_OPT_ENC_GEN_MR3,        // generated-input pair kernel: 0 = 4 rows per wave where the rule says so (default: 3)
    EFFI_OPT_C3_LEAN,            // 3-D end layers (1 -> 8, 8 -> 1 channels): 0 = the general vector-ALU bodies instead of the dedicated kernels
    EFFI_OPT_DYN_WIN,            // stage-2/3 warp kernel: unset = LDS-window form with its full window, n > 0 = window of n pixels, 0 = window form on global loads, -1 = the gather kernel
    EFFI_OPT_COUNT
};
constexpr long EFFI_OPT_UNSET = -0x7fffffffL;
long effi_option(int id);        // current value, EFFI_OPT_UNSET when neither the environment nor effi_set_option gave one

static inline hipStream_t effi_s(effi_stream_t s) {
    (void)hipGetLastError();            // drop a stale pending error (see EFFI_LAUNCH_CHECK); sticky errors come back on the next check
    return reinterpret_cast<hipStream_t>(s);
}
static inline int effi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

struct EffiPtrList {            // by-value kernel argument: device pointers of the source views
    const float* p[EFFI_MAX_VIEWS + 1];
    // View-table form (effi_*_tbl_f32 entries): non-null = the pointers live in DEVICE memory, tbl[0] = the reference map,
    // tbl[1 + v] = source view v (EFFI_MAX_VIEWS + 2 entries, unused ones null).  A captured hipGraph bakes kernel arguments in, a
    // table it reads at run time does not: the evaluation-set runner (scan_eval.py) points a replay at the cached feature maps of
    // its (reference, sources) item instead of copying 150 MB of them into static inputs.
    const float* const* tbl;
};
// the reference map of a forward warp kernel, whichever form the launch uses (uniform select; nothing is copied: a mutable copy of
// the list costs the kernels a private segment and 8-18 registers)
__device__ __forceinline__ const float* effi_resolve_views(const float* ref, const EffiPtrList& l) {
    return l.tbl ? l.tbl[0] : ref;
}
struct EffiOutList {
    float* p[EFFI_MAX_VIEWS + 1];
};

// ---- DPP cross-lane adds inside aligned groups of 2 / 4 / 8 lanes (full-rate VALU, no LDS) ----
__device__ __forceinline__ float effi_dpp_f(float v, int ctrl_sel) {
    int i = __float_as_int(v);
    int r;
    switch (ctrl_sel) {
        case 0:  r = __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
        case 1:  r = __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
        default: r = __builtin_amdgcn_update_dpp(0, i, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
    }
    return __int_as_float(r);
}

template <int LPP>
__device__ __forceinline__ float effi_group_sum(float v) {
    static_assert(LPP == 1 || LPP == 2 || LPP == 4 || LPP == 8, "lanes per pixel");
    if (LPP >= 2) v += effi_dpp_f(v, 0);
    if (LPP >= 4) v += effi_dpp_f(v, 1);
    if (LPP >= 8) v += effi_dpp_f(v, 2);
    return v;
}
template <int LPP>
__device__ __forceinline__ float effi_group_max(float v) {
    if (LPP >= 2) v = fmaxf(v, effi_dpp_f(v, 0));
    if (LPP >= 4) v = fmaxf(v, effi_dpp_f(v, 1));
    if (LPP >= 8) v = fmaxf(v, effi_dpp_f(v, 2));
    return v;
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): consecutive logical tiles land on the
// same XCD so that neighbouring tiles share that XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ int effi_xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__device__ __forceinline__ float effi_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Gate non-linearities of the SPLIT-precision convolution epilogues (conv2d.hip, conv_epilogue_store_t and the mask head): the
// convolution result they are applied to already carries ~1e-5 of relative error (three bf16 partial products), so the exponential is
// v_exp_f32 on x * log2(e) (argument rounding ~ |x| * 6e-8 relative) and the division a refined reciprocal (rcp + one Newton step,
// <= 1 ulp) -- ~7 vector instructions instead of ~24 for expf + IEEE division, 16 times per lane and tile in the GRU's z / r
// convolution, whose time is 56 % vector-instruction issue.  Absolute error <= 2e-7; NaN inputs propagate.  The exact-fp32 kernels
// (precision "fp32") keep expf / tanhf / IEEE division; -DEFFI_EXACT_EPILOGUES restores them here too (A/B builds).
__device__ __forceinline__ float effi_rcp_refined(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float effi_exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float effi_sigmoid_split(float x) {
#ifdef EFFI_EXACT_EPILOGUES
    return effi_sigmoid(x);
#else
    const float r = effi_rcp_refined(1.0f + effi_exp_fast(fminf(-x, 60.0f)));       // e^60 is finite; sigmoid(-60) = 9e-27
    return (x != x) ? x : r;
#endif
}
__device__ __forceinline__ float effi_tanh_split(float x) {
#ifdef EFFI_EXACT_EPILOGUES
    return tanhf(x);
#else
    const float r = 1.0f - 2.0f * effi_rcp_refined(1.0f + effi_exp_fast(fminf(2.0f * x, 60.0f)));   // tanh(30) = 1 in fp32
    return (x != x) ? x : r;
#endif
}

// scale_inv_depth (models/Effi_MVS_plus.py:138-148): normalised inverse depth -> (scaled, depth)
__device__ __forceinline__ float effi_inv_to_depth(float inv, float lo, float hi) {
    // min_disp = 1/max_depth with max_depth = 1/lo (models/Effi_MVS_plus.py:413-414): two reciprocals
    const float max_depth = 1.0f / lo, min_depth = 1.0f / hi;
    const float min_disp = 1.0f / max_depth, max_disp = 1.0f / min_depth;
    float s = min_disp + (max_disp - min_disp) * inv;
    s = fmaxf(s, 1e-4f);
    return 1.0f / s;
}

// depth_to_disp(depth, depth_min_, depth_max_) (models/Effi_MVS_plus.py:151-164) with the global range of the hypotheses
__device__ __forceinline__ float effi_depth_to_inv(float depth, float lo, float hi) {
    const float max_depth = 1.0f / lo, min_depth = 1.0f / hi;
    const float min_disp = 1.0f / max_depth, max_disp = 1.0f / min_depth;
    const float den = (max_disp - min_disp) + 1e-10f;
    const float s_ = 1.0f / depth;
    return (s_ - min_disp) / den;
}

// Divisions of the volume look-ups (lookup1d_index, effi_getcost_pixel): a refined reciprocal (v_rcp_f32 + one Newton step, <= 1 ulp) and,
// for a quotient, one residual step on top -- 3 / 6 instructions instead of the ~11 of the IEEE sequence, correctly rounded except for
// rare 1-ulp cases of CONTINUOUS functions of the query depth (linear interpolation has no jumps), as the warp kernels' projection
// (warpcorr.hip, project_xy).  The generated-input convolution spends 660 instructions per staged pixel in this prologue, 23 IEEE
// divisions among them.  -DEFFI_EXACT_LOOKUP restores the IEEE divisions (A/B builds).  The maps the path RETURNS
// (effi_inv_to_depth / effi_depth_to_inv) keep their IEEE divisions.
__device__ __forceinline__ float effi_lk_rcp(float b) {
#ifdef EFFI_EXACT_LOOKUP
    return 1.0f / b;
#else
    const float r0 = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r0, 1.0f), r0, r0);
#endif
}
__device__ __forceinline__ float effi_lk_div(float a, float b, float rb) {      // a / b given rb = effi_lk_rcp(b)
#ifdef EFFI_EXACT_LOOKUP
    (void)rb;
    return a / b;
#else
    const float q = a * rb;
    return fmaf(fmaf(-q, b, a), rb, q);
#endif
}

// the sampling position of a query in a Dp-long vector: first tap x0 and the two weights (the arithmetic depends on Dp, not on the volume)
__device__ __forceinline__ void lookup1d_index(int Dp, float q_depth, float dmin, float dmax, int& x0, float& w0, float& w1) {
    const float scaled = effi_lk_rcp(q_depth);                          // depth_to_disp, :156-164
    const float min_disp = effi_lk_rcp(dmax), max_disp = effi_lk_rcp(dmin);
    const float den = (max_disp - min_disp) + 1e-10f;
    const float disp = effi_lk_div(scaled - min_disp, den, effi_lk_rcp(den));
    const float dm1 = (float)(Dp - 1);
    const float t = disp * dm1;                                        // :123
    const float g = effi_lk_div(2.0f * t, dm1, effi_lk_rcp(dm1)) - 1.0f;   // :107
    float ix = ((g + 1.0f) * 0.5f) * dm1;                              // grid_sample, align_corners=True (x / 2 = x * 0.5 exactly)
    ix = fminf(fmaxf(ix, -2.0f), dm1 + 2.0f);                          // neutral: both taps stay out of range
    const float x0f = floorf(ix);
    x0 = (int)x0f;
    w1 = ix - x0f;
    w0 = (x0f + 1.0f) - ix;
}
__device__ __forceinline__ float lookup1d_fetch(const float* __restrict__ vol, long dstride, int Dp, int x0, float w0, float w1) {
    float r = 0.0f;
    if (x0 >= 0 && x0 <= Dp - 1) r = vol[x0 * dstride] * w0;
    if (x0 + 1 >= 0 && x0 + 1 <= Dp - 1) r = r + vol[(x0 + 1) * dstride] * w1;
    return r;
}
// GetCost.forward for ONE pixel (models/Effi_MVS_plus.py:257-303 behind scale_inv_depth :138-148): NQ hypotheses at inverse depth
// +-(NQ/2) intervals around the current estimate (module.py:554-570), looked up in the stage's two cached volumes -> cost[0..NQ) from
// cur_vol, cost[NQ..2NQ) from reg_vol.  cur / reg point at the pixel's vector (element stride cds / rds).  Shared by
// getcost_conv1x1_block (volume_ops.hip) and the generated-input convolution (conv2d_x3.hpp): one arithmetic, bitwise.
template <int NQ>
__device__ __forceinline__ void effi_getcost_pixel(float inv_or_depth, int input_is_depth, const float* __restrict__ disp_range, int n_range,
                                                   float itv, const float* __restrict__ cur, long cds, int Dcur,
                                                   const float* __restrict__ reg, long rds, int Dreg, float rlo, float rhi, float (&cost)[2 * NQ]) {
    float depth = inv_or_depth;
    if (!input_is_depth) depth = effi_inv_to_depth(depth, disp_range[0], disp_range[n_range - 1]);
    const float dv = effi_lk_rcp(depth);
    const float half = (float)(NQ / 2) * itv;
    const float smin = fmaxf(dv - half, 1e-4f);
    const float smax = fminf(fmaxf(dv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(NQ - 1);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const float s = fmaxf(smin + (float)k * step, 1e-5f);
        const float qd = effi_lk_rcp(s);
        // both volumes of a stage are equally long on the path: one position per query serves both (bitwise the same values)
        int x0;
        float w0, w1;
        lookup1d_index(Dcur, qd, rlo, rhi, x0, w0, w1);
        cost[k] = lookup1d_fetch(cur, cds, Dcur, x0, w0, w1);
        if (Dreg != Dcur) lookup1d_index(Dreg, qd, rlo, rhi, x0, w0, w1);
        cost[NQ + k] = lookup1d_fetch(reg, rds, Dreg, x0, w0, w1);
    }
}

// tile of the 7x7 kernels (64 x 4, i.e. one 256-byte row run per wave store instead of two 128-byte ones, measured the same: 25.2 us)
#define EFFI_C1K7_TX 32
#define EFFI_C1K7_TY 8

// 7x7 convolution of a single-channel map + ReLU, one (EFFI_C1K7_TX x EFFI_C1K7_TY pixel, 16 channel) tile of it: shared by
// conv2d_c1k7_relu_kernel (conv2d.hip) and encoder_inputs_kernel (volume_ops.hip).  Block of 256 threads.
template <int COUT>
__device__ __forceinline__ void effi_c1k7_relu_tile(const float* __restrict__ in, const float* __restrict__ wgt,
                                                    const float* __restrict__ bias, int h, int w, float* __restrict__ out,
                                                    int bx, int by, int bz) {
    constexpr int TXX = EFFI_C1K7_TX, TYY = EFFI_C1K7_TY, IWX = TXX + 6, IHY = TYY + 6, CG = 16;
    __shared__ float tile[IHY * IWX];
    const int tx = threadIdx.x % TXX, ty = threadIdx.x / TXX;
    const int x0 = bx * TXX, y0 = by * TYY;
    const int c0 = bz * CG;
    for (int e = threadIdx.x; e < IHY * IWX; e += 256) {
        const int yy = e / IWX, xx = e - yy * IWX;
        const int gy = y0 - 3 + yy, gx = x0 - 3 + xx;
        tile[e] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? in[(long)gy * w + gx] : 0.0f;
    }
    __syncthreads();
    const float* __restrict__ wg = wgt + c0;
    // channel pairs as 2-vectors: the 784 FMAs of a thread become 392 packed ones (v_pk_fma_f32, weights as SGPR pairs)
    typedef float effi_f32x2 __attribute__((ext_vector_type(2)));
    effi_f32x2 acc[CG / 2];
#pragma unroll
    for (int c = 0; c < CG / 2; ++c) acc[c] = effi_f32x2{bias[c0 + 2 * c], bias[c0 + 2 * c + 1]};
#pragma unroll
    for (int k = 0; k < 49; ++k) {
        const float v = tile[(ty + k / 7) * IWX + tx + k % 7];
        const effi_f32x2 vv = {v, v};
#pragma unroll
        for (int c = 0; c < CG / 2; ++c) {
            const effi_f32x2 wv = {wg[k * COUT + 2 * c], wg[k * COUT + 2 * c + 1]};
            acc[c] = __builtin_elementwise_fma(vv, wv, acc[c]);
        }
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x >= w || y >= h) return;
    const long hw = (long)h * w, pix = (long)y * w + x;
#pragma unroll
    for (int c = 0; c < CG; ++c) out[(c0 + c) * hw + pix] = fmaxf(acc[c / 2][c & 1], 0.0f);
}

// The same tile on the matrix cores in split precision (hi*hi + lo*hi + hi*lo on v_mfma_f32_16x16x32_bf16, fp32 accumulation): the
// VALU form above issues 392 packed FMAs per pixel and 16 channels.  K = (ky, kx8): a lane quarter lk owns kernel row ky = 4 s + lk
// in K-step s, its 8 K-values are 8 consecutive input columns of that row (kx = 0..6 and one zero weight; row 7 is all zero), read
// from the fp32 LDS tile with 8 conflict-free ds_read_b32 and split in registers.  One workgroup computes ALL COUT channels of a
// 32 x 8 pixel tile (16 row segments of 16 pixels, 4 per wave): the split input fragments are shared by the COUT / 16 output tiles.
// Weight fragments are built once per workgroup from the fp32 [49][COUT] table.
typedef float effi_f32x4_t __attribute__((ext_vector_type(4)));
typedef float effi_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 effi_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 effi_bf16x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void effi_split8(const float (&x)[8], effi_bf16x8_t& hi, effi_bf16x8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const effi_f32x2_t v = {x[e], x[e + 1]};
        const effi_bf16x2_t h2 = __builtin_convertvector(v, effi_bf16x2_t);
        const effi_bf16x2_t l2 = __builtin_convertvector(v - __builtin_convertvector(h2, effi_f32x2_t), effi_bf16x2_t);
        hi[e] = h2[0]; hi[e + 1] = h2[1];
        lo[e] = l2[0]; lo[e + 1] = l2[1];
    }
}

typedef __bf16 effi_bf16x4_t __attribute__((ext_vector_type(4)));

// Split-resident activation maps ("SR"): what a split-precision convolution wants to read is not the fp32 value x but its two
// bf16 halves hi = bf16(x), lo = bf16(x - hi) -- the same 4 bytes -- in the [octet of channels][pixel][8] order of its A
// fragments.  The layers of the GRU update block feed each other (models/update.py:33-49,69-99,114-141): every map is read by
// 1-2 convolutions, each of which used to redo the split for every staged pixel, halo included (half of the 1,054 vector
// instructions per wave of the z / r convolution).  Here the PRODUCER's epilogue splits once and stores
//     map[octet o][part q in {hi, lo}][row y + 1][column x + 1][channel e]      (bf16; [hp][wp] pixels per plane)
// and the consumer's staging is a plain 16-byte copy.  The one-pixel zero border (and zeros up to the tile overhang on the right /
// bottom: hp >= roundup(h, 16) + 2, wp >= roundup(w, 64) + 2) is the convolution's padding: no bounds tests, no zero page.  The
// values are bitwise those the consumer would have computed from the fp32 map (same conversion instructions), so a chain through
// SR maps equals the chain through fp32 maps bit for bit.
// The MFMA result layout gives lane l = li + 16 lk the 4 consecutive channels co0 = 16 n + 4 lk .. + 3 of pixel li: lanes l and l ^ 16
// hold the two halves of ONE octet of ONE pixel.  Stored as they stand that is two 8-byte pieces per 16-byte unit from
// non-adjacent lanes -- measured 10-25 % slower per kernel than the fp32 stores they replace.  Instead the pair trades halves
// (v_permlane16_swap: rows 1 / 3 of the first operand <-> rows 0 / 2 of the second): the even-lk lane ends up with the whole HI
// octet, the odd-lk lane with the whole LO octet, and every lane issues ONE 16-byte store -- per wave instruction four fully
// written 256-byte runs (hi / lo planes of two octets).  Must be called by both lanes of a pair (the pair shares its pixel, so a
// bounds test on the pixel is pair-uniform).
__device__ __forceinline__ void effi_sr_store4(unsigned short* __restrict__ base, int hp, int wp, int co0, int y, int x, const effi_f32x4_t v) {
    typedef unsigned effi_u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned effi_u32x4_t __attribute__((ext_vector_type(4)));
    const effi_bf16x4_t h4 = __builtin_convertvector(v, effi_bf16x4_t);
    const effi_u32x2_t hu = __builtin_bit_cast(effi_u32x2_t, h4);
#ifndef EFFI_BF16_ONLY
    const effi_bf16x4_t l4 = __builtin_convertvector(v - __builtin_convertvector(h4, effi_f32x4_t), effi_bf16x4_t);
    const effi_u32x2_t lu = __builtin_bit_cast(effi_u32x2_t, l4);
#else
    const effi_u32x2_t lu = hu;                 // hi only: the even lane still needs its partner's hi half
#endif
    const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
    // even lk: {own hi, partner's hi} = channels 0-3, 4-7 of the hi octet; odd lk: {partner's lo, own lo} = the lo octet
    const effi_u32x4_t oct = {s0[0], s1[0], s0[1], s1[1]};
    const int part = (co0 >> 2) & 1;            // = lk & 1
#ifdef EFFI_BF16_ONLY
    if (part) return;
#endif
    const long u = ((long)((co0 >> 3) * 2 + part) * hp + (y + 1)) * wp + (x + 1);     // 16-byte unit
    *reinterpret_cast<effi_u32x4_t*>(base + u * 8) = oct;
}

// The same store with the 16-byte unit index computed by the caller (u = ((co0 >> 3) * 2 + part) * hp * wp + (y + 1) * wp + x + 1 with
// part = (co0 >> 2) & 1): the batched epilogue of conv2d_x3.hpp derives the units of a lane's N-tiles from one base by adding a constant.
__device__ __forceinline__ void effi_sr_store4_at(unsigned short* __restrict__ base, unsigned u, int part, const effi_f32x4_t v) {
    typedef unsigned effi_u32x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned effi_u32x4_t __attribute__((ext_vector_type(4)));
    const effi_bf16x4_t h4 = __builtin_convertvector(v, effi_bf16x4_t);
    const effi_u32x2_t hu = __builtin_bit_cast(effi_u32x2_t, h4);
#ifndef EFFI_BF16_ONLY
    const effi_bf16x4_t l4 = __builtin_convertvector(v - __builtin_convertvector(h4, effi_f32x4_t), effi_bf16x4_t);
    const effi_u32x2_t lu = __builtin_bit_cast(effi_u32x2_t, l4);
#else
    const effi_u32x2_t lu = hu;
#endif
    const auto s0 = __builtin_amdgcn_permlane16_swap(hu[0], lu[0], false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(hu[1], lu[1], false, false);
    const effi_u32x4_t oct = {s0[0], s1[0], s0[1], s1[1]};
#ifdef EFFI_BF16_ONLY
    if (part) return;
#endif
    (void)part;
    *reinterpret_cast<effi_u32x4_t*>(base + (size_t)u * 8) = oct;
}

// 8 consecutive channels (one octet, co0 % 8 == 0) of pixel (y, x): one 16-byte store per part
__device__ __forceinline__ void effi_sr_store8(unsigned short* __restrict__ base, int hp, int wp, int co0, int y, int x, const float (&v)[8]) {
    effi_bf16x8_t hi, lo;
    effi_split8(v, hi, lo);
    const long u = ((long)((co0 >> 3) * 2) * hp + (y + 1)) * wp + (x + 1);
    *reinterpret_cast<effi_bf16x8_t*>(base + u * 8) = hi;
    *reinterpret_cast<effi_bf16x8_t*>(base + (u + (long)hp * wp) * 8) = lo;
}

template <int COUT>
__device__ __forceinline__ void effi_c1k7_relu_tile_x3(const float* __restrict__ in, const float* __restrict__ wgt,
                                                       const float* __restrict__ bias, int h, int w, float* __restrict__ out,
                                                       int bx, int by, unsigned short* __restrict__ out_sr = nullptr, int sr_hp = 0,
                                                       int sr_wp = 0) {
    constexpr int TXX = EFFI_C1K7_TX, TYY = EFFI_C1K7_TY, IWX = 40, IHY = TYY + 7, NT = COUT / 16;
    static_assert(TXX == 32 && TYY == 8, "16 row segments of 16 pixels, 4 per wave");
    __shared__ float tile[IHY * IWX];                      // rows y0-3 .. y0+TYY+3 (the extra row is multiplied by zero weights)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int x0 = bx * TXX, y0 = by * TYY;
    for (int e = tid; e < IHY * IWX; e += 256) {
        const int yy = e / IWX, xx = e - yy * IWX;
        const int gy = y0 - 3 + yy, gx = x0 - 3 + xx;
        tile[e] = (yy < IHY - 1 && gy >= 0 && gy < h && gx >= 0 && gx < w) ? in[(long)gy * w + gx] : 0.0f;
    }
    // weight fragments: lane (cout j = li, quarter lk) holds W[16 n + j][ky = 4 s + lk][kx = 0..7]
    effi_bf16x8_t wh[2][NT], wl[2][NT];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int ky = 4 * s_ + lk;
            float wv8[8];
#pragma unroll
            for (int kx = 0; kx < 8; ++kx)
                wv8[kx] = (ky < 7 && kx < 7) ? wgt[(ky * 7 + kx) * COUT + 16 * n + li] : 0.0f;
            effi_split8(wv8, wh[s_][n], wl[s_][n]);
        }
    __syncthreads();
    effi_f32x4_t acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = effi_f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    // segment m of wave wv: tile row 2 wv + (m >> 1), columns 16 (m & 1) .. + 15
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ry = 2 * wv + (m >> 1), cx = 16 * (m & 1) + li;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const float* row = tile + (ry + 4 * s_ + lk) * IWX + cx;
            float x8[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x8[i] = row[i];
            effi_bf16x8_t ah, al;
            effi_split8(x8, ah, al);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s_][n], ah, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s_][n], ah, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s_][n], al, acc[m][n], 0, 0, 0);
            }
        }
    }
    const long hw = (long)h * w;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int y = y0 + 2 * wv + (m >> 1), x = x0 + 16 * (m & 1) + li;
        if (y >= h || x >= w) continue;
        const long pix = (long)y * w + x;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            effi_f32x4_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(acc[m][n][r] + bias[16 * n + 4 * lk + r], 0.0f);
            if (out_sr) {                      // split-resident output (uniform branch)
                effi_sr_store4(out_sr, sr_hp, sr_wp, 16 * n + 4 * lk, y, x, v);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(16 * n + 4 * lk + r) * hw + pix] = v[r];
            }
        }
    }
}
