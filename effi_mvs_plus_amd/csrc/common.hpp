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

static inline hipStream_t effi_s(effi_stream_t s) {
    (void)hipGetLastError();            // drop a stale pending error (see EFFI_LAUNCH_CHECK); sticky errors come back on the next check
    return reinterpret_cast<hipStream_t>(s);
}
static inline int effi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

struct EffiPtrList {            // by-value kernel argument: device pointers of the source views
    const float* p[EFFI_MAX_VIEWS + 1];
};
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

template <int COUT>
__device__ __forceinline__ void effi_c1k7_relu_tile_x3(const float* __restrict__ in, const float* __restrict__ wgt,
                                                       const float* __restrict__ bias, int h, int w, float* __restrict__ out,
                                                       int bx, int by) {
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
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * n + 4 * lk + r;
                out[co * hw + pix] = fmaxf(acc[m][n][r] + bias[co], 0.0f);
            }
    }
}
