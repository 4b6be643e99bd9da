// K5/K6: 3-D convolutions (kernel 3, padding 1) of the cost regulariser and the cross-scale
// propagation blocks, BatchNorm folded into (weight, bias) by the host.
// Reference: models/module.py:124-160 (Conv3d), :168-203 (Deconv3d), :439-463, :501-516.
//
// Direct convolution on the vector ALUs: channel counts are 1..32, so an implicit GEMM would have
// N = Cout <= 32 and waste most of an MFMA tile; instead a 256-thread block (32 x 8 lanes in x,y)
// stages a halo'd input tile of a few input channels in LDS, every thread keeps ZPT x COUT_T
// accumulators in registers, each LDS value feeds >= 8 FMAs, and the weights (uniform across lanes)
// arrive through the scalar cache as SGPR operands of v_fma.  Planar [C][D][h][w] layout: lanes run
// along x, so tile loads and output stores are coalesced.
#include "common.hpp"

namespace {

#ifndef EFFI_C3_TX
#define EFFI_C3_TX 32          // tile of the vector-ALU kernels (A/B builds: -DEFFI_C3_TX=64 -DEFFI_C3_TY=4)
#define EFFI_C3_TY 8
#endif
constexpr int TX = EFFI_C3_TX, TY = EFFI_C3_TY;
static_assert(TX * TY == 256, "one thread per tile pixel");

struct SrcSet {                    // channel concatenation of up to EFFI_MAX_SRC planar tensors
    const float* p[EFFI_MAX_SRC];
    int ch[EFFI_MAX_SRC];
};

__device__ __forceinline__ const float* src_channel(const SrcSet& s, int c, long plane) {
    // channel c of cat(s.p[0], s.p[1], s.p[2]) -> pointer to its [D][h][w] block
    if (c < s.ch[0]) return s.p[0] + (long)c * plane;
    c -= s.ch[0];
    if (c < s.ch[1]) return s.p[1] + (long)c * plane;
    c -= s.ch[1];
    return s.p[2] + (long)c * plane;
}

// ------------------------------------------------------------------------------------------------
// forward convolution, stride (SZ, SXY, SXY)
// ------------------------------------------------------------------------------------------------
// DENSE: cout == COUT_T (every layer still on this kernel: cout 8 or 1), so the weight row stride is a compile-time constant
// and all 27*COUT_T weights of a channel are scalar loads at immediate offsets from one base (measured before: as many scalar
// ALU instructions as vector ones, spent on per-tap address arithmetic with the runtime stride).
// (a __device__ body so that two independent convolutions can share one grid: conv3d_k3_pair_kernel below)
template <int COUT_T, int SZ, int SXY, int ZPT, bool DENSE>
__device__ __forceinline__ void conv3d_k3_body(SrcSet src, int cin, const float* __restrict__ wgt,
                                               const float* __restrict__ bias, int cout,
                                               int D, int h, int w, int Do, int ho, int wo,
                                               int relu, const float* __restrict__ skip,
                                               float* __restrict__ out) {
    constexpr int CC = (SXY == 1) ? 4 : ((SZ == 1) ? 1 : 2);
    constexpr int IZ = (ZPT - 1) * SZ + 3, IY = (TY - 1) * SXY + 3, IX = (TX - 1) * SXY + 3;
    constexpr int PLANE = IZ * IY * IX;
    constexpr int PSZ = IY * IX, NPL = (PSZ + 255) / 256;   // plane size, fill slots per thread and plane
    __shared__ float tile[CC * PLANE];

    // 1-D grid, XCD-aware: each XCD owns a contiguous run of (z-group, xy-tile) work items, so tiles that
    // share halo planes / cache lines hit the same L2.  Order: xy-tile fastest, then z-group, then cout group.
    const int tiles_x = (wo + TX - 1) / TX, tiles_xy = tiles_x * ((ho + TY - 1) / TY), nzg = (Do + ZPT - 1) / ZPT;
    int lid = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_xy = lid % tiles_xy;
    lid /= tiles_xy;
    const int zg = lid % nzg, cg = lid / nzg;
    const int by = tile_xy / tiles_x, bx = tile_xy - by * tiles_x;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int ox = bx * TX + tx, oy = by * TY + ty, oz0 = zg * ZPT;
    const int co0 = cg * COUT_T;
    const int iz0 = oz0 * SZ - 1, iy0 = by * TY * SXY - 1, ix0 = bx * TX * SXY - 1;
    const long in_plane = (long)D * h * w;

    int poff[NPL];                            // offset of this thread's fill elements inside a z-slice, -1 = padding
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = threadIdx.x + k * 256;
        const int ly = e / IX, lx = e - ly * IX;
        const int gy = iy0 + ly, gx = ix0 + lx;
        poff[k] = ((e < PSZ) & (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w)) ? gy * w + gx : -1;
    }

    float acc[ZPT][COUT_T];
#pragma unroll
    for (int z = 0; z < ZPT; ++z)
#pragma unroll
        for (int c = 0; c < COUT_T; ++c) acc[z][c] = 0.0f;

    // Register prefetch: the loads of chunk k+1 are issued before the FMAs of chunk k and only waited
    // for when they are written to LDS, so the tile-fill latency hides behind the multiply phase.
    float pf[CC][IZ][NPL];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const bool cok = (c0 + c < cin);
            const float* __restrict__ cp = src_channel(src, cok ? c0 + c : 0, in_plane);
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz) {
                const int gz = iz0 + lz;
                const bool zok = cok & (gz >= 0) & (gz < D);
                const float* __restrict__ zp = cp + (zok ? (long)gz * h * w : 0);
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const float t = zp[max(poff[k], 0)];
                    pf[c][lz][k] = (zok & (poff[k] >= 0)) ? t : 0.0f;
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz)
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const int e = threadIdx.x + k * 256;
                    if (e < PSZ) tile[c * PLANE + lz * PSZ + e] = pf[c][lz][k];
                }
    };

    // ZPT == 1 (low-resolution layers, few waves per SIMD): the next chunk is prefetched into registers
    // behind the FMAs.  ZPT > 1: the extra registers would cost a wave per SIMD, and the other resident
    // workgroups already cover the fill latency, so the chunk is loaded in place.
    constexpr bool PF = (ZPT == 1);
    prefetch(0);
    stash();
    __syncthreads();
    for (int c0 = 0; c0 < cin; c0 += CC) {
        const int ccn = min(CC, cin - c0);
        const bool more = (c0 + CC < cin);
        if (PF && more) prefetch(c0 + CC);
        for (int c = 0; c < ccn; ++c) {
            const int wstride = DENSE ? COUT_T : cout;
            const float* __restrict__ wc = wgt + (long)(c0 + c) * 27 * wstride + (DENSE ? 0 : co0);
            const float* tc = tile + c * PLANE + (ty * SXY) * IX + tx * SXY;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float col[IZ];
#pragma unroll
                    for (int z = 0; z < IZ; ++z) col[z] = tc[z * (IY * IX) + ky * IX + kx];
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd) {
                        const float* __restrict__ wk = wc + (kd * 9 + ky * 3 + kx) * wstride;
#pragma unroll
                        for (int z = 0; z < ZPT; ++z)
#pragma unroll
                            for (int co = 0; co < COUT_T; ++co)
                                acc[z][co] = fmaf(col[z * SZ + kd], wk[co], acc[z][co]);
                    }
                }
        }
        if (more) {
            __syncthreads();
            if (!PF) prefetch(c0 + CC);
            stash();
            __syncthreads();
        }
    }

    if (ox >= wo || oy >= ho) return;
    const long out_plane = (long)Do * ho * wo;
#pragma unroll
    for (int z = 0; z < ZPT; ++z) {
        const int oz = oz0 + z;
        if (oz >= Do) break;
        const long o = ((long)oz * ho + oy) * wo + ox;
#pragma unroll
        for (int co = 0; co < COUT_T; ++co) {
            if (co0 + co >= cout) break;
            float v = acc[z][co] + (bias ? bias[co0 + co] : 0.0f);
            if (relu) v = fmaxf(v, 0.0f);
            if (skip) v = v + skip[(long)(co0 + co) * out_plane + o];
            out[(long)(co0 + co) * out_plane + o] = v;
        }
    }
}

template <int COUT_T, int SZ, int SXY, int ZPT, bool DENSE = false>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(SrcSet src, int cin, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias, int cout,
                                                        int D, int h, int w, int Do, int ho, int wo,
                                                        int relu, const float* __restrict__ skip,
                                                        float* __restrict__ out) {
    conv3d_k3_body<COUT_T, SZ, SXY, ZPT, DENSE>(src, cin, wgt, bias, cout, D, h, w, Do, ho, wo, relu, skip, out);
}

// Two independent convolutions of the same shape in one launch: blockIdx.y picks the argument set (the two cross-scale blocks
// CSP_R / CSP_C of a stage, models/Effi_MVS_plus.py:520-531, which would otherwise be forked onto two streams).
struct Conv3dCall {
    SrcSet src;
    const float* wgt;
    const float* bias;
    const float* skip;
    float* out;
};
template <int COUT_T, int SZ, int SXY, int ZPT>
__global__ __launch_bounds__(256) void conv3d_k3_pair_kernel(Conv3dCall a, Conv3dCall b, int cin, int cout, int D, int h, int w,
                                                             int Do, int ho, int wo, int relu) {
    // ONE inlined body behind scalar selects of the arguments (two inlined copies cost up to 3x the registers)
    const bool second = blockIdx.y != 0;
    SrcSet src;
#pragma unroll
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        src.p[i] = second ? b.src.p[i] : a.src.p[i];
        src.ch[i] = second ? b.src.ch[i] : a.src.ch[i];
    }
    conv3d_k3_body<COUT_T, SZ, SXY, ZPT, true>(src, cin, second ? b.wgt : a.wgt, second ? b.bias : a.bias, cout, D, h, w, Do, ho, wo,
                                               relu, second ? b.skip : a.skip, second ? b.out : a.out);
}

// ------------------------------------------------------------------------------------------------
// cin = 1 -> cout = 8, stride (1, SXY, SXY): the first layer of the regulariser (models/module.py:439) and the two entry layers of
// every cross-scale block (conv0 with SXY = 2, conv_cost with SXY = 1; models/module.py:505-506).
// The general body above spends its time around the arithmetic when there is ONE input channel: it fetches and stores its 4-channel
// chunk regardless (3 of 4 loads / LDS writes wasted), every load sits behind its own bounds branch, and the 216 weights of the channel
// do not fit the scalar registers -- the compiler spills them through v_writelane / v_readlane (490 v_readlane for 432 packed FMAs in
// the ISA of round 4).  Here: one plane slot per 256-thread pass (every thread always stores: no branch), all loads of the tile
// issued together from clamped offsets, the weights (27 x 8) + bias read from LDS as broadcast 16-byte quads, ZPT = 4 or 8 output
// planes per thread.  Same FMA order per output as the general body (ky, kx, kd): bitwise the same result.
// blockIdx.y picks the call (two independent convolutions of one shape in a launch, as conv3d_k3_pair_kernel).
// ------------------------------------------------------------------------------------------------
struct C1Call {
    const float* in;
    const float* wgt;       // [27][8]
    const float* bias;      // [8] or nullptr
    float* out;
};
template <int SXY, int ZPT>
__global__ __launch_bounds__(256) void conv3d_c1to8_kernel(C1Call ca, C1Call cb, int D, int h, int w, int ho, int wo, int relu) {
    constexpr int IZ = ZPT + 2, IY = (TY - 1) * SXY + 3, IX = (TX - 1) * SXY + 3;
    constexpr int PSZ = IY * IX, NPL = (PSZ + 255) / 256, PSZP = NPL * 256;          // plane slot padded to whole passes
    __shared__ float tile[IZ * PSZP];
    __shared__ __attribute__((aligned(16))) float wl[27 * 8 + 8];
    const bool second = blockIdx.y != 0;
    const float* __restrict__ in = second ? cb.in : ca.in;
    const float* __restrict__ wgt = second ? cb.wgt : ca.wgt;
    const float* __restrict__ bias = second ? cb.bias : ca.bias;
    float* __restrict__ out = second ? cb.out : ca.out;

    const int tid = threadIdx.x;
    const int tiles_x = (wo + TX - 1) / TX, tiles_xy = tiles_x * ((ho + TY - 1) / TY);
    int lid = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_xy = lid % tiles_xy, zg = lid / tiles_xy;
    const int by = tile_xy / tiles_x, bx = tile_xy - by * tiles_x;
    const int tx = tid % TX, ty = tid / TX;
    const int ox = bx * TX + tx, oy = by * TY + ty, oz0 = zg * ZPT;
    const int iz0 = oz0 - 1, iy0 = by * TY * SXY - 1, ix0 = bx * TX * SXY - 1;

    if (tid < 27 * 8) wl[tid] = wgt[tid];
    if (tid < 8) wl[27 * 8 + tid] = bias ? bias[tid] : 0.0f;

    int poff[NPL];                            // offset of this thread's fill elements inside a z-slice, -1 = padding
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = tid + k * 256;
        const int ly = e / IX, lx = e - ly * IX;
        const int gy = iy0 + ly, gx = ix0 + lx;
        poff[k] = ((e < PSZ) & (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w)) ? gy * w + gx : -1;
    }
    float pf[IZ][NPL];
    unsigned pb[NPL];                                  // in-plane byte offsets; the plane part of an address is a 32-bit scalar (host: D h w < 2^29)
#pragma unroll
    for (int k = 0; k < NPL; ++k) pb[k] = (unsigned)max(poff[k], 0) * 4u;
    const unsigned plane_b = (unsigned)(h * w) * 4u;
#pragma unroll
    for (int lz = 0; lz < IZ; ++lz) {
        const int gz = iz0 + lz;
        const bool zok = (gz >= 0) & (gz < D);
        const unsigned so = (unsigned)min(max(gz, 0), D - 1) * plane_b;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const float t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(in) + (so + pb[k]));
            pf[lz][k] = (zok & (poff[k] >= 0)) ? t : 0.0f;
        }
    }
#pragma unroll
    for (int lz = 0; lz < IZ; ++lz)
#pragma unroll
        for (int k = 0; k < NPL; ++k) tile[lz * PSZP + tid + k * 256] = pf[lz][k];
    __syncthreads();

    // channel pairs as 2-vectors: one v_pk_fma_f32 per (plane, pair) with the input value broadcast to both halves
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    f2 acc[ZPT][4];
#pragma unroll
    for (int z = 0; z < ZPT; ++z)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[z][c] = f2{0.0f, 0.0f};
    const float* tc = tile + (ty * SXY) * IX + tx * SXY;
    // a REAL loop over the nine (ky, kx) columns of taps: fully unrolled, the compiler fetches all 54 weight quads up front (404 registers)
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        float col[IZ];
#pragma unroll
        for (int z = 0; z < IZ; ++z) col[z] = tc[z * PSZP + ky * IX + kx];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const f4 w0 = *reinterpret_cast<const f4*>(&wl[(kd * 9 + t) * 8]);
            const f4 w1 = *reinterpret_cast<const f4*>(&wl[(kd * 9 + t) * 8 + 4]);
            const f2 wk[4] = {f2{w0[0], w0[1]}, f2{w0[2], w0[3]}, f2{w1[0], w1[1]}, f2{w1[2], w1[3]}};
#pragma unroll
            for (int z = 0; z < ZPT; ++z) {
                const f2 cv = {col[z + kd], col[z + kd]};
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[z][c] = __builtin_elementwise_fma(cv, wk[c], acc[z][c]);
            }
        }
    }

    if (ox >= wo || oy >= ho) return;
    const long out_plane = (long)D * ho * wo;
    float bv[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) bv[co] = wl[27 * 8 + co];
#pragma unroll
    for (int z = 0; z < ZPT; ++z) {
        const int oz = oz0 + z;
        if (oz >= D) break;
        const long o = ((long)oz * ho + oy) * wo + ox;
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            float v = acc[z][co >> 1][co & 1] + bv[co];
            if (relu) v = fmaxf(v, 0.0f);
            out[(long)co * out_plane + o] = v;
        }
    }
}

// option c3_lean: unset = both dedicated kernels; otherwise a mask, bit 0 = the 1 -> 8 kernel, bit 1 = the 8 -> 1 kernel (A/B, bisection)
static bool c3_lean(int bit) {
    const long v = effi_option(EFFI_OPT_C3_LEAN);
    return v == EFFI_OPT_UNSET || ((v >> bit) & 1);
}

// launch rule of the cin = 1 kernel: 8 planes per thread when the grid still covers the chip twice
int launch_c1to8(const C1Call& a, const C1Call* b, int D, int h, int w, int sxy, int relu, hipStream_t st) {
    if ((long)D * h * w >= (1L << 29)) return EFFI_ERR_UNSUPPORTED;              // 32-bit byte offsets inside the input volume
    const int ho = (h - 1) / sxy + 1, wo = (w - 1) / sxy + 1;
    const long tiles = (long)effi_cdiv(wo, TX) * effi_cdiv(ho, TY), ncall = b ? 2 : 1;
    const bool z8 = tiles * effi_cdiv(D, 8) * ncall >= 512;
    const dim3 grid((unsigned)(tiles * effi_cdiv(D, z8 ? 8 : 4)), (unsigned)ncall);
    const C1Call& bb = b ? *b : a;
    if (sxy == 1) {
        if (z8) hipLaunchKernelGGL((conv3d_c1to8_kernel<1, 8>), grid, dim3(256), 0, st, a, bb, D, h, w, ho, wo, relu);
        else hipLaunchKernelGGL((conv3d_c1to8_kernel<1, 4>), grid, dim3(256), 0, st, a, bb, D, h, w, ho, wo, relu);
    } else {
        if (z8) hipLaunchKernelGGL((conv3d_c1to8_kernel<2, 8>), grid, dim3(256), 0, st, a, bb, D, h, w, ho, wo, relu);
        else hipLaunchKernelGGL((conv3d_c1to8_kernel<2, 4>), grid, dim3(256), 0, st, a, bb, D, h, w, ho, wo, relu);
    }
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// cin = 8 -> cout = 1: the regulariser's last layer (stride 1, models/module.py:451 `prob`) and the transposed conv2 of every
// cross-scale block (stride (1,2,2), models/module.py:508).  Same recipe as conv3d_c1to8_kernel: the tile of a 4-channel chunk is
// fetched with all its loads in flight at once (clamped offsets, every thread stores, plane slots padded to whole half-passes), the
// 8 x 27 weights sit in LDS and are read as broadcast quads inside a REAL loop over the chunk's channels (so that at most one
// channel's weights are live), FMA order per output as in the general bodies (channel, then the bodies' tap order): bitwise equal.
// ------------------------------------------------------------------------------------------------
struct C8Call {
    const float* in;        // [8][D][h][w]
    const float* wgt;       // [8][27]
    const float* bias;      // [1] or nullptr
    float* out;
};
// DECONV = false: out[z][y][x] = sum_c sum_taps in[c][z+kd-1][y+ky-1][x+kx-1] w[c][kd][ky][kx]          (ZPT = 4 planes per thread)
// DECONV = true : transposed, stride (1,2,2), output_padding (0,1,1): a thread owns input (y, x) and the 2 x 2 outputs it maps to
template <bool DECONV, int ZPT>
__global__ __launch_bounds__(256) void conv3d_c8to1_kernel(C8Call ca, C8Call cb, int D, int h, int w, int relu) {
#ifndef EFFI_C8_CC
#define EFFI_C8_CC 2
#endif
    constexpr int IZ = ZPT + 2, CC = (ZPT == 8) ? 1 : EFFI_C8_CC;           // channels per chunk (in flight as registers during the multiplies)
    constexpr int IY = DECONV ? TY + 1 : TY + 2, IX = DECONV ? TX + 1 : TX + 2;
#ifndef EFFI_C8_PADMASK
#define EFFI_C8_PADMASK 127
#endif
    constexpr int PSZ = IY * IX, PSZP = (PSZ + EFFI_C8_PADMASK) & ~EFFI_C8_PADMASK, NPL = (PSZP + 255) / 256, LASTN = PSZP - 256 * (NPL - 1);
    __shared__ float tile[CC * IZ * PSZP];
    __shared__ __attribute__((aligned(16))) float wl[8 * 28 + 4];          // 27 weights per channel at stride 28, then the bias
    const bool second = blockIdx.y != 0;
    const float* __restrict__ in = second ? cb.in : ca.in;
    const float* __restrict__ wgt = second ? cb.wgt : ca.wgt;
    const float* __restrict__ bias = second ? cb.bias : ca.bias;
    float* __restrict__ out = second ? cb.out : ca.out;

    const int tid = threadIdx.x;
    const int tiles_x = (w + TX - 1) / TX, tiles_xy = tiles_x * ((h + TY - 1) / TY);
    int lid = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_xy = lid % tiles_xy, zg = lid / tiles_xy;
    const int by = tile_xy / tiles_x, bx = tile_xy - by * tiles_x;
    const int tx = tid % TX, ty = tid / TX;
    const int ix = bx * TX + tx, iy = by * TY + ty, z0 = zg * ZPT;
    const int iz0 = z0 - 1, iy0 = DECONV ? by * TY : by * TY - 1, ix0 = DECONV ? bx * TX : bx * TX - 1;
    const long in_plane = (long)D * h * w;

    if (tid < 8 * 27) wl[(tid / 27) * 28 + tid % 27] = wgt[tid];
    if (tid < 8) wl[tid * 28 + 27] = 0.0f;                                // the pad entry of a channel's seven quads
    if (tid == 0) wl[8 * 28] = bias ? bias[0] : 0.0f;

    int poff[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = tid + k * 256;
        const int ly = e / IX, lx = e - ly * IX;
        const int gy = iy0 + ly, gx = ix0 + lx;
        poff[k] = ((e < PSZ) & (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w)) ? gy * w + gx : -1;
    }
    const bool last_pass = tid < LASTN;                                  // wave-uniform (LASTN is a multiple of 128)

    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int NACC = DECONV ? ZPT * 4 : ZPT;
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;

    // the chunk's raw loads (converted to zero padding when stored): the second chunk is in flight while the first is multiplied
    float pf[CC][IZ][NPL];
    // address = the tensor's base (scalar) + a 32-bit byte offset: (channel, plane) part scalar, in-plane part per thread -- one
    // add per load, no 64-bit address pairs held across the multiply phase (the host checks 8 D h w < 2^29)
    unsigned pb[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) pb[k] = (unsigned)max(poff[k], 0) * 4u;
    const unsigned plane_b = (unsigned)(h * w) * 4u, chan_b = (unsigned)in_plane * 4u;
    auto fetch = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz) {
                const int gz = min(max(iz0 + lz, 0), D - 1);                 // planes outside the volume: any valid plane (zeroed when stored)
                const unsigned so = (unsigned)(c0 + c) * chan_b + (unsigned)gz * plane_b;
#pragma unroll
                for (int k = 0; k < NPL; ++k)
                    pf[c][lz][k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(in) + (so + pb[k]));
            }
        }
    };
#ifndef EFFI_C8_PREFETCH
#define EFFI_C8_PREFETCH 1
#endif
    if (EFFI_C8_PREFETCH) fetch(0);
#ifdef EFFI_C8_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int c0 = 0; c0 < 8; c0 += CC) {
        if (!EFFI_C8_PREFETCH) fetch(c0);
        if (c0) __syncthreads();                                         // the previous chunk's reads are done
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz) {
                const int gz = iz0 + lz;
                const bool zok = (gz >= 0) & (gz < D);
#pragma unroll
                for (int k = 0; k < NPL; ++k)
                    if (k + 1 < NPL || last_pass) tile[(c * IZ + lz) * PSZP + tid + k * 256] = (zok & (poff[k] >= 0)) ? pf[c][lz][k] : 0.0f;
            }
        __syncthreads();
        if (EFFI_C8_PREFETCH && c0 + CC < 8) fetch(c0 + CC);
#pragma unroll 1
        for (int c = 0; c < CC; ++c) {
            float wk[28];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const f4 t = *reinterpret_cast<const f4*>(&wl[(c0 + c) * 28 + 4 * q]);
                wk[4 * q] = t[0]; wk[4 * q + 1] = t[1]; wk[4 * q + 2] = t[2]; wk[4 * q + 3] = t[3];
            }
            const float* tc = tile + c * IZ * PSZP + ty * IX + tx;
#ifdef EFFI_C8_NO_B96
            acc[0] = fmaf(wk[27], 0.0f, acc[0]);                         // keeps the seventh quad a whole 16-byte read
#endif
            if constexpr (!DECONV) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float col[IZ];
#pragma unroll
                        for (int z = 0; z < IZ; ++z) col[z] = tc[z * PSZP + ky * IX + kx];
#pragma unroll
                        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                            for (int z = 0; z < ZPT; ++z) acc[z] = fmaf(col[z + kd], wk[kd * 9 + ky * 3 + kx], acc[z]);
                    }
            } else {
#pragma unroll
                for (int lz = 0; lz < IZ; ++lz)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            const float v = tc[lz * PSZP + dy * IX + dx];
#pragma unroll
                            for (int ay = 0; ay < 2; ++ay) {
                                const int ky = (dy == 0) ? (ay == 0 ? 1 : 2) : (ay == 1 ? 0 : -1);
                                if (ky < 0) continue;
#pragma unroll
                                for (int ax = 0; ax < 2; ++ax) {
                                    const int kx = (dx == 0) ? (ax == 0 ? 1 : 2) : (ax == 1 ? 0 : -1);
                                    if (kx < 0) continue;
#pragma unroll
                                    for (int az = 0; az < ZPT; ++az) {
                                        const int kd = az + 2 - lz;          // lz = az + 2 - kd
                                        if (kd < 0 || kd > 2) continue;
                                        acc[(az * 2 + ay) * 2 + ax] = fmaf(v, wk[kd * 9 + ky * 3 + kx], acc[(az * 2 + ay) * 2 + ax]);
                                    }
                                }
                            }
                        }
            }
        }
    }

    if (ix >= w || iy >= h) return;
    const float b = wl[8 * 28];
    if constexpr (!DECONV) {
#pragma unroll
        for (int z = 0; z < ZPT; ++z) {
            const int oz = z0 + z;
            if (oz >= D) break;
            float v = acc[z] + b;
            if (relu) v = fmaxf(v, 0.0f);
            out[((long)oz * h + iy) * w + ix] = v;
        }
    } else {
        const int ho = 2 * h, wo = 2 * w;
#pragma unroll
        for (int az = 0; az < ZPT; ++az) {
            const int oz = z0 + az;
            if (oz >= D) break;
#pragma unroll
            for (int ay = 0; ay < 2; ++ay) {
                float v0 = acc[(az * 2 + ay) * 2] + b, v1 = acc[(az * 2 + ay) * 2 + 1] + b;
                if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
                *reinterpret_cast<float2*>(out + ((long)oz * ho + 2 * iy + ay) * wo + 2 * ix) = make_float2(v0, v1);
            }
        }
    }
}

template <bool DECONV>
int launch_c8to1(const C8Call& a, const C8Call* b, int D, int h, int w, int relu, hipStream_t st) {
    if (8L * D * h * w >= (1L << 29)) return EFFI_ERR_UNSUPPORTED;              // 32-bit byte offsets inside the input tensor
    const long tiles = (long)effi_cdiv(w, TX) * effi_cdiv(h, TY), ncall = b ? 2 : 1;
    const bool z8 = tiles * effi_cdiv(D, 8) * ncall >= 512;
    const dim3 grid((unsigned)(tiles * effi_cdiv(D, z8 ? 8 : 4)), (unsigned)ncall);
    if (z8) hipLaunchKernelGGL((conv3d_c8to1_kernel<DECONV, 8>), grid, dim3(256), 0, st, a, b ? *b : a, D, h, w, relu);
    else hipLaunchKernelGGL((conv3d_c8to1_kernel<DECONV, 4>), grid, dim3(256), 0, st, a, b ? *b : a, D, h, w, relu);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// transposed convolution, stride (SZ, 2, 2), padding 1, output_padding (SZ-1, 1, 1).
// One thread per INPUT position: it produces the 2x2 (x SZ) output block that position owns, so every
// lane does identical work (no parity divergence) and each of the 27 taps is used exactly once.
// stride 2 along an axis:  out[2i]   = in[i]   * w[1]
//                          out[2i+1] = in[i]   * w[2] + in[i+1] * w[0]
// stride 1 along z:        out[z]    = sum_kd in[z + 1 - kd] * w[kd]
// ------------------------------------------------------------------------------------------------
template <int COUT_T, int SZ>
__device__ __forceinline__ void deconv3d_k3_body(const float* __restrict__ in, int cin,
                                                 const float* __restrict__ wgt,
                                                 const float* __restrict__ bias, int cout,
                                                 int D, int h, int w, int relu,
                                                 const float* __restrict__ skip, float* __restrict__ out) {
    constexpr int ZPT = (SZ == 2) ? 1 : 4;            // input z positions per thread
    constexpr int OZ = (SZ == 2) ? 2 : ZPT;           // output z positions per thread
    constexpr int CC = 4;
    constexpr int IZ = (SZ == 2) ? 2 : ZPT + 2, IY = TY + 1, IX = TX + 1;
    constexpr int PLANE = IZ * IY * IX;
    constexpr int PSZ = IY * IX, NPL = (PSZ + 255) / 256;
    __shared__ float tile[CC * PLANE];

    const int tiles_x = (w + TX - 1) / TX, tiles_xy = tiles_x * ((h + TY - 1) / TY), nzg = (D + ZPT - 1) / ZPT;
    int lid = effi_xcd_remap(blockIdx.x, gridDim.x);       // XCD-aware 1-D grid, as in conv3d_k3_kernel
    const int tile_xy = lid % tiles_xy;
    lid /= tiles_xy;
    const int zg = lid % nzg, cg = lid / nzg;
    const int by = tile_xy / tiles_x, bx = tile_xy - by * tiles_x;
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int ix = bx * TX + tx, iy = by * TY + ty, z0 = zg * ZPT;
    const int co0 = cg * COUT_T;
    const int iz0 = (SZ == 2) ? z0 : z0 - 1;
    const long in_plane = (long)D * h * w;

    int poff[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = threadIdx.x + k * 256;
        const int ly = e / IX, lx = e - ly * IX;
        const int gy = by * TY + ly, gx = bx * TX + lx;
        poff[k] = ((e < PSZ) & (gy < h) & (gx < w)) ? gy * w + gx : -1;
    }

    float acc[OZ][2][2][COUT_T];
#pragma unroll
    for (int a = 0; a < OZ; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int d = 0; d < COUT_T; ++d) acc[a][b][c][d] = 0.0f;

    float pf[CC][IZ][NPL];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const bool cok = (c0 + c < cin);
            const float* __restrict__ cp = in + (long)(cok ? c0 + c : 0) * in_plane;
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz) {
                const int gz = iz0 + lz;
                const bool zok = cok & (gz >= 0) & (gz < D);
                const float* __restrict__ zp = cp + (zok ? (long)gz * h * w : 0);
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const float t = zp[max(poff[k], 0)];
                    pf[c][lz][k] = (zok & (poff[k] >= 0)) ? t : 0.0f;
                }
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz)
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const int e = threadIdx.x + k * 256;
                    if (e < PSZ) tile[c * PLANE + lz * PSZ + e] = pf[c][lz][k];
                }
    };

    prefetch(0);
    stash();
    __syncthreads();
    for (int c0 = 0; c0 < cin; c0 += CC) {
        const int ccn = min(CC, cin - c0);
        const bool more = (c0 + CC < cin);
        if (more) prefetch(c0 + CC);
        for (int c = 0; c < ccn; ++c) {
            const int wstride = (COUT_T == 1) ? 1 : cout;       // the single-channel instance is only launched with cout == 1
            const float* __restrict__ wc = wgt + (long)(c0 + c) * 27 * wstride + co0;
            const float* tc = tile + c * PLANE + ty * IX + tx;
#pragma unroll
            for (int lz = 0; lz < IZ; ++lz)
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx) {
                        const float v = tc[lz * (IY * IX) + dy * IX + dx];
#pragma unroll
                        for (int ay = 0; ay < 2; ++ay) {
                            const int ky = (dy == 0) ? (ay == 0 ? 1 : 2) : (ay == 1 ? 0 : -1);
                            if (ky < 0) continue;
#pragma unroll
                            for (int ax = 0; ax < 2; ++ax) {
                                const int kx = (dx == 0) ? (ax == 0 ? 1 : 2) : (ax == 1 ? 0 : -1);
                                if (kx < 0) continue;
#pragma unroll
                                for (int az = 0; az < OZ; ++az) {
                                    int kd;
                                    if (SZ == 2) kd = (lz == 0) ? (az == 0 ? 1 : 2) : (az == 1 ? 0 : -1);
                                    else kd = az + 2 - lz;          // lz = az + 2 - kd
                                    if (kd < 0 || kd > 2) continue;
                                    const float* __restrict__ wk = wc + (kd * 9 + ky * 3 + kx) * wstride;
#pragma unroll
                                    for (int co = 0; co < COUT_T; ++co)
                                        acc[az][ay][ax][co] = fmaf(v, wk[co], acc[az][ay][ax][co]);
                                }
                            }
                        }
                    }
        }
        if (more) {
            __syncthreads();
            stash();
            __syncthreads();
        }
    }

    if (ix >= w || iy >= h) return;
    const int Do = SZ * D, ho = 2 * h, wo = 2 * w;
    const long out_plane = (long)Do * ho * wo;
#pragma unroll
    for (int az = 0; az < OZ; ++az) {
        const int oz = (SZ == 2) ? 2 * z0 + az : z0 + az;
        if (oz >= Do) break;
#pragma unroll
        for (int ay = 0; ay < 2; ++ay) {
            const long o = ((long)oz * ho + 2 * iy + ay) * wo + 2 * ix;
#pragma unroll
            for (int co = 0; co < COUT_T; ++co) {
                if (co0 + co >= cout) break;
                const float b = bias ? bias[co0 + co] : 0.0f;
                float v0 = acc[az][ay][0][co] + b, v1 = acc[az][ay][1][co] + b;
                if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
                const long oo = (long)(co0 + co) * out_plane + o;
                if (skip) {
                    const float2 s = *reinterpret_cast<const float2*>(skip + oo);
                    v0 = v0 + s.x;
                    v1 = v1 + s.y;
                }
                *reinterpret_cast<float2*>(out + oo) = make_float2(v0, v1);
            }
        }
    }
}

template <int COUT_T, int SZ>
__global__ __launch_bounds__(256) void deconv3d_k3_kernel(const float* __restrict__ in, int cin,
                                                          const float* __restrict__ wgt,
                                                          const float* __restrict__ bias, int cout,
                                                          int D, int h, int w, int relu,
                                                          const float* __restrict__ skip, float* __restrict__ out) {
    deconv3d_k3_body<COUT_T, SZ>(in, cin, wgt, bias, cout, D, h, w, relu, skip, out);
}

struct Deconv3dCall {
    const float* in;
    const float* wgt;
    const float* bias;
    const float* skip;
    float* out;
};
template <int COUT_T, int SZ>
__global__ __launch_bounds__(256) void deconv3d_k3_pair_kernel(Deconv3dCall a, Deconv3dCall b, int cin, int cout, int D, int h,
                                                               int w, int relu) {
    const bool second = blockIdx.y != 0;
    deconv3d_k3_body<COUT_T, SZ>(second ? b.in : a.in, cin, second ? b.wgt : a.wgt, second ? b.bias : a.bias, cout, D, h, w, relu,
                                 second ? b.skip : a.skip, second ? b.out : a.out);
}

template <int COUT_T, int SZ, int SXY, int ZPT>
int launch_conv_z(const SrcSet& s, int cin, const float* wgt, const float* bias, int cout, int D, int h, int w,
                  int relu, const float* skip, float* out, hipStream_t st) {
    const int Do = (D - 1) / SZ + 1, ho = (h - 1) / SXY + 1, wo = (w - 1) / SXY + 1;
    dim3 grid(effi_cdiv(wo, TX) * effi_cdiv(ho, TY) * effi_cdiv(Do, ZPT) * effi_cdiv(cout, COUT_T));
    if (cout == COUT_T)
        hipLaunchKernelGGL((conv3d_k3_kernel<COUT_T, SZ, SXY, ZPT, true>), grid, dim3(256), 0, st, s, cin, wgt, bias, cout, D, h, w,
                           Do, ho, wo, relu, skip, out);
    else
        hipLaunchKernelGGL((conv3d_k3_kernel<COUT_T, SZ, SXY, ZPT, false>), grid, dim3(256), 0, st, s, cin, wgt, bias, cout, D, h, w,
                           Do, ho, wo, relu, skip, out);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

// Output z-slices per thread: 4 (2 for stride 2) amortises the z halo of the LDS tile; low-resolution layers
// fall back to 1 so that the grid still has >= 2 workgroups per CU.
template <int COUT_T, int SZ, int SXY>
int launch_conv(const SrcSet& s, int cin, const float* wgt, const float* bias, int cout, int D, int h, int w,
                int relu, const float* skip, float* out, hipStream_t st) {
    constexpr int ZBIG = (SXY == 1) ? 4 : ((SZ == 1) ? 4 : 2);
    const int Do = (D - 1) / SZ + 1, ho = (h - 1) / SXY + 1, wo = (w - 1) / SXY + 1;
    const long blocks = (long)effi_cdiv(wo, TX) * effi_cdiv(ho, TY) * effi_cdiv(Do, ZBIG) * effi_cdiv(cout, COUT_T);
    if (blocks >= 384) return launch_conv_z<COUT_T, SZ, SXY, ZBIG>(s, cin, wgt, bias, cout, D, h, w, relu, skip, out, st);
    return launch_conv_z<COUT_T, SZ, SXY, 1>(s, cin, wgt, bias, cout, D, h, w, relu, skip, out, st);
}

}  // namespace

extern "C" int effi_conv3d_k3_f32(const float* const* srcs, const int* src_channels, int n_src, const float* weight,
                                  const float* bias, int cout, int D, int h, int w, int sz, int sxy, int relu,
                                  const float* skip, float* out, effi_stream_t stream) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !weight || !out) return EFFI_ERR_BADARG;
    if (D < 1 || h < 1 || w < 1 || cout < 1) return EFFI_ERR_BADARG;
    SrcSet s;
    int cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        s.p[i] = (i < n_src) ? srcs[i] : nullptr;
        s.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        cin += s.ch[i];
    }
    hipStream_t st = effi_s(stream);
    if (cin == 1 && cout == 8 && sz == 1 && (sxy == 1 || sxy == 2) && !skip && c3_lean(0)) {
        const C1Call c{srcs[0], weight, bias, out};
        return launch_c1to8(c, nullptr, D, h, w, sxy, relu, st);
    }
    if (cin == 8 && n_src == 1 && cout == 1 && sz == 1 && sxy == 1 && !skip && c3_lean(1)) {
        const C8Call c{srcs[0], weight, bias, out};
        return launch_c8to1<false>(c, nullptr, D, h, w, relu, st);
    }
    const bool c8 = (cout % 8 == 0);
    if (!c8 && cout != 1) return EFFI_ERR_UNSUPPORTED;
    if (sz == 1 && sxy == 1)
        return c8 ? launch_conv<8, 1, 1>(s, cin, weight, bias, cout, D, h, w, relu, skip, out, st)
                  : launch_conv<1, 1, 1>(s, cin, weight, bias, cout, D, h, w, relu, skip, out, st);
    if (!c8) return EFFI_ERR_UNSUPPORTED;
    if (sz == 2 && sxy == 2) return launch_conv<8, 2, 2>(s, cin, weight, bias, cout, D, h, w, relu, skip, out, st);
    if (sz == 1 && sxy == 2) return launch_conv<8, 1, 2>(s, cin, weight, bias, cout, D, h, w, relu, skip, out, st);
    return EFFI_ERR_UNSUPPORTED;
}

extern "C" int effi_deconv3d_k3_f32(const float* in, int cin, const float* weight, const float* bias, int cout, int D,
                                    int h, int w, int sz, int relu, const float* skip, float* out,
                                    effi_stream_t stream) {
    if (!in || !weight || !out || cin < 1 || cout < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    const int tiles = effi_cdiv(w, TX) * effi_cdiv(h, TY);
    if (sz == 2 && cout % 8 == 0) {
        if ((long)tiles * D * (cout / 8) >= 512)
            hipLaunchKernelGGL((deconv3d_k3_kernel<8, 2>), dim3(tiles * D * (cout / 8)), dim3(256), 0, st, in, cin, weight, bias,
                               cout, D, h, w, relu, skip, out);
        else   // low-resolution level: split the output channels finer so the grid covers the chip
            hipLaunchKernelGGL((deconv3d_k3_kernel<4, 2>), dim3(tiles * D * (cout / 4)), dim3(256), 0, st, in, cin, weight, bias,
                               cout, D, h, w, relu, skip, out);
    } else if (sz == 1 && cout == 1 && cin == 8 && !skip && c3_lean(1)) {
        const C8Call c{in, weight, bias, out};
        return launch_c8to1<true>(c, nullptr, D, h, w, relu, st);
    } else if (sz == 1 && cout == 1) {
        hipLaunchKernelGGL((deconv3d_k3_kernel<1, 1>), dim3(tiles * effi_cdiv(D, 4)), dim3(256), 0, st, in, cin, weight,
                           bias, cout, D, h, w, relu, skip, out);
    } else {
        return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

// ---- pair launches (see conv3d_k3_pair_kernel) ------------------------------------------------------------------------
namespace {
template <int SXY, int ZPT>
int launch_conv_pair(const Conv3dCall& a, const Conv3dCall& b, int cin, int D, int h, int w, int relu, hipStream_t st) {
    const int ho = (h - 1) / SXY + 1, wo = (w - 1) / SXY + 1;
    const dim3 grid(effi_cdiv(wo, TX) * effi_cdiv(ho, TY) * effi_cdiv(D, ZPT), 2);
    hipLaunchKernelGGL((conv3d_k3_pair_kernel<8, 1, SXY, ZPT>), grid, dim3(256), 0, st, a, b, cin, 8, D, h, w, D, ho, wo, relu);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}
}  // namespace

extern "C" int effi_conv3d_k3_pair_f32(const float* in_a, const float* weight_a, const float* bias_a, float* out_a,
                                       const float* in_b, const float* weight_b, const float* bias_b, float* out_b, int cin,
                                       int cout, int D, int h, int w, int sxy, int relu, effi_stream_t stream) {
    if (!in_a || !weight_a || !out_a || !in_b || !weight_b || !out_b || cin < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (cout != 8 || (sxy != 1 && sxy != 2)) return EFFI_ERR_UNSUPPORTED;
    Conv3dCall a, b;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src.p[i] = (i == 0) ? in_a : nullptr;
        b.src.p[i] = (i == 0) ? in_b : nullptr;
        a.src.ch[i] = b.src.ch[i] = (i == 0) ? cin : 0;
    }
    a.wgt = weight_a; a.bias = bias_a; a.skip = nullptr; a.out = out_a;
    b.wgt = weight_b; b.bias = bias_b; b.skip = nullptr; b.out = out_b;
    hipStream_t st = effi_s(stream);
    if (cin == 1 && c3_lean(0)) {
        const C1Call ca{in_a, weight_a, bias_a, out_a}, cb{in_b, weight_b, bias_b, out_b};
        return launch_c1to8(ca, &cb, D, h, w, sxy, relu, st);
    }
    const int ho = (h - 1) / sxy + 1, wo = (w - 1) / sxy + 1;
    const bool big = 2L * effi_cdiv(wo, TX) * effi_cdiv(ho, TY) * effi_cdiv(D, 4) >= 384;     // as launch_conv, for both grids
    if (sxy == 1) return big ? launch_conv_pair<1, 4>(a, b, cin, D, h, w, relu, st) : launch_conv_pair<1, 1>(a, b, cin, D, h, w, relu, st);
    return big ? launch_conv_pair<2, 4>(a, b, cin, D, h, w, relu, st) : launch_conv_pair<2, 1>(a, b, cin, D, h, w, relu, st);
}

extern "C" int effi_deconv3d_k3_pair_f32(const float* in_a, const float* weight_a, const float* bias_a, float* out_a,
                                         const float* in_b, const float* weight_b, const float* bias_b, float* out_b, int cin,
                                         int cout, int D, int h, int w, int sz, int relu, effi_stream_t stream) {
    if (!in_a || !weight_a || !out_a || !in_b || !weight_b || !out_b || cin < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (sz != 1 || cout != 1) return EFFI_ERR_UNSUPPORTED;
    if (cin == 8 && c3_lean(1)) {
        const C8Call ca{in_a, weight_a, bias_a, out_a}, cb{in_b, weight_b, bias_b, out_b};
        return launch_c8to1<true>(ca, &cb, D, h, w, relu, effi_s(stream));
    }
    const Deconv3dCall a{in_a, weight_a, bias_a, nullptr, out_a}, b{in_b, weight_b, bias_b, nullptr, out_b};
    const dim3 grid(effi_cdiv(w, TX) * effi_cdiv(h, TY) * effi_cdiv(D, 4), 2);
    hipLaunchKernelGGL((deconv3d_k3_pair_kernel<1, 1>), grid, dim3(256), 0, effi_s(stream), a, b, cin, cout, D, h, w, relu);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
