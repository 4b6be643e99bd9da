// ConvGRU (models/update.py:33-49) as ONE kernel on split-resident maps: z | r = sigmoid(convzr([h, x])), q = tanh(convq([r * h, x])),
// h' = (1 - z) h + z q.  Included by conv2d_sr.hip (compiled twice: split precision, and hi-only with -DEFFI_BF16_ONLY).
//
// Why: at 592x800 the two launches (z | r convolution, then q convolution + update) are bound by what they move, not by what they
// compute -- 320 + 384 bytes per pixel at hd = 16 (reads of h / x / r * h as split-resident maps, h and z as fp32 maps, writes of z,
// r * h, h' twice) at an effective 4-6.7 TB/s (MALL + HBM), matrix pipe 26 % busy.  r * h and z exist only to carry a value from the
// first launch to the second.  Here a workgroup evaluates z and r on a 16 x 16 lattice of pixels, keeps z in registers, writes r * h
// (split into hi / lo exactly as effi_sr_store4 does) into an LDS image, evaluates q on the SAME lattice from that image, and stores
// the 14 x 14 interior (a lattice point on the rim lacks a neighbour of r * h).  Per output pixel: 1.65 x (h, x) read once (18 x 18
// staged for 14 x 14 outputs), h fp32 read, h' written twice = ~400 bytes instead of 704, for 1.31 x the matrix work.
//
// MEASURED (592x800, hd 16, four launches per view): 62 us per launch against 36 + 27 us for the two launches it replaces -- no gain,
// although it moves 43 % fewer bytes; at hd 32 (one workgroup per CU: 109 KB of LDS) 69 against 50 us.  So the pair is bound neither
// by bandwidth nor (a variant with every load issued up front was slower still) by memory latency; what remains is the
// LDS-fragment / MFMA dependency chain at two workgroups per CU.  The kernel stays as an option (``gru_fused``, default 0) with its
// bitwise test; the default path keeps the two launches.
//
// The arithmetic per pixel is that of the two launches operation for operation (same chunk / K-step order, same split of r * h, same
// update expression): the result is BITWISE equal (tests/test_gpu_sr.py).
//
// A lattice point needs h (fp32, for r * h and the update) of pixels that belong to the neighbouring workgroups' output tiles, so
// the state is NOT updated in place: h_in / h_out and H_in / H_out are different buffers (the caller ping-pongs).
#pragma once
#include "conv2d_x3.hpp"

namespace {

struct GruFusedArgs {
    const unsigned short* H_in; const unsigned short* X;      // split-resident maps, hd channels each
    const float* h_in;                                       // fp32 [hd][h][w]
    const unsigned short* wzr; const float* bzr;             // convz | convr packed as ONE 3x3 layer, cin = 2 hd ([h, x]), cout = 2 hd
    const unsigned short* wq; const float* bq;               // convq, cin = 2 hd ([r * h, x]), cout = hd
    float* h_out; unsigned short* H_out;
    int h, w, hp, wp;
};

// NH = hd / 16.  256 threads: wave wv owns lattice rows 4 wv .. 4 wv + 3, a lane (li, lk) column li, channels 4 lk .. 4 lk + 3 of
// every 16-channel tile (the transposed MFMA layout of conv2d_k3_bf16x3_tile).
template <int NH>
__global__ __launch_bounds__(256) void gru_zr_q_fused_kernel(const GruFusedArgs g, int tiles_x, int ntiles) {
    constexpr int MR = 4, NTHR = 256, OT = 14, AW = 18, APIX = 18 * 18, APIXP = (APIX + 15) & ~15, NKS = 5;
    constexpr int NT1 = 2 * NH, NT2 = NH, NCH = 2 * NH;                  // N-tiles of the two layers; 16-channel chunks of either input
    constexpr int NPART = kHiOnly ? 1 : 2;
    constexpr int UNITS = 2 * 2 * APIXP;                                 // 16-byte units of one chunk image [part][octet][pixel]
    constexpr int NUA = (NPART * 2 * APIXP + NTHR - 1) / NTHR;
    constexpr int NBF1 = NKS * NT1 * 2 * 64, NBF2 = NKS * NT2 * 2 * 64;  // units of B per chunk
    constexpr int NB1 = (NBF1 + NTHR - 1) / NTHR, NB2 = (NBF2 + NTHR - 1) / NTHR;
    __shared__ __attribute__((aligned(16))) unsigned short lds_a[(NUA * NTHR > UNITS ? NUA * NTHR : UNITS) * 8];
    __shared__ __attribute__((aligned(16))) unsigned short lds_rh[NH][UNITS * 8];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[NB1 * NTHR * 8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    const int h = g.h, w = g.w, hp = g.hp, wp = g.wp;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * OT, y0 = ty_ * OT;           // output tile; lattice origin (y0 - 1, x0 - 1); staged (y0 - 2, x0 - 2)
    const long plane = (long)hp * wp;

    // staging: unit tid + 256 j of a chunk image <- its 16-byte unit of the map (map row = y + 1, column = x + 1); what lies
    // outside the map's planes reads unit 0 of the chunk (a corner of the zero border)
    int goff[NUA];
#pragma unroll
    for (int j = 0; j < NUA; ++j) {
        const int u = tid + j * NTHR;
        const int part = u / (2 * APIXP), r_ = u - part * (2 * APIXP);
        const int oct = r_ / APIXP, p = r_ - oct * APIXP;
        const int row = p / AW, col = p - row * AW;
        const int my = y0 - 1 + row, mx = x0 - 1 + col;
        const bool valid = (part < NPART) & (p < APIX) & (my >= 0) & (my < hp) & (mx >= 0) & (mx < wp);
        goff[j] = valid ? (int)(((long)(oct * 2 + part) * hp + my) * wp + mx) : 0;
    }
    // step k of the kernel = one 16-channel chunk of a layer: k < NCH: chunk k of [h, x] for z | r; k >= NCH: chunk k - NCH of
    // [r * h, x] for q (the r * h chunks come from the LDS image, only their weights are staged).  The operands of step k + 1 are
    // fetched into registers while step k runs on the matrix cores.
    f32x4 pa[NUA], tb[NB1];
    auto needs_a = [&](int k) { return k < NCH || k - NCH >= NH; };
    auto prefetch = [&](int k) {
        const bool l2 = k >= NCH;
        const int c = l2 ? k - NCH : k;
        if (needs_a(k)) {
            const unsigned short* src = (!l2 && c < NH) ? g.H_in : g.X;
            const f32x4* base = reinterpret_cast<const f32x4*>(src) + (long)(c < NH ? c : c - NH) * 4 * plane;
#pragma unroll
            for (int j = 0; j < NUA; ++j) pa[j] = base[goff[j]];
        }
        const unsigned short* wbf = l2 ? g.wq : g.wzr;
        const int nbf = l2 ? NBF2 : NBF1;
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            if (l2 && j >= NB2) continue;
            const int u = min(tid + j * NTHR, nbf - 1);
            tb[j] = *reinterpret_cast<const f32x4*>(wbf + ((long)c * nbf + u) * 8);
        }
    };
    auto stash = [&](int k) {
        if (needs_a(k)) {
#pragma unroll
            for (int j = 0; j < NUA; ++j) *reinterpret_cast<f32x4*>(&lds_a[(tid + j * NTHR) * 8]) = pa[j];
        }
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            if (k >= NCH && j >= NB2) continue;
            *reinterpret_cast<f32x4*>(&lds_b[(tid + j * NTHR) * 8]) = tb[j];
        }
    };
    int koff[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int item = 4 * s_ + lk;
        const int tap = min(item >> 1, 8), oct = item & 1;               // items 18, 19 are padding (B is zero there)
        koff[s_] = ((wv * MR + tap / 3) * AW + li + tap % 3 + oct * APIXP) * 8;
    }
    // one 16-channel chunk of a 3x3 layer: acc[m][n] += W[chunk] x image, the loop of conv2d_k3_bf16x3_tile
    auto mma = [&](auto nt_c, f32x4 (&acc)[MR][decltype(nt_c)::value], const unsigned short* img) {
        constexpr int NT = decltype(nt_c)::value;
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            bf16x8 ah[MR], al[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(&img[koff[s_] + m * AW * 8]);
                if (!kHiOnly) al[m] = *reinterpret_cast<const bf16x8*>(&img[2 * APIXP * 8 + koff[s_] + m * AW * 8]);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NT + n) * 2 + 0) * 64 + lane) * 8]);
                bf16x8 bl = bh;
                if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NT + n) * 2 + 1) * 64 + lane) * 8]);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[m], acc[m][n], 0, 0, 0);
                    if (!kHiOnly) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[m], acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
    };

    f32x4 acc1[MR][NT1], acc2[MR][NT2], zreg[MR][NH], hreg[MR][NH];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
        for (int n = 0; n < NT1; ++n) acc1[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int n = 0; n < NT2; ++n) acc2[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const int ix = x0 - 1 + li;
    auto load_h = [&]() {                                                // fp32 state at the lattice points (r * h, update); 0 outside the map
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int iy = y0 - 1 + wv * MR + m;
            const bool inimg = (iy >= 0) & (iy < h) & (ix >= 0) & (ix < w);
            const long pix = inimg ? (long)iy * w + ix : 0;
#pragma unroll
            for (int n = 0; n < NH; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) hreg[m][n][r] = inimg ? g.h_in[(long)(16 * n + 4 * lk + r) * hw + pix] : 0.0f;
        }
    };
    // z stays in registers; r * h goes to its LDS image in split form (the values effi_sr_store4 would have written to the map)
    auto gates = [&]() {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int p = (wv * MR + m + 1) * AW + li + 1;
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                const int co = 16 * n + 4 * lk;
                f32x4 rh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    zreg[m][n][r] = effi_sigmoid_split(acc1[m][n][r] + g.bzr[co + r]) * 1.0f;
                    rh[r] = effi_sigmoid_split(acc1[m][n + NH][r] + g.bzr[16 * NH + co + r]) * hreg[m][n][r];
                }
                const bf16x4 h4 = __builtin_convertvector(rh, bf16x4);
                const int e = ((lk >> 1) * APIXP + p) * 8 + (lk & 1) * 4;    // octet lk >> 1 of the tile's 16 channels, half lk & 1
                *reinterpret_cast<bf16x4*>(&lds_rh[n][e]) = h4;
                if (!kHiOnly) {
                    const bf16x4 l4 = __builtin_convertvector(rh - __builtin_convertvector(h4, f32x4), bf16x4);
                    *reinterpret_cast<bf16x4*>(&lds_rh[n][2 * APIXP * 8 + e]) = l4;
                }
            }
        }
    };
    // the 14 x 14 interior of the lattice: h' = (1 - z) h + z tanh(q), in both forms
    auto update = [&]() {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int i = wv * MR + m, iy = y0 - 1 + i;
            if (i < 1 || i > OT || li < 1 || li > OT || iy >= h || ix >= w) continue;      // (iy, ix >= 0 here)
            const long pix = (long)iy * w + ix;
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                const int co = 16 * n + 4 * lk;
                f32x4 gv;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    gv[r] = (1.0f - zreg[m][n][r]) * hreg[m][n][r] + zreg[m][n][r] * effi_tanh_split(acc2[m][n][r] + g.bq[co + r]);
#pragma unroll
                for (int r = 0; r < 4; ++r) g.h_out[(long)(co + r) * hw + pix] = gv[r];
                effi_sr_store4(g.H_out, hp, wp, co, iy, ix, gv);
            }
        }
    };

    // (Requesting EVERY operand of the workgroup up front at hd = 16 -- both input chunks, the fp32 state, the four weight blocks, 128
    // registers in flight -- measured SLOWER than one chunk ahead: 70 against 62 us per launch at 592x800.)
    // ---- z | r on the lattice ------------------------------------------------------------------------
    prefetch(0);
    stash(0);
    __syncthreads();
    for (int k = 0; k < NCH; ++k) {
        prefetch(k + 1);                                                 // (k + 1 = NCH: the first chunk of the q layer)
        mma(std::integral_constant<int, NT1>{}, acc1, lds_a);
        if (k + 1 < NCH) {
            __syncthreads();
            stash(k + 1);
            __syncthreads();
        }
    }
    load_h();
    gates();
    // ---- q on the same lattice ([r * h, x]) --------------------------------------------------------------------------------------
    __syncthreads();                                                     // r * h written, the z | r layer's fragments read
    stash(NCH);
    __syncthreads();
    for (int k = NCH; k < 2 * NCH; ++k) {
        if (k + 1 < 2 * NCH) prefetch(k + 1);
        mma(std::integral_constant<int, NT2>{}, acc2, (k - NCH < NH) ? lds_rh[k - NCH] : lds_a);
        if (k + 1 < 2 * NCH) {
            __syncthreads();
            stash(k + 1);
            __syncthreads();
        }
    }
    update();
}

template <int NH>
static int launch_gru_fused(const GruFusedArgs& g, hipStream_t st) {
    const int tiles_x = effi_cdiv(g.w, 14), ntiles = tiles_x * effi_cdiv(g.h, 14);
    hipLaunchKernelGGL((gru_zr_q_fused_kernel<NH>), dim3(ntiles), dim3(256), 0, st, g, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

}  // namespace

// h_out / H_out must not alias h_in / H_in (see the header of this file).  hd in {16, 32}; maps of the geometry effi_sr_geometry gives.
extern "C" int EFFI_FN(effi_gru_zr_q_fused_bf16x3_sr)(const void* H_in, const void* X, const float* h_in, const void* wzr_pack, const float* bias_zr,
                                                      const void* wq_pack, const float* bias_q, int hd, int h, int w, int hp, int wp,
                                                      float* h_out, void* H_out, effi_stream_t stream) {
    if (!H_in || !X || !h_in || !wzr_pack || !bias_zr || !wq_pack || !bias_q || !h_out || !H_out) return EFFI_ERR_BADARG;
    if (h < 1 || w < 1 || hp < h + 2 || wp < w + 2) return EFFI_ERR_BADARG;
    if (H_out == H_in || H_out == X || h_out == h_in) return EFFI_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(H_in) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(H_out)) & 15) return EFFI_ERR_BADARG;
    if ((long)hp * wp * 4 >= (1L << 31)) return EFFI_ERR_UNSUPPORTED;
    const GruFusedArgs g{reinterpret_cast<const unsigned short*>(H_in), reinterpret_cast<const unsigned short*>(X), h_in,
                         reinterpret_cast<const unsigned short*>(wzr_pack), bias_zr, reinterpret_cast<const unsigned short*>(wq_pack), bias_q,
                         h_out, reinterpret_cast<unsigned short*>(H_out), h, w, hp, wp};
    hipStream_t st = effi_s(stream);
    if (hd == 16) return launch_gru_fused<1>(g, st);
    if (hd == 32) return launch_gru_fused<2>(g, st);
    return EFFI_ERR_UNSUPPORTED;
}
