// K9: 2-D convolutions of the GRU update block (models/update.py:14-15,36-38,73-81,109-112) as an
// implicit GEMM on the fp32 matrix cores: M = pixels, N = output channels, K = (tap, input channel),
// v_mfma_f32_16x16x4_f32 (exact fp32: bitwise a k-ordered fmaf chain, same rate as the vector ALU but
// one VGPR per operand and no per-lane weight broadcast).
//
// A 256-thread block (4 waves) owns a 16x16 pixel tile and ALL output channels; wave v owns rows
// 4v..4v+3 (4 M-tiles of 16 pixels) x NT N-tiles of 16 channels -> 4*NT accumulator tiles.
//   A operand  (lane l: pixel l&15, k = l>>4): read from a halo'd LDS tile [16 ch][18][18] whose channel
//              planes are padded to 336 floats (= 16 mod 32), so every ds_read_b32 is conflict-free;
//              the next 16-channel chunk is fetched into registers while the current one is multiplied
//              (double-buffered LDS, one barrier per chunk).
//   B operand  (lane l: k = l>>4, cout l&15): weights are host-packed in exactly that lane order, so each
//              k-step's B tile is ONE coalesced 256-byte global load (L1/L2 resident, shared by all blocks).
// Epilogues (bias, activation, GRU gating, depth-head update) are fused; see EFFI_EPI_* in the header.
#include "conv2d_x3.hpp"

namespace {


template <int KS, int NT, int EPI>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(const Conv2dArgs a) {
    constexpr int R = KS / 2, TILE = 16, IW = TILE + 2 * R, CC = 16;
    constexpr int PL = (KS == 3) ? 336 : 272;                 // plane stride, == 16 (mod 32)
    constexpr int NE = CC * IW * IW;                          // elements staged per chunk
    constexpr int NLD = (NE + 255) / 256;
    __shared__ float lds[2][CC * PL];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const int h = a.h, w = a.w;
    const long hw = (long)h * w;

    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    float st[NLD];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = tid + j * 256;
            float v = 0.0f;
            if (e < NE) {
                const int c = e / (IW * IW);
                const int r = e - c * (IW * IW);
                const int yy = r / IW, xx = r - yy * IW;
                const int gy = y0 - R + yy, gx = x0 - R + xx;
                int cg = chunk * CC + c;
                if (cg < a.cin && gy >= 0 && gy < h && gx >= 0 && gx < w) {
                    const float* p;
                    if (cg < a.ch[0]) p = a.src[0];
                    else if ((cg -= a.ch[0]) < a.ch[1]) p = a.src[1];
                    else { cg -= a.ch[1]; p = a.src[2]; }
                    v = p[(long)cg * hw + (long)gy * w + gx];
                }
            }
            st[j] = v;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = tid + j * 256;
            if (e < NE) {
                const int c = e / (IW * IW);
                const int r = e - c * (IW * IW);
                lds[buf][c * PL + r] = st[j];
            }
        }
    };

    const int nchunks = (a.kgroups + 3) / 4;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) fetch(ch + 1);
        const int kgn = min(4, a.kgroups - ch * 4);
        for (int kq = 0; kq < kgn; ++kq) {
            const float* ab = &lds[buf][(kq * 4 + lk) * PL + (wv * 4) * IW + li];
            const float* __restrict__ wb = a.wpack + (long)(ch * 4 + kq) * (KS * KS) * NT * 64 + lane;
#pragma unroll
            for (int tap = 0; tap < KS * KS; ++tap) {
                const int ky = tap / KS, kx = tap % KS;
                float av[4], bv[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) bv[n] = wb[(tap * NT + n) * 64];
#pragma unroll
                for (int m = 0; m < 4; ++m) av[m] = ab[(m + ky) * IW + kx];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) stash(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds pixels x = x0 + 4*lk + r (r = 0..3) of row y, channel 16*n + li
    const int x = x0 + 4 * lk;
    const bool vec = ((w & 3) == 0);
    float lo = 0.0f, hi = 0.0f;
    if (EPI == EFFI_EPI_HEAD) { lo = a.disp_range[0]; hi = a.disp_range[a.n_range - 1]; }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int y = y0 + wv * 4 + m;
        if (y >= h || x >= w) continue;
        const long pix = (long)y * w + x;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = n * 16 + li;
            if (co >= a.cout) continue;
            const float b = a.bias[co];
            float v[4] = {acc[m][n][0] + b, acc[m][n][1] + b, acc[m][n][2] + b, acc[m][n][3] + b};
            float* dst;
            if (EPI == EFFI_EPI_PLAIN) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], a.act);
                dst = a.out0 + (long)co * hw + pix;
            } else if (EPI == EFFI_EPI_GRU_ZR) {
                if (co < a.hd) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = effi_sigmoid(v[r]);
                    dst = a.out0 + (long)co * hw + pix;
                } else {
                    const float* hp = a.aux0 + (long)(co - a.hd) * hw + pix;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (x + r < w) ? effi_sigmoid(v[r]) * hp[r] : 0.0f;
                    dst = a.out1 + (long)(co - a.hd) * hw + pix;
                }
            } else if (EPI == EFFI_EPI_GRU_Q) {
                const float* hp = a.aux0 + (long)co * hw + pix;
                const float* zp = a.aux1 + (long)co * hw + pix;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (x + r < w) {
                        const float z = zp[r];
                        v[r] = (1.0f - z) * hp[r] + z * tanhf(v[r]);
                    }
                }
                dst = a.out0 + (long)co * hw + pix;
            } else {  // EFFI_EPI_HEAD: single channel
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (x + r < w) v[r] = a.aux0[pix + r] + tanhf(v[r]);
                dst = a.out0 + pix;
                float* d2 = a.out1 + pix;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (x + r < w) d2[r] = effi_inv_to_depth(v[r], lo, hi);
            }
            if (vec) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (x + r < w) dst[r] = v[r];
            }
        }
    }
}

// Epilogue of one lane's result: v[0..3] = conv + bias for pixels x..x+3 of row y, output channel co.
// Shared by the fp32-MFMA kernel and the split-bf16 kernel (same C/D fragment layout).
template <int EPI, bool ALIGNED>
__device__ __forceinline__ void conv_epilogue_store(const Conv2dArgs& a, float (&v)[4], int co, int x, int y, long pix, long hw,
                                                    int zpl, float lo, float hi) {
    const int h = a.h, w = a.w;
    float* dst;
    if (EPI == EFFI_EPI_NHWC) {              // channel-last output: 16 lanes (channels) write one 64-B run per pixel
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (x + r < w) a.out0[(pix + r) * a.cout + co] = apply_act(v[r], a.act);
        return;
    }
    if (EPI == EFFI_EPI_PLAIN || EPI == EFFI_EPI_ADD_UP2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], a.act);
        if (EPI == EFFI_EPI_ADD_UP2) {      // + nearest-upsampled coarser map (two source pixels per float4)
            const float* up = a.aux0 + (long)co * ((long)(h >> 1) * (w >> 1)) + (long)(y >> 1) * (w >> 1) + (x >> 1);
            const float u0 = up[0], u1 = (x + 2 < w) ? up[1] : 0.0f;
            v[0] += u0;
            v[1] += u0;
            v[2] += u1;
            v[3] += u1;
        }
        dst = a.out0 + (long)co * a.ostride + (long)zpl * hw + pix;
        if (!ALIGNED) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (x + r < w) dst[r] = v[r];
            return;
        }
    } else if (EPI == EFFI_EPI_GRU_ZR) {
        if (co < a.hd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = effi_sigmoid(v[r]);
            dst = a.out0 + (long)co * hw + pix;
        } else {
            const float4 hv = *reinterpret_cast<const float4*>(a.aux0 + (long)(co - a.hd) * hw + pix);
            v[0] = effi_sigmoid(v[0]) * hv.x;
            v[1] = effi_sigmoid(v[1]) * hv.y;
            v[2] = effi_sigmoid(v[2]) * hv.z;
            v[3] = effi_sigmoid(v[3]) * hv.w;
            dst = a.out1 + (long)(co - a.hd) * hw + pix;
        }
    } else if (EPI == EFFI_EPI_GRU_Q) {
        const float4 hv = *reinterpret_cast<const float4*>(a.aux0 + (long)co * hw + pix);
        const float4 zv = *reinterpret_cast<const float4*>(a.aux1 + (long)co * hw + pix);
        v[0] = (1.0f - zv.x) * hv.x + zv.x * tanhf(v[0]);
        v[1] = (1.0f - zv.y) * hv.y + zv.y * tanhf(v[1]);
        v[2] = (1.0f - zv.z) * hv.z + zv.z * tanhf(v[2]);
        v[3] = (1.0f - zv.w) * hv.w + zv.w * tanhf(v[3]);
        dst = a.out0 + (long)co * hw + pix;
    } else {  // EFFI_EPI_HEAD
        const float4 iv = *reinterpret_cast<const float4*>(a.aux0 + pix);
        v[0] = iv.x + tanhf(v[0]);
        v[1] = iv.y + tanhf(v[1]);
        v[2] = iv.z + tanhf(v[2]);
        v[3] = iv.w + tanhf(v[3]);
        dst = a.out0 + pix;
        *reinterpret_cast<float4*>(a.out1 + pix) =
            make_float4(effi_inv_to_depth(v[0], lo, hi), effi_inv_to_depth(v[1], lo, hi),
                        effi_inv_to_depth(v[2], lo, hi), effi_inv_to_depth(v[3], lo, hi));
    }
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
}


// ------------------------------------------------------------------------------------------------
// v3 (used whenever w % 4 == 0): both operands come from LDS inside the k-loop, workgroups are persistent.
//   * chunk = CC input channels (CC/4 k-groups; 16 when the accumulators leave room, else 8):
//     A tile [CC][TR+2][24] floats (columns x0-4 .. x0+19, so every row is fetched as 6 aligned float4)
//     and the chunk's B fragments [CC/4][KS*KS][NT][64].
//   * work items are (tile, chunk) pairs; while item i is multiplied, item i+1 (the next chunk, or the
//     FIRST chunk of the workgroup's next tile) is prefetched into registers with branch-free float4
//     loads that stay in flight during the whole multiply phase (no vmcnt wait in the k-loop), and is
//     written to LDS between two barriers.  A workgroup walks tiles blockIdx.x, +gridDim.x, ... so the
//     global-load latency of a tile's first chunk is exposed only once per workgroup.
//   * the k-steps of a chunk are fully unrolled with double-buffered fragments.
//   * MR = rows per wave (1, 2 or 4): small images use small MR so the grid still covers the 256 CUs.
// ------------------------------------------------------------------------------------------------
template <int KS, int NT, int MR, int EPI, int CC, bool PERSIST, bool ALIGNED = true, int S = 1>
__global__ __launch_bounds__(256) void conv2d_mfma_v2_kernel(const Conv2dArgs a, int tiles_x, int ntiles) {
    // S = stride (1, or 2 for the 5x5 down-sampling convs of the feature pyramid): output tile 16 x TR, input
    // tile S*(TR-1)+KS rows x S*15+KS columns, fetched from column S*x0 - XLEFT in aligned float4 units.
    // (For S = 2 the A reads have a 2-lane stride: 2-way bank conflicts, irrelevant at ~1 LDS read per MFMA.)
    constexpr int R = KS / 2, TR = 4 * MR, AR = S * (TR - 1) + KS;
    constexpr int XLEFT = (KS == 1) ? 0 : 4, XOFF = XLEFT - R;
    constexpr int AW = (KS == 1) ? 16 : ((XOFF + S * 15 + KS + 3) / 4) * 4, AQ = AW / 4;
    constexpr int PL0 = AR * AW;
    constexpr int PLA = PL0 + ((16 - (PL0 % 32)) + 32) % 32;          // channel-plane stride == 16 (mod 32)
    constexpr int KG = CC / 4, T = KG * KS * KS;
    constexpr int NA = CC * AR * AQ;                                    // float4 per A chunk
    constexpr int NA4 = (NA + 255) / 256;
    constexpr int NB = T * NT * 16;                                     // float4 per B chunk
    constexpr int NB4 = (NB + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds_a[CC * PLA];
    __shared__ __attribute__((aligned(16))) float lds_b[T * NT * 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w, hin = a.hin, win = a.win;
    const long hw = (long)h * w, hwin = (long)hin * win;
    const int zpl = a.zcount ? (int)blockIdx.y : 0;              // plane of a z-batched launch
    // XCD-aware tile order (non-persistent launches): workgroups are dealt round-robin to the 8 XCDs, so
    // logical tiles are assigned such that each XCD owns a contiguous run of tiles -- x/y-neighbouring tiles,
    // whose halo rows share cache lines, then hit the same L2 instead of re-fetching over the fabric.
    int tile = PERSIST ? (int)blockIdx.x : effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;

    // per-thread state of the A prefetch: LDS slot and global offset (inside one channel map) of each of
    // this thread's float4 elements; the offset depends on the tile and is refreshed by setup().  Channel /
    // row / column are re-derived from the element index where needed (a few integer ops) rather than
    // kept in registers.
    int a_lds[NA4], a_off[NA4], a_gx[ALIGNED ? 1 : NA4];
#pragma unroll
    for (int j = 0; j < NA4; ++j) {
        const int f = min(tid + j * 256, NA - 1);
        const int c = f / (AR * AQ);
        const int r = f - c * (AR * AQ);
        const int row = r / AQ;
        a_lds[j] = c * PLA + row * AW + 4 * (r - row * AQ);
    }
    auto setup = [&](int t, int& x0, int& y0) {
        const int ty_ = t / tiles_x;
        x0 = (t - ty_ * tiles_x) * 16;
        y0 = ty_ * TR;
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            const int f = min(tid + j * 256, NA - 1);
            const int r = f % (AR * AQ);
            const int row = r / AQ;
            const int gy = S * y0 - R + row, gx = S * x0 - XLEFT + 4 * (r - row * AQ);
            // ALIGNED (win % 4 == 0): a float4 is entirely inside or outside the row.  Otherwise the segment may
            // straddle the row end; it is then fetched component-wise (prefetch) and a_off only says "row exists"
            const bool xin = ALIGNED ? ((gx >= 0) & (gx < win)) : ((gx + 3 >= 0) & (gx < win));
            a_off[j] = ((tid + j * 256 < NA) & (gy >= 0) & (gy < hin) & xin) ? gy * win + gx : -1;
            if (!ALIGNED) a_gx[j] = gx;
        }
    };

    float4 pa[NA4], pb[NB4];
    const int nchunks = (a.kgroups + KG - 1) / KG;
    const long wchunk = (long)T * NT * 64;                              // floats of wpack per chunk
    const long wtotal = (long)a.kgroups * KS * KS * NT * 64;

    auto prefetch = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            int cg = ch * CC + min(tid + j * 256, NA - 1) / (AR * AQ);
            bool ok = (a_off[j] >= 0) & (cg < a.cin);
            int srci = 0;
            if (cg >= a.ch[0]) {
                cg -= a.ch[0];
                srci = 1;
                if (cg >= a.ch[1]) { cg -= a.ch[1]; srci = 2; }
            }
            const float* p = (srci == 0) ? a.src[0] : (srci == 1 ? a.src[1] : a.src[2]);
            if (a.zcount) {                                   // source s = input plane S*z + s - 1
                const int zz = S * zpl + srci - 1;
                ok &= (zz >= 0) & (zz < a.zin);
                p += (long)max(min(zz, a.zin - 1), 0) * hwin;
            }
            if (ALIGNED) {
                const float4 t = *reinterpret_cast<const float4*>(ok ? p + (long)cg * a.cstride + a_off[j] : a.wpack);
                pa[j] = ok ? t : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            } else {
                const float* q = p + (long)cg * a.cstride + a_off[j];
                float e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool eok = ok & (a_gx[j] + i >= 0) & (a_gx[j] + i < win);
                    const float t = *(eok ? q + i : a.wpack);
                    e[i] = eok ? t : 0.0f;
                }
                pa[j] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < NB4; ++j) {
            const long f = (long)ch * wchunk + 4L * (tid + j * 256);
            const bool ok = (tid + j * 256 < NB) & (f < wtotal);     // partial last chunk: the tail is zero
            const float4 t = *reinterpret_cast<const float4*>(a.wpack + (ok ? f : 0));
            pb[j] = ok ? t : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < NA4; ++j)
            if (tid + j * 256 < NA) *reinterpret_cast<float4*>(&lds_a[a_lds[j]]) = pa[j];
#pragma unroll
        for (int j = 0; j < NB4; ++j)
            if (tid + j * 256 < NB) *reinterpret_cast<float4*>(&lds_b[4 * (tid + j * 256)]) = pb[j];
    };

    int x0, y0;                       // tile being multiplied
    setup(tile, x0, y0);
    prefetch(0);
    stash();
    __syncthreads();
    const float* ab = &lds_a[lk * PLA + (S * wv * MR) * AW + S * li + XOFF];
    const float* bb = &lds_b[lane];
    float lo = 0.0f, hi = 0.0f;
    if (EPI == EFFI_EPI_HEAD) { lo = a.disp_range[0]; hi = a.disp_range[a.n_range - 1]; }

    while (true) {
        f32x4 acc[MR][NT];
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const int next_tile = PERSIST ? tile + (int)gridDim.x : ntiles;
        int nx0 = 0, ny0 = 0;
        for (int ch = 0; ch < nchunks; ++ch) {
            bool have_next = true;
            if (ch + 1 < nchunks) {
                prefetch(ch + 1);
            } else if (next_tile < ntiles) {
                setup(next_tile, nx0, ny0);
                prefetch(0);
            } else {
                have_next = false;
            }
            // k-loop of the chunk: KG k-groups (rolled) x KS*KS taps (unrolled, fragments double-buffered)
#pragma unroll 1
            for (int kg = 0; kg < KG; ++kg) {
                const float* abk = ab + kg * 4 * PLA;
                const float* bbk = bb + kg * (KS * KS) * NT * 64;
                float af[2][MR], bf[2][NT];
                auto frag = [&](int tap, int s) {
                    const int ky = tap / KS, kx = tap % KS;
#pragma unroll
                    for (int n = 0; n < NT; ++n) bf[s][n] = bbk[(tap * NT + n) * 64];
#pragma unroll
                    for (int m = 0; m < MR; ++m) af[s][m] = abk[(S * m + ky) * AW + kx];
                };
                frag(0, 0);
#pragma unroll
                for (int tap = 0; tap < KS * KS; ++tap) {
                    if (tap + 1 < KS * KS) frag(tap + 1, (tap + 1) & 1);
#pragma unroll
                    for (int m = 0; m < MR; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tap & 1][m], bf[tap & 1][n], acc[m][n], 0, 0, 0);
                }
            }
            if (have_next) {
                __syncthreads();
                stash();
                __syncthreads();
            }
        }

        // ---- epilogue of tile (x0, y0): pixels x = x0 + 4*lk + r of row y, channel 16*n + li
        const int x = x0 + 4 * lk;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int y = y0 + wv * MR + m;
            if (y >= h || x >= w) continue;
            const long pix = (long)y * w + x;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n * 16 + li;
                if (co >= a.cout) continue;
                const float b = a.bias[co];
                float v[4] = {acc[m][n][0] + b, acc[m][n][1] + b, acc[m][n][2] + b, acc[m][n][3] + b};
                conv_epilogue_store<EPI, ALIGNED>(a, v, co, x, y, pix, hw, zpl, lo, hi);
            }
        }
        if (next_tile >= ntiles) break;
        tile = next_tile;
        x0 = nx0;
        y0 = ny0;
    }
}


// ------------------------------------------------------------------------------------------------
// TWO chained 3x3 layers in one kernel, the intermediate map in LDS only ("tail" of the update block's encoder,
// models/update.py:87-96: cor = relu(convc2(cor1)), dfm = relu(convd2(dfm1)), out = relu(convc(cat(convd(cat(cor, dfm)), context)))).
// At 592x800 the two intermediate 16-channel maps cost 120 MB of HBM traffic per GRU iteration (written by one launch, read with
// halo by the next) around layers whose MFMA work is a few microseconds; here they never leave the CU:
//   * a workgroup owns 12 rows x 16 columns of the output.  Phase A computes the first layer on the tile grown by one pixel
//     (14 x 18 = 252 pixels = 16 MFMA pixel groups, 4 per wave; 1.31x the pixels of the tile) from an input tile grown by two
//     (16 rows x 24 columns staged as in the single-layer kernel).  A lane group's 16 pixels are 16 consecutive LINEAR indices of the
//     14 x 18 region, so a group may wrap a row: the fragment address is per (lane, group) base + per (lane, K-step) tap offset.
//   * its result (bias, ReLU, zero outside the map = the second layer's zero padding) is split to hi/lo bf16 and written in the
//     A-tile layout [octet][pixel][8 ch] (one ds_write_b64 per lane, N-tile and group): "mid", 32 KB for 2 x 16 channels.
//   * phase B is the single-layer kernel's loop reading its A fragments from mid (row pitch 18) with all its weight chunks resident,
//     followed by the fused 1x1 epilogue of conv2d_k3_bf16x3_tile (EFFI_EPI_K1).
// Same split-precision arithmetic as the two launches it replaces; the first layer's values are identical, the second layer sees
// them as hi + lo exactly as its own staging would have split them, so the results are bitwise those of the two-launch form.
// ------------------------------------------------------------------------------------------------
struct EncTailArgs {
    const float* src_a;      // cor1 [hd][h][w]
    const float* src_b;      // dfm1 [hd][h][w]
    const float* w_a;        // convc2, bf16x3 fragments (packing.pack_conv2d_bf16x3)
    const float* bias_a;     // padded to 16*NTH
    const float* w_b;        // convd2
    const float* bias_b;
    const float* w_d;        // convd [2 hd -> cmix]
    const float* bias_d;     // padded to 16*NTB
    const float* extra;      // context [c_extra][h][w]
    int c_extra;
    const float* w2;         // convc as 1x1 fragments (packing.pack_conv1x1_after)
    const float* bias2;      // padded to 16*NT2
    int cout2;
    int h, w;
    float* out;              // [cout2][h][w]
    const float* zeros;
};

template <int NTH, int NTB>
__global__ __launch_bounds__(256) void encoder_tail_bf16x3_kernel(const EncTailArgs a, int tiles_x, int ntiles) {
    constexpr int TR = 12, TW = 16, RW = TW + 2, RH = TR + 2, NPM = RW * RH;          // 252 mid pixels
    static_assert((NPM + 15) / 16 == 16, "16 pixel groups, 4 per wave (the last group has 12 real pixels)");
    constexpr int AR = TR + 4, AW = TW + 8, AQ = AW / 4, APIX = AR * AW, CCH = 16, NKS = 5;
    constexpr int NITEMS = (APIX / 4) * 2;                                           // 192 staging items
    constexpr int NOCT = 4 * NTH;                                                    // mid octets: cor (2 NTH), dfm (2 NTH)
    constexpr int NBFA = NKS * NTH * 2 * 64, NBFB = NKS * NTB * 2 * 64;              // 16-byte units of B per chunk
    constexpr int NCHB = 2 * NTH;                                                    // phase-B chunks
    constexpr int NBU = (NBFA > NCHB * NBFB) ? NBFA : NCHB * NBFB;
    constexpr int NB4A = (NBFA + 255) / 256, NB4B = (NCHB * NBFB + 255) / 256;
    constexpr int NBP = ((NB4A > NB4B) ? NB4A : NB4B) * 256;
    static_assert(NBP >= NBU, "weight buffer");
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[APIX * CCH];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[APIX * CCH];
    __shared__ __attribute__((aligned(16))) unsigned short mid_h[NOCT * NPM * 8];
    __shared__ __attribute__((aligned(16))) unsigned short mid_l[NOCT * NPM * 8];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[NBP * 8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * TW, y0 = ty_ * TR;

    // staging item of this thread: (pixel quad, octet) of the 16 x 24 input tile whose origin is (y0 - 2, x0 - 4)
    const bool stager = tid < NITEMS;
    const int pq = stager ? tid % (APIX / 4) : 0, soct = stager ? tid / (APIX / 4) : 0;
    const int srow = pq / AQ, sqx = pq - srow * AQ;
    const int sgy = y0 - 2 + srow, sgx = x0 - 4 + 4 * sqx;
    const bool s_in = stager & (sgy >= 0) & (sgy < h) & (sgx >= 0) & (sgx < w);
    const int s_off = sgy * w + sgx;
    const int s_lds = (soct * APIX + srow * AW + 4 * sqx) * 8;

    // with one chunk per first-layer convolution (NTH == 1) both convolutions' input octets are requested up front: the second
    // one's latency then hides behind the first one's MFMA phase instead of being paid with two workgroups per CU
    f32x4 pa[8], pa2[8];
    auto prefetch_into = [&](f32x4 (&dst)[8], const float* src, int ch) {
        const int cb = ch * CCH + soct * 8;
        const float* q = s_in ? src + ((long)cb * hw + s_off) : a.zeros;
        const long step = s_in ? hw : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dst[e] = *reinterpret_cast<const f32x4*>(q);
            q += step;
        }
    };
    auto prefetch = [&](const float* src, int ch) { prefetch_into(pa, src, ch); };
    auto stash = [&](const float* wpk, int ch) {
        const unsigned short* wbf = reinterpret_cast<const unsigned short*>(wpk);
        f32x4 tb[NB4A];
#pragma unroll
        for (int j = 0; j < NB4A; ++j) {
            const int u = min(tid + j * 256, NBFA - 1);
            tb[j] = *reinterpret_cast<const f32x4*>(wbf + ((long)ch * NBFA + u) * 8);
        }
        if (stager) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                bf16x8 hi, lo;
                split_octet(pa, px, hi, lo);
                *reinterpret_cast<bf16x8*>(&lds_ah[s_lds + px * 8]) = hi;
                if (!kHiOnly) *reinterpret_cast<bf16x8*>(&lds_al[s_lds + px * 8]) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < NB4A; ++j) *reinterpret_cast<f32x4*>(&lds_b[(tid + j * 256) * 8]) = tb[j];
    };

    // phase A addressing: group g = 4 wv + m, linear pixel p = 16 g + li of the 14 x 18 region whose origin is (y0 - 1, x0 - 1)
    int gbase[4], pmid[4];
    bool pin[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int p_ = min(16 * (4 * wv + m) + li, NPM - 1);  // lanes past the region repeat its last pixel (same value, same slot)
        const int pr = p_ / RW, pc = p_ - pr * RW;
        pmid[m] = p_;
        gbase[m] = (pr * AW + pc + 2) * 8;                    // staged row pr + dy, staged column pc + dx + 2
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        pin[m] = (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w);
    }
    int koffa[NKS], koffb[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int item = 4 * s_ + lk;
        const int tap = min(item >> 1, 8), oct = item & 1;    // items 18, 19 are padding (B is zero there)
        koffa[s_] = ((tap / 3) * AW + tap % 3 + oct * APIX) * 8;
        koffb[s_] = ((3 * wv + tap / 3) * RW + li + tap % 3 + oct * NPM) * 8;
    }

    // ---- phase A: the two first-layer convolutions, one after the other ---------------------------------------
    prefetch(a.src_a, 0);
    if (NTH == 1) prefetch_into(pa2, a.src_b, 0);
#pragma unroll 1
    for (int job = 0; job < 2; ++job) {
        const float* src = job ? a.src_b : a.src_a;
        const float* wpk = job ? a.w_b : a.w_a;
        const float* bias = job ? a.bias_b : a.bias_a;
        f32x4 acc[4][NTH];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTH; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
        for (int ch = 0; ch < NTH; ++ch) {
            __syncthreads();                                  // the previous chunk's fragments are no longer read
            stash(wpk, ch);
            __syncthreads();
            if (ch + 1 < NTH) prefetch(src, ch + 1);
            else if (job == 0) {
                if (NTH == 1) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pa[e] = pa2[e];
                } else {
                    prefetch(a.src_b, 0);
                }
            }
#pragma unroll
            for (int s_ = 0; s_ < NKS; ++s_) {
                bf16x8 ah[4], al[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    ah[m] = *reinterpret_cast<const bf16x8*>(&lds_ah[gbase[m] + koffa[s_]]);
                    if (!kHiOnly) al[m] = *reinterpret_cast<const bf16x8*>(&lds_al[gbase[m] + koffa[s_]]);
                }
#pragma unroll
                for (int n = 0; n < NTH; ++n) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NTH + n) * 2 + 0) * 64 + lane) * 8]);
                    bf16x8 bl = bh;
                    if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NTH + n) * 2 + 1) * 64 + lane) * 8]);
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[m], acc[m][n], 0, 0, 0);
                        if (!kHiOnly) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[m], acc[m][n], 0, 0, 0);
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[m], acc[m][n], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // bias + ReLU, zero outside the map, split, into mid: lane = (pixel, channels 16 n + 4 lk .. + 3) -> half an octet slot
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NTH; ++n) {
                f32x4 vf;
#pragma unroll
                for (int r = 0; r < 4; ++r) vf[r] = pin[m] ? fmaxf(acc[m][n][r] + bias[n * 16 + 4 * lk + r], 0.0f) : 0.0f;
                const bf16x4 vh = __builtin_convertvector(vf, bf16x4);
                const bf16x4 vl = __builtin_convertvector(vf - __builtin_convertvector(vh, f32x4), bf16x4);
                const int o = ((job * 2 * NTH + 2 * n + (lk >> 1)) * NPM + pmid[m]) * 8 + 4 * (lk & 1);
                *reinterpret_cast<bf16x4*>(&mid_h[o]) = vh;
                if (!kHiOnly) *reinterpret_cast<bf16x4*>(&mid_l[o]) = vl;
            }
    }
    __syncthreads();                                          // mid complete, phase A's weight chunk no longer read

    // ---- phase B: convd over mid (all its weight chunks resident), then the fused 1x1 ---------------------------
    {
        const unsigned short* wbf = reinterpret_cast<const unsigned short*>(a.w_d);
#pragma unroll
        for (int j = 0; j < NB4B; ++j) {
            const int u = min(tid + j * 256, NCHB * NBFB - 1);
            *reinterpret_cast<f32x4*>(&lds_b[(tid + j * 256) * 8]) = *reinterpret_cast<const f32x4*>(wbf + (long)u * 8);
        }
    }
    __syncthreads();
    constexpr int MR = 3;
    f32x4 acc[MR][NTB];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NTB; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ch = 0; ch < NCHB; ++ch) {
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            bf16x8 ah[MR], al[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(&mid_h[ch * 2 * NPM * 8 + koffb[s_] + m * RW * 8]);
                if (!kHiOnly) al[m] = *reinterpret_cast<const bf16x8*>(&mid_l[ch * 2 * NPM * 8 + koffb[s_] + m * RW * 8]);
            }
#pragma unroll
            for (int n = 0; n < NTB; ++n) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[(ch * NBFB + ((s_ * NTB + n) * 2 + 0) * 64 + lane) * 8]);
                bf16x8 bl = bh;
                if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[(ch * NBFB + ((s_ * NTB + n) * 2 + 1) * 64 + lane) * 8]);
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
    }

    // fused 1x1 over cat(convd + bias, context), ReLU: as the EFFI_EPI_K1 epilogue of conv2d_k3_bf16x3_tile
    const unsigned short* w2 = reinterpret_cast<const unsigned short*>(a.w2);
    const int nt2 = (a.cout2 + 15) >> 4;
    bf16x4 xh[MR][NTB + 1], xl[MR][NTB + 1];
    bool inside[MR];
    long pixm[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int x = x0 + li, y = y0 + 3 * wv + m;
        inside[m] = (y < h) & (x < w);
        pixm[m] = inside[m] ? (long)y * w + x : 0;
        f32x4 ex;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * lk + r;
            ex[r] = (inside[m] && c < a.c_extra) ? a.extra[(long)c * hw + pixm[m]] : 0.0f;
        }
#pragma unroll
        for (int n = 0; n <= NTB; ++n) {
            f32x4 vf;
#pragma unroll
            for (int r = 0; r < 4; ++r) vf[r] = (n < NTB) ? acc[m][n < NTB ? n : 0][r] + a.bias_d[(n < NTB ? n : 0) * 16 + 4 * lk + r] : ex[r];
            xh[m][n] = __builtin_convertvector(vf, bf16x4);
            xl[m][n] = __builtin_convertvector(vf - __builtin_convertvector(xh[m][n], f32x4), bf16x4);
        }
    }
    for (int t = 0; t < nt2; ++t) {
        bf16x4 wh[NTB + 1], wl[NTB + 1];
#pragma unroll
        for (int n = 0; n <= NTB; ++n) {
            const long f = ((long)(t * (NTB + 1) + n) * 2) * 64 + lane;
            wh[n] = *reinterpret_cast<const bf16x4*>(w2 + f * 4);
            wl[n] = *reinterpret_cast<const bf16x4*>(w2 + (f + 64) * 4);
        }
        const int co = t * 16 + 4 * lk;
        float b2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) b2[r] = a.bias2[co + r];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int n = 0; n <= NTB; ++n) {
                o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[n], xh[m][n], o, 0, 0, 0);
                if (!kHiOnly) {
                    o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wl[n], xh[m][n], o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[n], xl[m][n], o, 0, 0, 0);
                }
            }
            if (inside[m]) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < a.cout2) a.out[(long)(co + r) * hw + pixm[m]] = fmaxf(o[r] + b2[r], 0.0f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Stride-1 3-D convolution with a ROLLING window of input planes (split precision, same MFMA scheme as above).
// The z-batched form above re-reads every input plane for each of the three output planes it contributes to; here a
// workgroup owns an (x, y) tile and a run of ZT consecutive output planes, keeps planes z-1, z, z+1 of all input channels in
// LDS (bf16 hi/lo, [slot][octet][pixel][8]) and fetches ONE new plane per output plane (into registers while the current
// plane is multiplied).  cin = 8*NOCT (NOCT in {1, 2}; one or two sources of 8/16 channels), cout <= 16*NT.
//   * K index = (dz, tap, octet): 27*NOCT items, 4 per K-step -> NKS = 7 (cin 8) or 14 (cin 16) K-steps; the whole B
//     operand ([NKS][NT][hi|lo][64][8] bf16, 14-56 KB) is copied to LDS once per workgroup.
//   * a lane's A fragment address = lane base + per-K-step constant + base of the slot that currently holds plane z+dz-1;
//     the slot bases rotate with z and are re-selected once per plane (NKS selects), row offsets are immediates.
// ------------------------------------------------------------------------------------------------
template <int NOCT, int NT, int MR>
__device__ __forceinline__ void conv3d_roll_bf16x3_body(const Conv2dArgs a, int tiles_x, int ntiles, int zt) {
    constexpr int TR = 4 * MR, AR = TR + 2, AW = 24, AQ = 6, XOFF = 3, XLEFT = 4;
    constexpr int APIX = AR * AW, NQ = APIX / 4, NITEMS = NQ * NOCT;
    constexpr int NIT = 27 * NOCT, NKS = (NIT + 3) / 4;
    constexpr int NBF = NKS * NT * 2 * 64;                             // 16-byte units of B
    constexpr int SLOT = NOCT * APIX * 8;                              // bf16 elements per plane slot
    static_assert(NITEMS <= 256, "one staging item per thread");
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[3 * SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[3 * SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[NBF * 8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w, D = a.zcount;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * 16, y0 = ty_ * TR;
    const int z0 = blockIdx.y * zt, z1 = min(z0 + zt, D);

    // staging item of this thread: (pixel quad, octet); an octet never straddles the two sources (ch[0] % 8 == 0)
    const bool stager = tid < NITEMS;
    const int pq = stager ? tid % NQ : 0, soct = stager ? tid / NQ : 0;
    const int srow = pq / AQ, sqx = pq - srow * AQ;
    const int sgy = y0 - 1 + srow, sgx = x0 - XLEFT + 4 * sqx;
    const int s_off = (stager & (sgy >= 0) & (sgy < h) & (sgx >= 0) & (sgx < w)) ? sgy * w + sgx : -1;
    const int s_lds = (soct * APIX + srow * AW + 4 * sqx) * 8;
    const float* s_src = (soct * 8 < a.ch[0]) ? a.src[0] + (long)(soct * 8) * a.cstride
                                              : a.src[1] + (long)(soct * 8 - a.ch[0]) * a.cstride;     // stagers only: soct < NOCT

    // two register buffers: the plane fetched for iteration z + 2 is in flight during iterations z and z + 1.  The buffers hold
    // the RAW loads (padding comes from a zero page), so no load is consumed before its conversion.
    f32x4 pa[2][8];
    auto prefetch = [&](auto buf_t, int z) {
        constexpr int bf = decltype(buf_t)::value;
        const bool ok = (s_off >= 0) & (z >= 0) & (z < D);
        const float* base = ok ? s_src + ((long)z * hw + s_off) : a.zeros;
        const long step = ok ? a.cstride : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[bf][e] = *reinterpret_cast<const f32x4*>(base + (long)e * step);
    };
    auto stash = [&](auto buf_t, int slot) {
        constexpr int bf = decltype(buf_t)::value;
        if (stager) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                bf16x8 hi, lo;
                split_octet(pa[bf], px, hi, lo);
                *reinterpret_cast<bf16x8*>(&lds_ah[slot * SLOT + s_lds + px * 8]) = hi;
                *reinterpret_cast<bf16x8*>(&lds_al[slot * SLOT + s_lds + px * 8]) = lo;
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;

    // B: whole operand, once
    {
        const unsigned short* wbf = reinterpret_cast<const unsigned short*>(a.wpack);
        for (int u = tid; u < NBF; u += 256)
            *reinterpret_cast<float4*>(&lds_b[u * 8]) = *reinterpret_cast<const float4*>(wbf + (long)u * 8);
    }

    // per K-step constants of this lane: element offset inside a slot of item 4s + lk = (dz, tap, oct), dz in the low bits
    int kconst[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int item = min(4 * s_ + lk, NIT - 1);                    // padding items read valid data, their B is zero
        const int oct = item % NOCT, dt = item / NOCT;
        const int dz = dt / 9, tap = dt - dz * 9;
        kconst[s_] = ((oct * APIX + (tap / 3) * AW + tap % 3) * 8) * 4 + dz;
    }
    const int lane_base = ((wv * MR) * AW + li + XOFF) * 8;

    f32x4 acc[MR][NT], rbias[NT];                                      // bias quads of the lane's N-tiles: once per workgroup, not per plane
#pragma unroll
    for (int n = 0; n < NT; ++n) rbias[n] = *reinterpret_cast<const f32x4*>(a.bias + n * 16 + 4 * lk);
    prefetch(B0{}, z0 - 1);
    prefetch(B1{}, z0);
    stash(B0{}, 0);
    prefetch(B0{}, z0 + 1);
    stash(B1{}, 1);
    prefetch(B1{}, z0 + 2);
    int rot = 0;                                                       // slot holding plane z - 1
    // one output plane: buffer `bt` holds plane z + 1 on entry and is refilled with plane z + 3
    auto plane = [&](auto bt, int z) {
        int s2 = rot + 2;
        s2 -= (s2 >= 3) ? 3 : 0;
        stash(bt, s2);                                                 // plane z + 1
        __syncthreads();
        if (z + 2 < z1) prefetch(bt, z + 3);
        int s1 = rot + 1;
        s1 -= (s1 >= 3) ? 3 : 0;
        const int rb0 = rot * SLOT, rb1 = s1 * SLOT, rb2 = s2 * SLOT;
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            const int dz = kconst[s_] & 3;
            const int off = lane_base + (kconst[s_] >> 2) + (dz == 0 ? rb0 : (dz == 1 ? rb1 : rb2));
            bf16x8 ah[MR], al[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(&lds_ah[off + m * AW * 8]);
                if (!kHiOnly) al[m] = *reinterpret_cast<const bf16x8*>(&lds_al[off + m * AW * 8]);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NT + n) * 2 + 0) * 64 + lane) * 8]);
                bf16x8 bl = bh;
                if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NT + n) * 2 + 1) * 64 + lane) * 8]);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    // weights x pixels: D[cout][pixel] (transposed fragment, see conv_epilogue_store_t)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[m], acc[m][n], 0, 0, 0);
                    if (!kHiOnly) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[m], acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[m], acc[m][n], 0, 0, 0);
                    }
                }
            }
        }
        const int x = x0 + li;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int y = y0 + wv * MR + m;
            if (y >= h || x >= w) continue;
            const long pix = (long)y * w + x;
#pragma unroll
            for (int n = 0; n < NT; ++n) conv_epilogue_store_t<EFFI_EPI_PLAIN>(a, acc[m][n], n * 16 + 4 * lk, pix, hw, z, 0, 0, &rbias[n]);
        }
        __syncthreads();                                               // all reads of slot `rot` done before it is refilled
        rot = s1;
    };
    for (int z = z0; z < z1; z += 2) {
        plane(B0{}, z);
        if (z + 1 < z1) plane(B1{}, z + 1);
    }
}

// cout <= 8: the upper half of the 16-row MFMA tile would multiply zero weights.  ROW-PAIR form: rows 0-7 of the tile are the output
// channels of image row y, rows 8-15 the same channels of row y + 1, and the K index runs over the 4 x 3 window of taps both rows see:
// K = (dz, dy in 0..3, dx, octet), 36 NOCT items = 9 NOCT K-steps for TWO rows instead of 2 x 7 NOCT (the weights of row y are
// zero at dy = 3, those of row y + 1 at dy = 0; packing.pack_conv3d_roll_bf16x3 emits this operand when cout <= 8): 36 % fewer
// MFMAs and A-fragment reads.  The B operand is 9 NOCT x 2 KB, so with 16 input channels the plane slots are stored 20 pixels wide
// (columns x0 - 2 .. x0 + 17 of the 24 fetched) to keep two workgroups per CU.
template <int NOCT, int MR>
__device__ __forceinline__ void conv3d_roll_rp_bf16x3_body(const Conv2dArgs a, int tiles_x, int ntiles, int zt) {
    static_assert(MR % 2 == 0, "rows go in pairs");
    constexpr int TR = 4 * MR, AR = TR + 2, AW = (NOCT == 2) ? 20 : 24, AQ = 6, XOFF = (NOCT == 2) ? 1 : 3, XLEFT = 4;
    constexpr int CSH = (NOCT == 2) ? 2 : 0;                           // columns dropped on the left of the fetched 24
    constexpr int APIX = AR * AW, NQ = AR * AQ, NITEMS = NQ * NOCT;
    constexpr int NIT = 36 * NOCT, NKS = NIT / 4;
    constexpr int SLOT = NOCT * APIX * 8;                              // bf16 elements per plane slot
    constexpr int MP = MR / 2;
    static_assert(NITEMS <= 256, "one staging item per thread");
    // The weight fragments live in REGISTERS (9 NOCT K-steps x (hi, lo) x 4 VGPRs = 72 NOCT): a wave re-read all of them from LDS for
    // every output plane -- half of the kernel's LDS traffic, and LDS bandwidth is what bounds it (per workgroup and plane 295 KB
    // of fragment reads = 2,300 clocks at 128 B/clk against 860 clocks of MFMA; measured 2,100).  Without the 37 KB weight image a
    // workgroup needs 38 KB of LDS and occupancy is set by registers.
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[3 * SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[3 * SLOT];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w, D = a.zcount;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * 16, y0 = ty_ * TR;
    const int z0 = blockIdx.y * zt, z1 = min(z0 + zt, D);

    const bool stager = tid < NITEMS;
    const int pq = stager ? tid % NQ : 0, soct = stager ? tid / NQ : 0;
    const int srow = pq / AQ, sqx = pq - srow * AQ;
    const int sgy = y0 - 1 + srow, sgx = x0 - XLEFT + 4 * sqx;
    const int s_off = (stager & (sgy >= 0) & (sgy < h) & (sgx >= 0) & (sgx < w)) ? sgy * w + sgx : -1;
    const int s_col = 4 * sqx - CSH;                                   // slot column of the quad's first pixel (may be -2 / 18)
    const int s_lds = (soct * APIX + srow * AW + s_col) * 8;
    const float* s_src = (soct * 8 < a.ch[0]) ? a.src[0] + (long)(soct * 8) * a.cstride
                                              : a.src[1] + (long)(soct * 8 - a.ch[0]) * a.cstride;

    // planes in flight in registers: ONE (fetched one output plane ahead).  With a second buffer (two planes ahead, the form of the
    // kernel below) the 16-channel instantiation passes 256 registers (one workgroup per CU) and the 8-channel one drops from 3 to 2
    constexpr int NBUF = 1;
    f32x4 pa[NBUF][8];
    auto prefetch = [&](auto buf_t, int z) {
        constexpr int bf = decltype(buf_t)::value;
        const bool ok = (s_off >= 0) & (z >= 0) & (z < D);
        const float* base = ok ? s_src + ((long)z * hw + s_off) : a.zeros;
        const long step = ok ? a.cstride : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) pa[bf][e] = *reinterpret_cast<const f32x4*>(base + (long)e * step);
    };
    auto stash = [&](auto buf_t, int slot) {
        constexpr int bf = decltype(buf_t)::value;
        if (stager) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                if (CSH && (s_col + px < 0 || s_col + px >= AW)) continue;     // columns outside the 20 kept
                bf16x8 hi, lo;
                split_octet(pa[bf], px, hi, lo);
                *reinterpret_cast<bf16x8*>(&lds_ah[slot * SLOT + s_lds + px * 8]) = hi;
                if (!kHiOnly) *reinterpret_cast<bf16x8*>(&lds_al[slot * SLOT + s_lds + px * 8]) = lo;
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    bf16x8 breg_h[NKS], breg_l[NKS];
    {
        const unsigned short* wbf = reinterpret_cast<const unsigned short*>(a.wpack);
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            breg_h[s_] = *reinterpret_cast<const bf16x8*>(wbf + ((long)(s_ * 2 + 0) * 64 + lane) * 8);
            if (!kHiOnly) breg_l[s_] = *reinterpret_cast<const bf16x8*>(wbf + ((long)(s_ * 2 + 1) * 64 + lane) * 8);
        }
    }
    // item 4 s + lk = (dz, dy, dx, oct): element offset inside a slot, dz in the low bits
    int kconst[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int item = 4 * s_ + lk;
        const int oct = item % NOCT, dt = item / NOCT;
        const int dz = dt / 12, rem = dt - dz * 12;
        kconst[s_] = ((oct * APIX + (rem / 3) * AW + rem % 3) * 8) * 4 + dz;
    }
    const int lane_base = ((wv * MR) * AW + li + XOFF) * 8;

    f32x4 acc[MP];
    const f32x4 rbias = *reinterpret_cast<const f32x4*>(a.bias + 4 * (lk & 1));   // once per workgroup, not per plane
    using BL = std::integral_constant<int, NBUF - 1>;                  // the second buffer, or the only one
    prefetch(B0{}, z0 - 1);
    if (NBUF == 2) prefetch(BL{}, z0);
    stash(B0{}, 0);
    prefetch(B0{}, NBUF == 2 ? z0 + 1 : z0);
    stash(BL{}, 1);
    prefetch(BL{}, NBUF == 2 ? z0 + 2 : z0 + 1);
    int rot = 0;
    auto plane = [&](auto bt, int z) {
        int s2 = rot + 2;
        s2 -= (s2 >= 3) ? 3 : 0;
        stash(bt, s2);
        __syncthreads();
        if (z + NBUF < z1) prefetch(bt, z + NBUF + 1);
        int s1 = rot + 1;
        s1 -= (s1 >= 3) ? 3 : 0;
        const int rb0 = rot * SLOT, rb1 = s1 * SLOT, rb2 = s2 * SLOT;
#pragma unroll
        for (int m = 0; m < MP; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            const int dz = kconst[s_] & 3;
            const int off = lane_base + (kconst[s_] >> 2) + (dz == 0 ? rb0 : (dz == 1 ? rb1 : rb2));
            const bf16x8 bh = breg_h[s_];
            const bf16x8 bl = kHiOnly ? bh : breg_l[s_];
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&lds_ah[off + m * 2 * AW * 8]);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[m], 0, 0, 0);
                if (!kHiOnly) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&lds_al[off + m * 2 * AW * 8]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[m], 0, 0, 0);
                }
            }
        }
        // lane (li, lk): lk < 2 -> row y, channels 4 lk ..; lk >= 2 -> row y + 1, channels 4 (lk - 2) ..
        const int x = x0 + li;
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            const int y = y0 + wv * MR + 2 * m + (lk >> 1);
            if (y >= h || x >= w) continue;
            conv_epilogue_store_t<EFFI_EPI_PLAIN>(a, acc[m], 4 * (lk & 1), (long)y * w + x, hw, z, 0, 0, &rbias);
        }
        __syncthreads();
        rot = s1;
    };
    for (int z = z0; z < z1; z += 2) {
        plane(B0{}, z);
        if (z + 1 < z1) plane(BL{}, z + 1);
    }
}

template <int NOCT, int MR>
__global__ __launch_bounds__(256) void conv3d_roll_rp_bf16x3_kernel(const Conv2dArgs a, int tiles_x, int ntiles, int zt) {
    conv3d_roll_rp_bf16x3_body<NOCT, MR>(a, tiles_x, ntiles, zt);
}

template <int NOCT, int MR>
__global__ __launch_bounds__(256) void conv3d_roll_rp_bf16x3_pair_kernel(const Conv2dArgs a, const Conv2dArgs b, int tiles_x, int ntiles,
                                                                         int zt) {
    Conv2dArgs c = a;
    if (blockIdx.z) {
#pragma unroll
        for (int i = 0; i < EFFI_MAX_SRC; ++i) c.src[i] = b.src[i];
        c.wpack = b.wpack;
        c.bias = b.bias;
        c.out0 = b.out0;
    }
    conv3d_roll_rp_bf16x3_body<NOCT, MR>(c, tiles_x, ntiles, zt);
}

// ------------------------------------------------------------------------------------------------
// Cross-scale block (models/module.py:501-516), layers conv0 | conv_cost -> conv1 in ONE kernel: the 16 input channels of conv1 --
// relu(conv0(x)) (1 -> 8 channels, stride (1,2,2)) and relu(conv_cost(prior)) (1 -> 8) -- are GENERATED plane by plane into the
// rolling window of the row-pair kernel above instead of being written by two launches (conv3d_c1to8_kernel, conv3d.hip) and read
// back: per block 2 x 8 channels x D x h x w x 4 B less written and read (stage 3 of cfg3: 120 MB of 180 MB per launch), two
// launches less, and no global load inside the plane loop (the rolling kernel's plane step was bound by the latency of its fetch).
//   * a workgroup owns a 16 x 8 tile of coarse pixels (MR = 2) and ALL D planes (D <= 14: the prior tile of every plane sits in LDS,
//     the fine volume's tile in a ring of four planes fetched one plane step ahead);
//   * generation: 200 work items = (row, pixel pair, octet) of the 10 x 20 slot region; per item the 27 taps in the order of
//     conv3d_c1to8_kernel ((ky, kx) outer, kd inner, fmaf), + bias, relu, zero outside the map / the volume (conv1's padding), then
//     the (hi, lo) split of the staging code above: the slot contents are BITWISE those of the three-launch path, and so is conv1;
//   * the matrix part (K order, weight fragments in registers, epilogue) is that of conv3d_roll_rp_bf16x3_body<2, 2>.
// blockIdx.z picks the block (CSP_R[s] / CSP_C[s] share x).
// ------------------------------------------------------------------------------------------------
struct CspGenCall {
    const float* prior;                   // [D][h][w]
    const float* w0; const float* b0;     // conv0:     [27][8], [8]   (BatchNorm folded)
    const float* wc; const float* bc;     // conv_cost: [27][8], [8]
    const float* w1; const float* b1;     // conv1: row-pair operand (packing.pack_conv3d_roll_bf16x3), bias [16]
    float* out;                           // [8][D][h][w]
};
constexpr int kCspMaxD = 14;

__global__ __launch_bounds__(256) void csp_gen_roll_rp_kernel(const float* __restrict__ xf, int H, int W, CspGenCall ca, CspGenCall cb,
                                                              int D, int h, int w, int tiles_x, int ntiles) {
    constexpr int MR = 2, NOCT = 2, TR = 4 * MR, AR = TR + 2, AW = 20, XOFF = 1;
    constexpr int APIX = AR * AW, NIT = 36 * NOCT, NKS = NIT / 4, SLOT = NOCT * APIX * 8, MP = MR / 2;
    constexpr int FR = 2 * AR + 1, FC = 2 * AW + 1, FPL = FR * FC, NFL = (FPL + 255) / 256;       // fine tile of a plane: 21 x 41
    constexpr int PR = AR + 2, PC = AW + 2, PPL = PR * PC;                                       // prior tile of a plane: 12 x 22
    constexpr int NGEN = AR * (AW / 2) * 2;                                                      // generation items
    static_assert(NGEN <= 256, "one generation item per thread");
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[3 * SLOT];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[kHiOnly ? 8 : 3 * SLOT];
    __shared__ float s_fine[4 * FPL];
    __shared__ float s_prior[(kCspMaxD + 2) * PPL];
    __shared__ __attribute__((aligned(16))) float s_w[2][27 * 8 + 8];

    const CspGenCall& c = blockIdx.z ? cb : ca;
    const float* __restrict__ prior = c.prior;
    float* __restrict__ out = c.out;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * 16, y0 = ty_ * TR;

    // ---- once per workgroup: weights of the two generated layers, prior tile of every plane (zero planes in front and behind) ----
    if (tid < 27 * 8) {
        s_w[0][tid] = c.w0[tid];
        s_w[1][tid] = c.wc[tid];
    }
    if (tid < 8) {
        s_w[0][27 * 8 + tid] = c.b0[tid];
        s_w[1][27 * 8 + tid] = c.bc[tid];
    }
    for (int e = tid; e < (D + 2) * PPL; e += 256) {
        const int pl = e / PPL, r_ = e - pl * PPL;
        const int py = r_ / PC, px = r_ - py * PC;
        const int gz = pl - 1, gy = y0 - 2 + py, gx = x0 - 3 + px;
        const bool ok = (gz >= 0) & (gz < D) & (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w);
        s_prior[e] = ok ? prior[(long)gz * hw + (long)gy * w + gx] : 0.0f;
    }
    // fine tile: element e of a plane -> offset inside the plane (or -1: padding)
    int foff[NFL];
#pragma unroll
    for (int k = 0; k < NFL; ++k) {
        const int e = tid + k * 256;
        const int fy = e / FC, fx = e - fy * FC;
        const int gy = 2 * (y0 - 1) - 1 + fy, gx = 2 * (x0 - 2) - 1 + fx;
        foff[k] = ((e < FPL) & (gy >= 0) & (gy < H) & (gx >= 0) & (gx < W)) ? gy * W + gx : -1;
    }
    const long HW = (long)H * W;
    float ff[NFL];
    auto fine_fetch = [&](int p) {                    // plane p of the fine volume -> registers (any valid address when outside)
        const float* __restrict__ zp = xf + (long)min(max(p, 0), D - 1) * HW;
#pragma unroll
        for (int k = 0; k < NFL; ++k) ff[k] = zp[max(foff[k], 0)];
    };
    auto fine_stash = [&](int p) {                    // registers -> ring slot p & 3, zero where padding
        const bool zok = (p >= 0) & (p < D);
        float* dst = s_fine + (p & 3) * FPL;
#pragma unroll
        for (int k = 0; k < NFL; ++k) {
            const int e = tid + k * 256;
            if (e < FPL) dst[e] = (zok & (foff[k] >= 0)) ? ff[k] : 0.0f;
        }
    };

    // weight fragments of conv1 in registers (see conv3d_roll_rp_bf16x3_body)
    bf16x8 breg_h[NKS], breg_l[NKS];
    {
        const unsigned short* wbf = reinterpret_cast<const unsigned short*>(c.w1);
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            breg_h[s_] = *reinterpret_cast<const bf16x8*>(wbf + ((long)(s_ * 2 + 0) * 64 + lane) * 8);
            if (!kHiOnly) breg_l[s_] = *reinterpret_cast<const bf16x8*>(wbf + ((long)(s_ * 2 + 1) * 64 + lane) * 8);
        }
    }
    int kconst[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int item = 4 * s_ + lk;
        const int oct = item % NOCT, dt = item / NOCT;
        const int dz = dt / 12, rem = dt - dz * 12;
        kconst[s_] = ((oct * APIX + (rem / 3) * AW + rem % 3) * 8) * 4 + dz;
    }
    const int lane_base = ((wv * MR) * AW + li + XOFF) * 8;

    // generation item of this thread: slot row gr, columns 2 gcp / 2 gcp + 1, octet goct (0: conv0 over the fine ring, 1: conv_cost)
    const bool gen = tid < NGEN;
    const int goct = (tid >= NGEN / 2) ? 1 : 0, git = gen ? tid - goct * (NGEN / 2) : 0;
    const int gr = git / (AW / 2), gcp = git - gr * (AW / 2);
    const int ggy = y0 - 1 + gr, ggx = x0 - 2 + 2 * gcp;
    const bool gin0 = gen & (ggy >= 0) & (ggy < h) & (ggx >= 0) & (ggx < w);
    const bool gin1 = gen & (ggy >= 0) & (ggy < h) & (ggx + 1 >= 0) & (ggx + 1 < w);
    // element (ky, kx) of pixel 0 sits at base + ky * rstride + kx; pixel 1 is pstride further
    const int g_rs = goct ? PC : FC, g_ps = goct ? 1 : 2;
    const int g_base = goct ? gr * PC + 2 * gcp : (2 * gr) * FC + 4 * gcp;
    const int g_lds = (goct * APIX + gr * AW + 2 * gcp) * 8;
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto generate = [&](int p) {                      // plane p of the 16 channels -> slot (p + 3) % 3
        const int slot = (p + 3) % 3;
        if (!gen) return;
        unsigned short* dh = lds_ah + slot * SLOT + g_lds;
        unsigned short* dl = lds_al + (kHiOnly ? 0 : slot * SLOT + g_lds);
        if (p < 0 || p >= D) {                        // outside the volume: conv1's zero padding
            const f4 z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            *reinterpret_cast<f4*>(dh) = z4;
            *reinterpret_cast<f4*>(dh + 8) = z4;
            if (!kHiOnly) {
                *reinterpret_cast<f4*>(dl) = z4;
                *reinterpret_cast<f4*>(dl + 8) = z4;
            }
            return;
        }
        // plane p + kd - 1 of the input: fine ring slot (p + kd - 1) & 3, or prior plane index p + kd (index 0 is the zero plane)
        const float* pl[3];
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) pl[kd] = (goct ? s_prior + (p + kd) * PPL : s_fine + ((p + kd - 1) & 3) * FPL) + g_base;
        f2 acc[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[j][q] = f2{0.0f, 0.0f};
        const float* wl = s_w[goct];
#ifndef CSP_ABL
#define CSP_ABL 0
#endif
#pragma unroll 1
        for (int t = 0; t < ((CSP_ABL & 1) ? 1 : 9); ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
            const int o = ky * g_rs + kx;
#pragma unroll
            for (int kd = 0; kd < 3; ++kd) {
                const float v0 = pl[kd][o], v1 = pl[kd][o + g_ps];
                const f4 w0 = *reinterpret_cast<const f4*>(&wl[(kd * 9 + t) * 8]);
                const f4 w1 = *reinterpret_cast<const f4*>(&wl[(kd * 9 + t) * 8 + 4]);
                const f2 wk[4] = {f2{w0[0], w0[1]}, f2{w0[2], w0[3]}, f2{w1[0], w1[1]}, f2{w1[2], w1[3]}};
                const f2 c0 = {v0, v0}, c1 = {v1, v1};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0][q] = __builtin_elementwise_fma(c0, wk[q], acc[0][q]);
                    acc[1][q] = __builtin_elementwise_fma(c1, wk[q], acc[1][q]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool in = j ? gin1 : gin0;
            float v[8];
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const float t_ = fmaxf(acc[j][ch >> 1][ch & 1] + wl[27 * 8 + ch], 0.0f);
                v[ch] = in ? t_ : 0.0f;
            }
            effi_bf16x8_t hi, lo;
            effi_split8(v, hi, lo);
            *reinterpret_cast<effi_bf16x8_t*>(dh + j * 8) = hi;
            if (!kHiOnly) *reinterpret_cast<effi_bf16x8_t*>(dl + j * 8) = lo;
        }
    };

    // ---- prologue: fine planes -1 .. 2 in the ring, generated planes -1 and 0 in their slots ----
    fine_fetch(0);
    fine_stash(-1);                                   // zeros (the values of plane 0 are in flight)
    fine_stash(0);
    fine_fetch(1);
    fine_stash(1);
    fine_fetch(2);
    __syncthreads();                                  // weights, prior tile, fine planes -1 .. 1
    generate(-1);
    generate(0);
    fine_stash(2);
    fine_fetch(3);
    __syncthreads();

    f32x4 acc[MP];
    float bias4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = c.b1[4 * (lk & 1) + r];
    for (int z = 0; z < D; ++z) {
        generate(z + 1);                              // reads fine planes z .. z + 2 / prior planes; writes the slot plane z - 2 used
        __syncthreads();
        const int rb0 = ((z + 2) % 3) * SLOT, rb1 = (z % 3) * SLOT, rb2 = ((z + 1) % 3) * SLOT;      // planes z - 1, z, z + 1
#pragma unroll
        for (int m = 0; m < MP; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s_ = 0; s_ < ((CSP_ABL & 2) ? 1 : NKS); ++s_) {
            const int dz = kconst[s_] & 3;
            const int off = lane_base + (kconst[s_] >> 2) + (dz == 0 ? rb0 : (dz == 1 ? rb1 : rb2));
            const bf16x8 bh = breg_h[s_];
            const bf16x8 bl = kHiOnly ? bh : breg_l[s_];
#pragma unroll
            for (int m = 0; m < MP; ++m) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&lds_ah[off + m * 2 * AW * 8]);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[m], 0, 0, 0);
                if (!kHiOnly) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&lds_al[off + m * 2 * AW * 8]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[m], 0, 0, 0);
                }
            }
        }
        // lane (li, lk): lk < 2 -> row y, channels 4 lk ..; lk >= 2 -> row y + 1, channels 4 (lk - 2) ..
        const int x = x0 + li;
#pragma unroll
        for (int m = 0; m < MP; ++m) {
            const int y = y0 + wv * MR + 2 * m + (lk >> 1);
            if (y >= h || x >= w) continue;
            if ((CSP_ABL & 4) && acc[m][0] != 1.2345e30f) continue;
            float* dst = out + (long)(4 * (lk & 1)) * ((long)D * hw) + (long)z * hw + (long)y * w + x;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(long)r * ((long)D * hw)] = fmaxf(acc[m][r] + bias4[r], 0.0f);
        }
        fine_stash(z + 3);                            // ring slot of plane z - 1, last read by generate(z)
        if (z + 4 < D + 2) fine_fetch(z + 4);
        __syncthreads();                              // slot reads of plane z - 1 done; fine plane z + 3 visible
    }
}

template <int NOCT, int NT, int MR>
__global__ __launch_bounds__(256) void conv3d_roll_bf16x3_kernel(const Conv2dArgs a, int tiles_x, int ntiles, int zt) {
    conv3d_roll_bf16x3_body<NOCT, NT, MR>(a, tiles_x, ntiles, zt);
}

// Two independent convolutions of the same shape in one launch (blockIdx.z picks the argument set): conv1 of the two
// cross-scale blocks of a stage.
template <int NOCT, int NT, int MR>
__global__ __launch_bounds__(256) void conv3d_roll_bf16x3_pair_kernel(const Conv2dArgs a, const Conv2dArgs b, int tiles_x, int ntiles,
                                                                      int zt) {
    // the two calls differ in sources, weights, bias and output only: ONE inlined body behind scalar selects
    Conv2dArgs c = a;
    if (blockIdx.z) {
#pragma unroll
        for (int i = 0; i < EFFI_MAX_SRC; ++i) c.src[i] = b.src[i];
        c.wpack = b.wpack;
        c.bias = b.bias;
        c.out0 = b.out0;
    }
    conv3d_roll_bf16x3_body<NOCT, NT, MR>(c, tiles_x, ntiles, zt);
}

// ------------------------------------------------------------------------------------------------
// Transposed 3-D convolution, kernel 3 / stride 2 / padding 1 / output_padding 1 (U-Net up-sampling levels,
// models/module.py:448-450) on the bf16 matrix cores in split precision.
// Per dimension out[2i] = in[i]*W[1] and out[2i+1] = in[i]*W[2] + in[i+1]*W[0].  With output parity P = (pz,py,px) and input
// neighbour N = (nz,ny,nx) in {0,1}^3 the layer is ONE GEMM per input position: M = 16 input pixels of a row,
// N = cout per parity (8 parity tiles of 16), K = (neighbour, cin); only neighbours N <= P contribute (27 of 64 blocks), and a
// K-step (nz,ny; both nx; 16 channels) is skipped for the parities it cannot reach (18 of 32 K-step x parity pairs remain).
// A tile in LDS: planes z, z+1 x (TR+1) rows x 20 columns x 16 channels (hi/lo bf16); B: the 18 fragments of the chunk.
// Epilogue: the two x-parities of a channel sit in two accumulators of the same lane -> one float2 store per channel row,
// 16 lanes = 128 contiguous bytes of the output row; bias, ReLU and the skip tensor (added after the ReLU) are fused.
// ------------------------------------------------------------------------------------------------
struct DeconvArgs {
    const float* in;
    const void* wpack;       // bf16 [cin/16][18][hi|lo][64][8]
    const float* bias;       // [16]
    const float* skip;       // [cout][2D][2h][2w] or null
    const float* zeros;
    float* out;
    int cin, cout, D, h, w, relu;
};

__host__ __device__ constexpr bool deconv_step_used(int p, int s) { return (s & ~(p >> 1)) == 0; }
__host__ __device__ constexpr int deconv_slot(int p, int s) {
    int n = 0;
    for (int pp = 0; pp < 8; ++pp)
        for (int ss = 0; ss < 4; ++ss) {
            if (pp == p && ss == s) return n;
            if (deconv_step_used(pp, ss)) ++n;
        }
    return n;
}

// HALF (cout <= 8): the two x-parities of a channel share ONE N-tile (row j = px*8 + co), i.e. 4 parity tiles (pz,py), 9 B
// fragments and half the MFMAs; the x-parities then sit in lanes lk and lk+2 of the same store instruction, which together
// still cover 128 contiguous bytes of the output row.
__host__ __device__ constexpr int deconv_slot_half(int pzy, int s) {       // fragment index when only (pz,py) tiles exist
    int n = 0;
    for (int pp = 0; pp < 4; ++pp)
        for (int ss = 0; ss < 4; ++ss) {
            if (pp == pzy && ss == s) return n;
            if (deconv_step_used(2 * pp, ss)) ++n;
        }
    return n;
}

template <int MR, bool ALIGNED, bool HALF>
__global__ __launch_bounds__(256) void deconv3d_s2_bf16x3_kernel(const DeconvArgs a, int tiles_x, int tiles_xy) {
    constexpr int TR = 4 * MR, AR = TR + 1, AW = 20, AQ = 5, APIX = AR * AW, NQ = APIX / 4;
    constexpr int NITEMS = 4 * NQ;                                     // (octet, plane, pixel quad)
    constexpr int NP = HALF ? 4 : 8;                                    // parity tiles
    constexpr int NSLOT = HALF ? 9 : 18, NBF = NSLOT * 2 * 64;         // 16-byte units of B per chunk
    static_assert(NITEMS <= 256, "one staging item per thread");
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[4 * APIX * 8];     // [plane][octet][pixel][8]
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[4 * APIX * 8];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[NBF * 8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w, D = a.D;
    const long hw = (long)h * w, cstride = (long)D * hw;
    // XCD-aware order over (z, tile): workgroups are dealt to the 8 XCDs round-robin, so with the plain (tile, z) grid every XCD's
    // L2 fetched the whole input (PMC: 48.5 MB fetched per launch against 35.5 MB algorithmic, L2 hit 33 %); a contiguous run of
    // (z, tile) per XCD makes each L2 fetch its own z-slab only.
    const int lin = effi_xcd_remap(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int z = lin / (int)gridDim.x;
    const int t = lin - z * (int)gridDim.x;
    const int ty_ = t / tiles_x;
    const int x0 = (t - ty_ * tiles_x) * 16, y0 = ty_ * TR;

    // staging item
    const bool stager = tid < NITEMS;
    const int it = stager ? tid : 0;
    const int quad = it % NQ, pl = (it / NQ) & 1, soct = it / (2 * NQ);
    const int srow = quad / AQ, sqx = quad - srow * AQ;
    const int gy = y0 + srow, gx = x0 + 4 * sqx, gz = z + pl;
    const bool rowin = stager & (gy < h) & (gz < D);
    const long s_off = (long)gz * hw + (long)gy * w + gx;
    const int s_lds = ((pl * 2 + soct) * APIX + srow * AW + 4 * sqx) * 8;

    int koff[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
        const int nz = s_ >> 1, ny = s_ & 1, nx = lk >> 1, oct = lk & 1;
        koff[s_] = ((nz * 2 + oct) * APIX + (wv * MR + ny) * AW + li + nx) * 8;
    }

    f32x4 acc[MR][NP];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[m][p] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const unsigned short* wbf = reinterpret_cast<const unsigned short*>(a.wpack);
    const int nchunks = a.cin / 16;
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch) __syncthreads();
        // ---- A: 8 channels x 4 pixels per stager, padding from the zero page ----
        f32x4 pa[8];
        {
            const float* base = a.in + (long)(ch * 16 + soct * 8) * cstride + s_off;
            if (ALIGNED) {                                             // w % 4 == 0: a quad is inside or outside as a whole
                const bool in = rowin & (gx < w);
                const float* q = in ? base : a.zeros;
                const long step = in ? cstride : 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) pa[e] = *reinterpret_cast<const f32x4*>(q + (long)e * step);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool in = rowin & (gx + i < w);
                    const float* q = in ? base + i : a.zeros;
                    const long step = in ? cstride : 0;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pa[e][i] = q[(long)e * step];
                }
            }
        }
        // ---- B: the chunk's 18 fragments ----
        for (int u = tid; u < NBF; u += 256)
            *reinterpret_cast<f32x4*>(&lds_b[u * 8]) = *reinterpret_cast<const f32x4*>(wbf + ((long)ch * NBF + u) * 8);
        if (stager) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                bf16x8 hi, lo;
                split_octet(pa, px, hi, lo);
                *reinterpret_cast<bf16x8*>(&lds_ah[s_lds + px * 8]) = hi;
                if (!kHiOnly) *reinterpret_cast<bf16x8*>(&lds_al[s_lds + px * 8]) = lo;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            bf16x8 ah[MR], al[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(&lds_ah[koff[s_] + m * AW * 8]);
                if (!kHiOnly) al[m] = *reinterpret_cast<const bf16x8*>(&lds_al[koff[s_] + m * AW * 8]);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int pfull = HALF ? 2 * p : p;                     // (pz,py) decide which K-steps reach a parity
                if (!deconv_step_used(pfull, s_)) continue;
                const int slot = HALF ? deconv_slot_half(p, s_) : deconv_slot(p, s_);   // compile-time after unrolling
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[((slot * 2 + 0) * 64 + lane) * 8]);
                bf16x8 bl = bh;
                if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[((slot * 2 + 1) * 64 + lane) * 8]);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[m], acc[m][p], 0, 0, 0);
                    if (!kHiOnly) {
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah[m], acc[m][p], 0, 0, 0);
                        acc[m][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al[m], acc[m][p], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue ----
    const int x = x0 + li;
    const int Ho = 2 * h, Wo = 2 * w;
    const long ostride = (long)(2 * D) * Ho * Wo;
    if (HALF) {
        // lane = (input pixel li, tile rows 4*lk .. 4*lk+3) with row j = px*8 + co: px = lk >> 1, co = 4*(lk & 1) + r
        const int px = lk >> 1, cb = 4 * (lk & 1);
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = a.bias[cb + r];
        // the skip values of a row's 16 outputs are fetched together BEFORE its stores: load -> add -> store per element is a chain of
        // 16 dependent round trips per row (skip and out may alias as far as the compiler knows, so it keeps that order)
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int y = y0 + wv * MR + m;
            if (y >= h || x >= w) continue;
            float sk[4][4];
#pragma unroll
            for (int pzy = 0; pzy < 4; ++pzy) {
                const long o = ((long)(2 * z + (pzy >> 1)) * Ho + (2 * y + (pzy & 1))) * Wo + 2 * x + px;
#pragma unroll
                for (int r = 0; r < 4; ++r) sk[pzy][r] = (a.skip && cb + r < a.cout) ? a.skip[(long)(cb + r) * ostride + o] : 0.0f;
            }
#pragma unroll
            for (int pzy = 0; pzy < 4; ++pzy) {
                const long o = ((long)(2 * z + (pzy >> 1)) * Ho + (2 * y + (pzy & 1))) * Wo + 2 * x + px;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = cb + r;
                    if (co >= a.cout) break;
                    float v = acc[m][pzy][r] + bv[r];
                    if (a.relu) v = fmaxf(v, 0.0f);
                    const long oo = (long)co * ostride + o;
                    if (a.skip) v += sk[pzy][r];
                    a.out[oo] = v;
                }
            }
        }
        return;
    }
    // lane = (input pixel li, channels 4*lk .. 4*lk+3); parities px = 0/1 pair up into one float2
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = a.bias[4 * lk + r];
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int y = y0 + wv * MR + m;
        if (y >= h || x >= w) continue;
        float2 sk[4][4];                                               // fetched together before the row's stores (see above)
#pragma unroll
        for (int pzy = 0; pzy < 4; ++pzy) {
            const long o = ((long)(2 * z + (pzy >> 1)) * Ho + (2 * y + (pzy & 1))) * Wo + 2 * x;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sk[pzy][r] = (a.skip && 4 * lk + r < a.cout) ? *reinterpret_cast<const float2*>(a.skip + (long)(4 * lk + r) * ostride + o) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int pzy = 0; pzy < 4; ++pzy) {
            const long o = ((long)(2 * z + (pzy >> 1)) * Ho + (2 * y + (pzy & 1))) * Wo + 2 * x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * lk + r;
                if (co >= a.cout) break;
                float v0 = acc[m][HALF ? 0 : 2 * pzy][r] + bv[r], v1 = acc[m][HALF ? 0 : 2 * pzy + 1][r] + bv[r];
                if (a.relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); }
                const long oo = (long)co * ostride + o;
                if (a.skip) {
                    v0 += sk[pzy][r].x;
                    v1 += sk[pzy][r].y;
                }
                *reinterpret_cast<float2*>(a.out + oo) = make_float2(v0, v1);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 7x7 convolution of a single-channel map (convd1, models/update.py:76,90) + ReLU; vector ALUs.
// ------------------------------------------------------------------------------------------------
// A workgroup computes 16 output channels (blockIdx.z = channel group) of a 32 x 8 pixel tile: the coarse stages have few
// pixels and 32 / 48 channels, so the channel groups are what fills the chip there and what keeps the per-thread FMA chain
// short (784 instead of 2352 at 48 channels).  COUT (the weight row stride) is a template parameter so that every weight is a
// scalar load at a compile-time offset from the group's base.
template <int COUT>
__global__ __launch_bounds__(256) void conv2d_c1k7_relu_kernel(const float* __restrict__ in,
                                                               const float* __restrict__ wgt,
                                                               const float* __restrict__ bias, int h, int w,
                                                               float* __restrict__ out) {
    effi_c1k7_relu_tile<COUT>(in, wgt, bias, h, w, out, blockIdx.x, blockIdx.y, blockIdx.z);     // common.hpp
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolution with ONE output channel (depth head conv2, models/update.py:15,21): an MFMA tile would
// be 15/16 padding, so this runs on the vector ALUs.  64x16 pixel tile, 4 pixels per thread along x (each
// LDS row segment of 6 values feeds 12 FMAs), 4 input channels per LDS chunk with register prefetch.
// EPI_PLAIN: out0 = act(conv + bias); EPI_HEAD: out0 = aux0 + tanh(conv + bias), out1 = scaled depth.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void conv2d_cout1_k3_kernel(const Conv2dArgs a, const float* __restrict__ wraw) {
    constexpr int TXP = 64, TYP = 16, IW = TXP + 2, IH = TYP + 2, PSZ = IW * IH, CC = 4;
    constexpr int NPL = (PSZ + 255) / 256;                         // 5
    __shared__ float tile[CC * PSZ];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * TXP, y0 = blockIdx.y * TYP;
    const int h = a.h, w = a.w;
    const long hw = (long)h * w;
    int poff[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int e = tid + k * 256;
        const int ly = e / IW, lx = e - ly * IW;
        const int gy = y0 - 1 + ly, gx = x0 - 1 + lx;
        poff[k] = ((e < PSZ) & (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w)) ? gy * w + gx : -1;
    }
    float pf[CC][NPL];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            int cg = c0 + c;
            const bool cok = cg < a.cin;
            const float* p = a.src[0];
            if (cok && cg >= a.ch[0]) {
                cg -= a.ch[0];
                p = a.src[1];
                if (cg >= a.ch[1]) { cg -= a.ch[1]; p = a.src[2]; }
            }
            p += cok ? (long)cg * hw : 0;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const float t = p[max(poff[k], 0)];
                pf[c][k] = (cok & (poff[k] >= 0)) ? t : 0.0f;
            }
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                const int e = tid + k * 256;
                if (e < PSZ) tile[c * PSZ + e] = pf[c][k];
            }
    };
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    prefetch(0);
    stash();
    __syncthreads();
    for (int c0 = 0; c0 < a.cin; c0 += CC) {
        const bool more = (c0 + CC < a.cin);
        if (more) prefetch(c0 + CC);
        const int ccn = min(CC, a.cin - c0);
        for (int c = 0; c < ccn; ++c) {
            const float* __restrict__ wc = wraw + (long)(c0 + c) * 9;       // [cin][3][3] of the single output channel
            const float* tc = tile + c * PSZ + ty * IW + 4 * tx;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float r[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) r[i] = tc[ky * IW + i];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wc[ky * 3 + kx];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(r[i + kx], wv, acc[i]);
                }
            }
        }
        if (more) {
            __syncthreads();
            stash();
            __syncthreads();
        }
    }
    const int y = y0 + ty, x = x0 + 4 * tx;
    if (y >= h || x >= w) return;
    const long pix = (long)y * w + x;
    const float b = a.bias[0];
    float lo = 0.0f, hi = 0.0f;
    if (EPI == EFFI_EPI_HEAD) { lo = a.disp_range[0]; hi = a.disp_range[a.n_range - 1]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (x + i >= w) break;
        float v = acc[i] + b;
        if (EPI == EFFI_EPI_HEAD) {
            v = a.aux0[pix + i] + tanhf(v);
            a.out1[pix + i] = effi_inv_to_depth(v, lo, hi);
        } else {
            v = apply_act(v, a.act);
        }
        a.out0[pix + i] = v;
    }
}

// Measured on MI355X (tools/bench_conv2d.py): 8-channel chunks with one tile per workgroup
// beat 16-channel chunks, persistent workgroups and a same-accumulator ("run-ordered") MFMA schedule -- each of
// those costs registers, i.e. waves per SIMD, and the kernel is bound by its LDS->MFMA phase, not by global
// latency -- so only that configuration is instantiated; CC / PERSIST stay as parameters for future A/B runs.
template <int KS, int NT, int MR, int EPI>
int launch2d_v2(const Conv2dArgs& a, hipStream_t st) {
    const int tiles_x = effi_cdiv(a.w, 16), ntiles = tiles_x * effi_cdiv(a.h, 4 * MR);
    hipLaunchKernelGGL((conv2d_mfma_v2_kernel<KS, NT, MR, EPI, 8, false>), dim3(ntiles), dim3(256), 0, st, a, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

template <int KS, int NT, int EPI>
int launch2d(const Conv2dArgs& a, hipStream_t st) {
    if ((a.w & 3) == 0) {
        // rows per wave: the largest of {4, 2, 1} that still yields >= 2 workgroups per CU (256 CUs)
        const long cols = effi_cdiv(a.w, 16);
        int mr;
        if (cols * effi_cdiv(a.h, 16) >= 512 || (KS == 1 && cols * effi_cdiv(a.h, 16) >= 256)) mr = 4;
        else if (cols * effi_cdiv(a.h, 8) >= 512 || KS == 1) mr = 2;
        else mr = 1;
        if (KS == 1 && mr == 1) mr = 2;
        if (mr == 4) return launch2d_v2<KS, NT, 4, EPI>(a, st);
        if (mr == 2) return launch2d_v2<KS, NT, 2, EPI>(a, st);
        return launch2d_v2<KS, NT, (KS == 3 ? 1 : 2), EPI>(a, st);
    }
    dim3 grid(effi_cdiv(a.w, 16), effi_cdiv(a.h, 16));
    hipLaunchKernelGGL((conv2d_mfma_kernel<KS, NT, EPI>), grid, dim3(256), 0, st, a);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

template <int KS, int EPI>
int dispatch_nt(const Conv2dArgs& a, int nt, hipStream_t st) {
    switch (nt) {
        case 1: return launch2d<KS, 1, EPI>(a, st);
        case 2: return launch2d<KS, 2, EPI>(a, st);
        case 3: return launch2d<KS, 3, EPI>(a, st);
        case 4: return launch2d<KS, 4, EPI>(a, st);
        case 6: return launch2d<KS, 6, EPI>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

}  // namespace

// ---- 3-D convolution (k3, stride 1, pad 1) as z-batched 2-D convolutions on the matrix cores -----------------
template <int NT, int MR, bool ALIGNED>
static int launch3d_planes(const Conv2dArgs& a, hipStream_t st) {
    const int tiles_x = effi_cdiv(a.w, 16), ntiles = tiles_x * effi_cdiv(a.h, 4 * MR);
    hipLaunchKernelGGL((conv2d_mfma_v2_kernel<3, NT, MR, EFFI_EPI_PLAIN, 8, false, ALIGNED>), dim3(ntiles, a.zcount), dim3(256), 0, st,
                       a, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

template <int NT>
static int dispatch3d_planes(const Conv2dArgs& a, hipStream_t st) {
    const long cols = effi_cdiv(a.w, 16);
    const bool al = (a.w & 3) == 0;
    // rows per wave: keep >= 2 workgroups per CU over all planes
    if (cols * effi_cdiv(a.h, 16) * a.zcount >= 512) return al ? launch3d_planes<NT, 4, true>(a, st) : launch3d_planes<NT, 4, false>(a, st);
    if (cols * effi_cdiv(a.h, 8) * a.zcount >= 512) return al ? launch3d_planes<NT, 2, true>(a, st) : launch3d_planes<NT, 2, false>(a, st);
    return al ? launch3d_planes<NT, 1, true>(a, st) : launch3d_planes<NT, 1, false>(a, st);
}

#ifndef EFFI_BF16_ONLY
extern "C" int effi_conv3d_k3s1_mfma_f32(const float* in, int cin, const float* wpack, const float* bias, int cout, int D,
                                         int h, int w, int relu, float* out, effi_stream_t stream) {
    if (!in || !wpack || !bias || !out || cin < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (cout != 16 && cout != 32) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {       // source s = plane z + s - 1 of the same tensor
        a.src[i] = in;
        a.ch[i] = cin;
    }
    a.cin = 3 * cin;
    a.kgroups = (a.cin + 3) / 4;
    a.wpack = wpack;
    a.bias = bias;
    a.cout = cout;
    a.h = h;
    a.w = w;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = (long)D * h * w;
    a.ostride = (long)D * h * w;
    a.zcount = a.zin = D;
    a.hin = h;
    a.win = w;
    return cout == 16 ? dispatch3d_planes<1>(a, effi_s(stream)) : dispatch3d_planes<2>(a, effi_s(stream));
}
#endif

// Stride-2 form (models/module.py:442,445: conv2 8->16, conv4 16->32): output plane z reads input planes 2z-1, 2z, 2z+1.
template <int NT, int MR, bool ALIGNED>
static int launch3d_planes_s2(const Conv2dArgs& a, hipStream_t st) {
    const int tiles_x = effi_cdiv(a.w, 16), ntiles = tiles_x * effi_cdiv(a.h, 4 * MR);
    hipLaunchKernelGGL((conv2d_mfma_v2_kernel<3, NT, MR, EFFI_EPI_PLAIN, 8, false, ALIGNED, 2>), dim3(ntiles, a.zcount), dim3(256), 0,
                       st, a, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

template <int NT>
static int dispatch3d_planes_s2(const Conv2dArgs& a, hipStream_t st) {
    const long cols = effi_cdiv(a.w, 16);
    const bool al = (a.win & 3) == 0 && (a.w & 3) == 0;
    if (cols * effi_cdiv(a.h, 8) * a.zcount >= 512) return al ? launch3d_planes_s2<NT, 2, true>(a, st) : launch3d_planes_s2<NT, 2, false>(a, st);
    return al ? launch3d_planes_s2<NT, 1, true>(a, st) : launch3d_planes_s2<NT, 1, false>(a, st);
}

#ifndef EFFI_BF16_ONLY
extern "C" int effi_conv3d_k3s2_mfma_f32(const float* in, int cin, const float* wpack, const float* bias, int cout, int D,
                                         int h, int w, int relu, float* out, effi_stream_t stream) {
    if (!in || !wpack || !bias || !out || cin < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (cout != 16 && cout != 32) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {       // source s = input plane 2z + s - 1 of the same tensor
        a.src[i] = in;
        a.ch[i] = cin;
    }
    a.cin = 3 * cin;
    a.kgroups = (a.cin + 3) / 4;
    a.wpack = wpack;
    a.bias = bias;
    a.cout = cout;
    a.hin = h;
    a.win = w;
    a.h = (h - 1) / 2 + 1;
    a.w = (w - 1) / 2 + 1;
    a.zin = D;
    a.zcount = (D - 1) / 2 + 1;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = (long)D * h * w;
    a.ostride = (long)a.zcount * a.h * a.w;
    return cout == 16 ? dispatch3d_planes_s2<1>(a, effi_s(stream)) : dispatch3d_planes_s2<2>(a, effi_s(stream));
}
#endif

#ifndef EFFI_BF16_ONLY
extern "C" int effi_conv2d_f32(const float* const* srcs, const int* src_channels, int n_src, const float* wpack,
                               const float* bias, int cout, int ks, int h, int w, int epilogue, int act,
                               const float* aux0, const float* aux1, const float* disp_range, int n_range,
                               float* out0, float* out1, effi_stream_t stream) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack || !bias || !out0) return EFFI_ERR_BADARG;
    if (cout < 1 || h < 1 || w < 1 || (ks != 1 && ks != 3)) return EFFI_ERR_BADARG;
    Conv2dArgs a;
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : nullptr;
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        a.cin += a.ch[i];
    }
    a.kgroups = (a.cin + 3) / 4;
    a.wpack = wpack;
    a.bias = bias;
    a.cout = cout;
    a.h = h;
    a.w = w;
    a.act = act;
    a.hd = cout / 2;
    a.aux0 = aux0;
    a.aux1 = aux1;
    a.disp_range = disp_range;
    a.n_range = n_range;
    a.out0 = out0;
    a.out1 = out1;
    a.cstride = (long)h * w;
    a.ostride = (long)h * w;
    a.zcount = 0;
    a.hin = h;
    a.win = w;
    const int nt = (cout + 15) / 16;
    hipStream_t st = effi_s(stream);
    if (cout == 1 && ks == 3 && (long)h * w >= 262144 && (epilogue == EFFI_EPI_PLAIN || epilogue == EFFI_EPI_HEAD)) {
        // single output channel on a large map: vector-ALU kernel (an MFMA tile would be 15/16 padding);
        // it reads plain [cin][9] weights that the host appends behind the packed block (packing.py).
        // Small maps keep the MFMA kernel, whose finer tiling still fills the chip.
        if (epilogue == EFFI_EPI_HEAD && (!aux0 || !out1 || !disp_range || n_range < 2)) return EFFI_ERR_BADARG;
        const float* wraw = wpack + (long)a.kgroups * 9 * 64;
        dim3 grid(effi_cdiv(w, 64), effi_cdiv(h, 16));
        if (epilogue == EFFI_EPI_HEAD)
            hipLaunchKernelGGL(conv2d_cout1_k3_kernel<EFFI_EPI_HEAD>, grid, dim3(256), 0, st, a, wraw);
        else
            hipLaunchKernelGGL(conv2d_cout1_k3_kernel<EFFI_EPI_PLAIN>, grid, dim3(256), 0, st, a, wraw);
        return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
    }
    switch (epilogue) {
        case EFFI_EPI_PLAIN:
            if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
            return ks == 3 ? dispatch_nt<3, EFFI_EPI_PLAIN>(a, nt, st) : dispatch_nt<1, EFFI_EPI_PLAIN>(a, nt, st);
        case EFFI_EPI_ADD_UP2:
            if (ks != 1 || !aux0 || (h & 1) || (w & 3)) return EFFI_ERR_BADARG;
            return dispatch_nt<1, EFFI_EPI_ADD_UP2>(a, nt, st);
        case EFFI_EPI_NHWC:
            if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH || (w & 3)) return EFFI_ERR_BADARG;
            return ks == 3 ? dispatch_nt<3, EFFI_EPI_NHWC>(a, nt, st) : dispatch_nt<1, EFFI_EPI_NHWC>(a, nt, st);
        case EFFI_EPI_GRU_ZR:
            if (ks != 3 || !aux0 || !out1 || (cout % 32) != 0) return EFFI_ERR_BADARG;
            if (nt == 2) return launch2d<3, 2, EFFI_EPI_GRU_ZR>(a, st);
            if (nt == 4) return launch2d<3, 4, EFFI_EPI_GRU_ZR>(a, st);
            if (nt == 6) return launch2d<3, 6, EFFI_EPI_GRU_ZR>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        case EFFI_EPI_GRU_Q:
            if (ks != 3 || !aux0 || !aux1 || (cout % 16) != 0) return EFFI_ERR_BADARG;
            if (nt == 1) return launch2d<3, 1, EFFI_EPI_GRU_Q>(a, st);
            if (nt == 2) return launch2d<3, 2, EFFI_EPI_GRU_Q>(a, st);
            if (nt == 3) return launch2d<3, 3, EFFI_EPI_GRU_Q>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        case EFFI_EPI_HEAD:
            if (ks != 3 || cout != 1 || !aux0 || !out1 || !disp_range || n_range < 2) return EFFI_ERR_BADARG;
            return launch2d<3, 1, EFFI_EPI_HEAD>(a, st);
        default:
            return EFFI_ERR_BADARG;
    }
}
#endif

// ---- 5x5 stride-2 convolution (feature pyramid down-sampling) ---------------------------------------------------
template <int NT, int MR, bool ALIGNED>
static int launch_k5s2(const Conv2dArgs& a, hipStream_t st) {
    const int tiles_x = effi_cdiv(a.w, 16), ntiles = tiles_x * effi_cdiv(a.h, 4 * MR);
    hipLaunchKernelGGL((conv2d_mfma_v2_kernel<5, NT, MR, EFFI_EPI_PLAIN, 4, false, ALIGNED, 2>), dim3(ntiles), dim3(256), 0, st, a,
                       tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

template <int NT>
static int dispatch_k5s2(const Conv2dArgs& a, hipStream_t st) {
    const long cols = effi_cdiv(a.w, 16);
    const bool al = (a.win & 3) == 0 && (a.w & 3) == 0;
    if (cols * effi_cdiv(a.h, 8) >= 512) return al ? launch_k5s2<NT, 2, true>(a, st) : launch_k5s2<NT, 2, false>(a, st);
    return al ? launch_k5s2<NT, 1, true>(a, st) : launch_k5s2<NT, 1, false>(a, st);
}

#ifndef EFFI_BF16_ONLY
extern "C" int effi_conv2d_k5s2_f32(const float* in, int cin, const float* wpack, const float* bias, int cout, int hin,
                                    int win, int act, float* out, effi_stream_t stream) {
    if (!in || !wpack || !bias || !out || cin < 1 || cout < 1 || hin < 1 || win < 1) return EFFI_ERR_BADARG;
    if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
    Conv2dArgs a;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i == 0) ? in : nullptr;
        a.ch[i] = (i == 0) ? cin : 0;
    }
    a.cin = cin;
    a.kgroups = (cin + 3) / 4;
    a.wpack = wpack;
    a.bias = bias;
    a.cout = cout;
    a.hin = hin;
    a.win = win;
    a.h = (hin - 1) / 2 + 1;
    a.w = (win - 1) / 2 + 1;
    a.act = act;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = (long)hin * win;
    a.ostride = (long)a.h * a.w;
    a.zcount = 0;
    hipStream_t st = effi_s(stream);
    switch ((cout + 15) / 16) {
        case 1: return dispatch_k5s2<1>(a, st);
        case 2: return dispatch_k5s2<2>(a, st);
        case 4: return dispatch_k5s2<4>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}
#endif

// ---- 5x5 stride-2 convolution in split precision (the pyramid's down-sampling layers) ---------------------------
// out[y][x] = sum W[ky][kx] in[2y + ky - 2][2x + kx - 2]: the fp32-MFMA form above is bound by the fp32 MFMA rate (K = 25 cin in steps of 4);
// here the products are 3 bf16 MFMAs of K = 32 as in the 3x3 kernels.
//   * chunk = 8 input channels (one octet), K item = tap (ky, kx): 25 items = 7 K-steps (the last three are zero);
//   * the input region of a 16 x (4 MR) output tile is (8 MR + 3) rows x 40 columns (float4-aligned from 2 x0 - 4).  It is stored with
//     the columns DE-INTERLEAVED by parity, [row][parity][20 slots][8 ch]: the 16 pixels of a lane group then read 16 consecutive slots
//     for every tap (slot = li + 1 + (kx >> 1) of parity kx & 1), conflict-free, although they are 2 input columns apart;
//   * cout > 32 is split over blockIdx.y in groups of NT = 2 N-tiles (the weight fragments of a chunk are 14 KB per N-tile).
namespace {
// KS = 3 with ZB: the stride-(2,2,2) 3-D convolution of the U-Net (conv2 / conv4) as z-batched stride-2 2-D convolutions: blockIdx.z =
// output plane z, chunk = (input plane 2z + dz - 1, octet), 9 taps = 3 K-steps per chunk; a.cin = channels per plane.
template <int KS, int NT, int MR, bool ZB>
__global__ __launch_bounds__(256) void conv2d_s2_bf16x3_kernel(const Conv2dArgs a, int tiles_x, int ntiles) {
    constexpr int PAD = KS / 2, NTAP = KS * KS;
    constexpr int TR = 4 * MR, IR = 2 * TR + KS - 2, NQ = 10, NSLOT = 20, NKS = (NTAP + 3) / 4;
    constexpr int ROWE = 2 * NSLOT * 8;                                // bf16 elements per staged input row (both parities)
    constexpr int NITEMS = IR * NQ;                                    // (row, pixel quad) staging items
    constexpr int NIT = (NITEMS + 255) / 256;
    constexpr int NBF = NKS * NT * 2 * 64, NB4 = (NBF + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[IR * ROWE];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[IR * ROWE];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[NB4 * 256 * 8];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w, hin = a.hin, win = a.win;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(blockIdx.x, gridDim.x);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * 16, y0 = ty_ * TR;
    const int ng = blockIdx.y;                                         // group of NT output tiles
    const unsigned short* wbf = reinterpret_cast<const unsigned short*>(a.wpack);
    const int noct = (a.cin + 7) >> 3;
    const int nchunks = ZB ? 3 * noct : noct;
    const int ngroups = gridDim.y;
    const int zpl = ZB ? blockIdx.z : 0;

    f32x4 pa[NIT][8];
    auto prefetch = [&](int ch) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * 256;
            const int r = item / NQ, q = item - r * NQ;
            const int gy = 2 * y0 - PAD + r, gx = 2 * x0 - 4 + 4 * q;
            bool in = (item < NITEMS) & (gy >= 0) & (gy < hin) & (gx >= 0) & (gx < win);
            const int dz = ZB ? ch / noct : 0;
            const int cb = (ZB ? ch - dz * noct : ch) * 8;
            const int zz = 2 * zpl + dz - 1;
            if (ZB) in &= (zz >= 0) & (zz < a.zin);
            const int emax = min(a.cin - cb, 8) - 1;
            const float* qp = in ? a.src[0] + ((long)cb * a.cstride + (ZB ? (long)zz * hin * win : 0L) + (long)gy * win + gx) : a.zeros;
            const long step = in ? a.cstride : 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                pa[it][e] = *reinterpret_cast<const f32x4*>(qp);
                qp += (e < emax) ? step : 0;
            }
        }
    };
    auto stash = [&](int ch) {
        f32x4 tb[NB4];
#pragma unroll
        for (int j = 0; j < NB4; ++j) {
            const int u = min(tid + j * 256, NBF - 1);
            tb[j] = *reinterpret_cast<const f32x4*>(wbf + (((long)ch * ngroups + ng) * NBF + u) * 8);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = tid + it * 256;
            if (item < NITEMS) {
                const int r = item / NQ, q = item - r * NQ;
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    bf16x8 hi, lo;
                    split_octet(pa[it], px, hi, lo);
                    const int o = r * ROWE + ((px & 1) * NSLOT + 2 * q + (px >> 1)) * 8;
                    *reinterpret_cast<bf16x8*>(&lds_ah[o]) = hi;
                    if (!kHiOnly) *reinterpret_cast<bf16x8*>(&lds_al[o]) = lo;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB4; ++j) *reinterpret_cast<f32x4*>(&lds_b[(tid + j * 256) * 8]) = tb[j];
    };

    int koff[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int tap = min(4 * s_ + lk, NTAP - 1);                     // the items past the last tap are padding (B is zero there)
        const int ky = tap / KS, kx = tap - ky * KS;
        const int off = kx - PAD + 4;                                   // column offset from the aligned origin 2 x0 - 4 for pixel 0
        koff[s_] = ky * ROWE + ((off & 1) * NSLOT + (off >> 1)) * 8;
    }
    const int lane_base = (2 * wv * MR) * ROWE + li * 8;

    f32x4 acc[MR][NT];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    prefetch(0);
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            bf16x8 ah[MR], al[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                ah[m] = *reinterpret_cast<const bf16x8*>(&lds_ah[lane_base + koff[s_] + m * 2 * ROWE]);
                if (!kHiOnly) al[m] = *reinterpret_cast<const bf16x8*>(&lds_al[lane_base + koff[s_] + m * 2 * ROWE]);
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
        if (ch + 1 < nchunks) {
            __syncthreads();
            stash(ch + 1);
            __syncthreads();
        }
    }
    const int x = x0 + li;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int y = y0 + wv * MR + m;
        if (y >= h || x >= w) continue;
        const long pix = (long)y * w + x;
#pragma unroll
        for (int n = 0; n < NT; ++n) conv_epilogue_store_t<EFFI_EPI_PLAIN>(a, acc[m][n], (ng * NT + n) * 16 + 4 * lk, pix, hw, zpl);
    }
}

}  // namespace

template <int KS, int NT, bool ZB>
static int launch_s2_x3(const Conv2dArgs& a, int ngroups, hipStream_t st) {
    const int cols = effi_cdiv(a.w, 16);
    const unsigned planes = ZB ? (unsigned)a.zcount : 1u;
    if ((long)cols * effi_cdiv(a.h, 8) * ngroups * planes >= 400) {
        const int ntiles = cols * effi_cdiv(a.h, 8);
        hipLaunchKernelGGL((conv2d_s2_bf16x3_kernel<KS, NT, 2, ZB>), dim3(ntiles, ngroups, planes), dim3(256), 0, st, a, cols, ntiles);
    } else {
        const int ntiles = cols * effi_cdiv(a.h, 4);
        hipLaunchKernelGGL((conv2d_s2_bf16x3_kernel<KS, NT, 1, ZB>), dim3(ntiles, ngroups, planes), dim3(256), 0, st, a, cols, ntiles);
    }
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

extern "C" int EFFI_FN(effi_conv2d_k5s2_bf16x3_f32)(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int hin,
                                           int win, int act, float* out, effi_stream_t stream) {
    if (!in || !wpack_bf16 || !bias || !out || cin < 1 || cout < 1 || hin < 1 || win < 1) return EFFI_ERR_BADARG;
    if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
    if ((win & 3) || cout > 128) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = in;
        a.ch[i] = (i == 0) ? cin : 0;
    }
    a.cin = cin;
    a.kgroups = 0;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.hin = hin;
    a.win = win;
    a.h = (hin - 1) / 2 + 1;
    a.w = (win - 1) / 2 + 1;
    a.act = act;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = (long)hin * win;
    a.ostride = (long)a.h * a.w;
    a.zcount = a.zin = 0;
    hipStream_t st = effi_s(stream);
    const int nt = (cout + 15) / 16;
    if (nt == 1) return launch_s2_x3<5, 1, false>(a, 1, st);
    return launch_s2_x3<5, 2, false>(a, (nt + 1) / 2, st);   // weights packed in groups of two N-tiles (zero-padded)
}

extern "C" int EFFI_FN(effi_conv3d_k3s2_bf16x3_f32)(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int D,
                                           int h, int w, int relu, float* out, effi_stream_t stream) {
    if (!in || !wpack_bf16 || !bias || !out || cin < 1 || cout < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if ((w & 3) || cout > 64) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = in;
        a.ch[i] = (i == 0) ? cin : 0;
    }
    a.cin = cin;                                 // channels per input plane
    a.kgroups = 0;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.hin = h;
    a.win = w;
    a.h = (h - 1) / 2 + 1;
    a.w = (w - 1) / 2 + 1;
    a.zin = D;
    a.zcount = (D - 1) / 2 + 1;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = (long)D * h * w;
    a.ostride = (long)a.zcount * a.h * a.w;
    hipStream_t st = effi_s(stream);
    const int nt = (cout + 15) / 16;
    if (nt == 1) return launch_s2_x3<3, 1, true>(a, 1, st);
    return launch_s2_x3<3, 2, true>(a, (nt + 1) / 2, st);
}


static int fill_bf16x3_plain(Conv2dArgs& a, const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                             const float* bias, int cout, int h, int w, int act, float* out0) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack_bf16 || !bias || !out0) return EFFI_ERR_BADARG;
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : srcs[0];
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        if (i + 1 < n_src && (src_channels[i] & 7)) return EFFI_ERR_UNSUPPORTED;
        a.cin += a.ch[i];
    }
    a.kgroups = (a.cin + 3) / 4;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = act;
    a.hd = cout / 2;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out0;
    a.out1 = nullptr;
    a.cstride = a.ostride = (long)h * w;
    a.zcount = a.zin = 0;
    return EFFI_OK;
}

extern "C" int EFFI_FN(effi_conv2d_k3_bf16x3_pair_f32)(const float* const* srcs_a, const int* src_channels_a, int n_src_a,
                                              const void* wpack_a, const float* bias_a, float* out_a,
                                              const float* const* srcs_b, const int* src_channels_b, int n_src_b,
                                              const void* wpack_b, const float* bias_b, float* out_b, int cout, int h, int w,
                                              int act, effi_stream_t stream) {
    if (cout < 1 || h < 1 || w < 1 || act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
    if (w & 3) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a0, a1;
    int rc = fill_bf16x3_plain(a0, srcs_a, src_channels_a, n_src_a, wpack_a, bias_a, cout, h, w, act, out_a);
    if (rc != EFFI_OK) return rc;
    rc = fill_bf16x3_plain(a1, srcs_b, src_channels_b, n_src_b, wpack_b, bias_b, cout, h, w, act, out_b);
    if (rc != EFFI_OK) return rc;
    hipStream_t st = effi_s(stream);
    switch ((cout + 15) / 16) {
        case 1: return launch_bf16x3_pair<1>(a0, a1, st);
        case 2: return launch_bf16x3_pair<2>(a0, a1, st);
        case 3: return launch_bf16x3_pair<3>(a0, a1, st);
        case 4: return launch_bf16x3_pair<4>(a0, a1, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

extern "C" int EFFI_FN(effi_conv2d_k3_bf16x3_f32)(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                         const float* bias, int cout, int h, int w, int epilogue, int act, const float* aux0,
                                         const float* aux1, const float* disp_range, int n_range, float* out0, float* out1,
                                         effi_stream_t stream) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack_bf16 || !bias || !out0) return EFFI_ERR_BADARG;
    if (cout < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (w & 3) return EFFI_ERR_UNSUPPORTED;                 // rows must be float4-aligned (callers fall back to the fp32 kernel)
    Conv2dArgs a;
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : srcs[0];
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        if (i + 1 < n_src && (src_channels[i] & 7)) return EFFI_ERR_UNSUPPORTED;   // an octet of channels lies in one source
        a.cin += a.ch[i];
    }
    a.kgroups = (a.cin + 3) / 4;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = act;
    a.hd = cout / 2;
    a.aux0 = aux0;
    a.aux1 = aux1;
    a.disp_range = disp_range;
    a.n_range = n_range;
    a.out0 = out0;
    a.out1 = out1;
    a.cstride = a.ostride = (long)h * w;
    a.zcount = 0;
    const int nt = (cout + 15) / 16;
    hipStream_t st = effi_s(stream);
    switch (epilogue) {
        case EFFI_EPI_PLAIN:
            if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
            return dispatch_bf16x3<EFFI_EPI_PLAIN>(a, nt, st);
        case EFFI_EPI_NHWC:
            if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
            return dispatch_bf16x3<EFFI_EPI_NHWC>(a, nt, st);
        case EFFI_EPI_ADD_SHUF2:
        case EFFI_EPI_NHWC_ADD_SHUF2:
            if (!aux0 || (h & 1) || (w & 1) || act != EFFI_ACT_NONE) return EFFI_ERR_BADARG;
            if (nt == 1) return epilogue == EFFI_EPI_ADD_SHUF2 ? launch_bf16x3<1, EFFI_EPI_ADD_SHUF2>(a, st)
                                                               : launch_bf16x3<1, EFFI_EPI_NHWC_ADD_SHUF2>(a, st);
            if (nt == 2) return epilogue == EFFI_EPI_ADD_SHUF2 ? launch_bf16x3<2, EFFI_EPI_ADD_SHUF2>(a, st)
                                                               : launch_bf16x3<2, EFFI_EPI_NHWC_ADD_SHUF2>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        case EFFI_EPI_GRU_ZR:
            if (!aux0 || !out1 || (cout % 32) != 0) return EFFI_ERR_BADARG;
            if (nt == 2) return launch_bf16x3<2, EFFI_EPI_GRU_ZR>(a, st);
            if (nt == 4) return launch_bf16x3<4, EFFI_EPI_GRU_ZR>(a, st);
            if (nt == 6) return launch_bf16x3<6, EFFI_EPI_GRU_ZR>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        case EFFI_EPI_GRU_Q:
            if (!aux0 || !aux1 || (cout % 16) != 0) return EFFI_ERR_BADARG;
            if (nt == 1) return launch_bf16x3<1, EFFI_EPI_GRU_Q>(a, st);
            if (nt == 2) return launch_bf16x3<2, EFFI_EPI_GRU_Q>(a, st);
            if (nt == 3) return launch_bf16x3<3, EFFI_EPI_GRU_Q>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        default:
            return EFFI_ERR_UNSUPPORTED;
    }
}

extern "C" int EFFI_FN(effi_conv2d_k3_k1_bf16x3_f32)(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                            const float* bias, int cout1, int relu1, const float* extra, int c_extra,
                                            const void* w2pack_bf16, const float* bias2, int cout2, int relu, int h, int w,
                                            float* out, effi_stream_t stream) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack_bf16 || !bias || !w2pack_bf16 || !bias2 || !out)
        return EFFI_ERR_BADARG;
    if (cout1 < 1 || cout2 < 1 || h < 1 || w < 1 || c_extra < 0 || (c_extra > 0 && !extra)) return EFFI_ERR_BADARG;
    if ((w & 3) || cout1 > 96 || c_extra > 16 || cout2 > 96) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : srcs[0];
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        if (i + 1 < n_src && (src_channels[i] & 7)) return EFFI_ERR_UNSUPPORTED;
        a.cin += a.ch[i];
    }
    a.kgroups = relu1 ? 1 : 0;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout1;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = c_extra;
    a.aux0 = c_extra ? extra : bias2;
    a.aux1 = reinterpret_cast<const float*>(w2pack_bf16);
    a.disp_range = bias2;
    a.n_range = cout2;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = a.ostride = (long)h * w;
    a.zcount = a.zin = 0;
    hipStream_t st = effi_s(stream);
    switch ((cout1 + 15) / 16) {
        case 1: return launch_bf16x3<1, EFFI_EPI_K1>(a, st);
        case 2: return launch_bf16x3<2, EFFI_EPI_K1>(a, st);
        case 3: return launch_bf16x3<3, EFFI_EPI_K1>(a, st);
        case 4: return launch_bf16x3<4, EFFI_EPI_K1>(a, st);
        case 6: return launch_bf16x3<6, EFFI_EPI_K1>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

// ------------------------------------------------------------------------------------------------
// The pyramid's first block at full resolution (models/module.py:353-356, conv0 = Conv2d(3, 8) -> Conv2d(8, 8), BN folded, ReLU
// after each) as ONE kernel: both layers have at most 8 input channels, i.e. ONE octet, so a layer is 9 K-items (taps) = 3 K-steps
// instead of the 5 of a 16-channel chunk, and the 8-channel intermediate (61 MB at 1184x1600, written and re-read by the two-launch
// form) stays in LDS.  Same construction as encoder_tail_bf16x3_kernel (12 x 16 output tile, first layer on the tile grown by
// one pixel = 16 linearised pixel groups, its result split to hi/lo in the A-tile layout), but these layers are memory-bound and
// the LDS footprint is 33 KB, so the halo recompute is cheap and the occupancy stays high.
// Weights: [3 K-steps][hi|lo][64 lanes][8] bf16 per layer (packing.pack_conv2d_bf16x3_oct), lane = q*16 + j holds
// W[j][e][tap = 4 s + q] (zero for taps 9..11, channels >= cin, couts >= cout).
// ------------------------------------------------------------------------------------------------
namespace {
struct ConvTwiceArgs {
    const float* src;        // [cin][h][w], cin <= 8
    int cin;
    const float* w_a;        // first layer fragments
    const float* bias_a;     // padded to 16
    const float* w_b;        // second layer
    const float* bias_b;     // padded to 16
    int cout;                // second layer's output channels (<= 16); the first layer's are 8
    int h, w;
    float* out;              // [cout][h][w]
    const float* zeros;
};

__global__ __launch_bounds__(256) void conv2d_k3_twice_oct_kernel(const ConvTwiceArgs a, int tiles_x, int ntiles) {
    // ROW-PAIR operands (both layers have at most 8 output channels): MFMA rows 0-7 = the channels of an image row, rows 8-15 = the
    // same channels of the row below, K = (dy in 0..3, dx) over the 4 x 3 window both rows see = 12 taps = exactly 3 K-steps for TWO
    // rows (packing.pack_conv2d_bf16x3_oct: W[j][e][dy][dx] for rows j < 8, zero at dy = 3; W[j-8][e][dy-1][dx] for j >= 8, zero at dy = 0)
    constexpr int TR = 16, TW = 16, RW = TW + 2, RH = TR + 2, NPM = RW * RH;          // 324 mid pixels = 9 row pairs x 18 columns
    constexpr int NPP = (RH / 2) * RW;                                               // 162 pair-pixels of the first layer
    constexpr int NGA = 3;                                                           // 12 groups of 16 pair-pixels, 3 per wave
    constexpr int AR = TR + 4, AW = TW + 8, AQ = AW / 4, APIX = AR * AW, NKS = 3;
    constexpr int NITEMS = APIX / 4;                                                 // 120 staging items (one octet)
    constexpr int NBF = NKS * 2 * 64;                                                // 16-byte units of B per layer
    static_assert(4 * NGA * 16 >= NPP, "groups cover the region");
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[APIX * 8];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[APIX * 8];
    __shared__ __attribute__((aligned(16))) unsigned short mid_h[NPM * 8];
    __shared__ __attribute__((aligned(16))) unsigned short mid_l[NPM * 8];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[2 * 512 * 8];      // both layers' fragments (2 x 384 units, padded)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w;
    const long hw = (long)h * w;
    // persistent workgroups: a contiguous run of tiles each (neighbouring tiles share their halos in the same XCD's L2); the next
    // tile's input is requested before the current tile is multiplied, the weights are fetched once
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t0 = effi_xcd_remap(blockIdx.x, gridDim.x) * per, t1 = min(t0 + per, ntiles);
    if (t0 >= t1) return;

    {
        const unsigned short* wa = reinterpret_cast<const unsigned short*>(a.w_a);
        const unsigned short* wb = reinterpret_cast<const unsigned short*>(a.w_b);
        for (int u = tid; u < NBF; u += 256) {
            *reinterpret_cast<f32x4*>(&lds_b[u * 8]) = *reinterpret_cast<const f32x4*>(wa + (long)u * 8);
            *reinterpret_cast<f32x4*>(&lds_b[(512 + u) * 8]) = *reinterpret_cast<const f32x4*>(wb + (long)u * 8);
        }
    }
    // input tile 20 x 24 with origin (y0 - 2, x0 - 4): threads 0..119 own a pixel quad each
    const bool stager = tid < NITEMS;
    const int srow = stager ? tid / AQ : 0, sqx = stager ? tid - srow * AQ : 0;
    const int s_lds = (srow * AW + 4 * sqx) * 8;
    const int emax = a.cin - 1;
    f32x4 pa[8];
    auto prefetch = [&](int tile) {
        const int ty_ = tile / tiles_x;
        const int x0 = (tile - ty_ * tiles_x) * TW, y0 = ty_ * TR;
        const int sgy = y0 - 2 + srow, sgx = x0 - 4 + 4 * sqx;
        const bool s_in = (sgy >= 0) & (sgy < h) & (sgx >= 0) & (sgx < w);
        const float* q = s_in ? a.src + ((long)sgy * w + sgx) : a.zeros;
        const long step = s_in ? hw : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pa[e] = *reinterpret_cast<const f32x4*>(q);
            q += (e < emax) ? step : 0;                    // channels past the last real one re-read it (their weights are zero)
        }
    };

    // K-step s: lane quarter lk owns tap 4 s + lk = (dy, dx) = ((4 s + lk) / 3, (4 s + lk) % 3) of the 4 x 3 window
    int koffa[NKS], koffb[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int tap = 4 * s_ + lk;
        koffa[s_] = ((tap / 3) * AW + tap % 3) * 8;
        koffb[s_] = ((4 * wv + tap / 3) * RW + li + tap % 3) * 8;
    }
    // first layer: group g = 3 wv + m, pair-pixel pp = 16 g + li = (pair row pp / 18, column pp % 18) of the 18 x 18 region
    int gbase[NGA], pmid[NGA], prow[NGA], pcol[NGA];
#pragma unroll
    for (int m = 0; m < NGA; ++m) {
        const int pp = min(16 * (NGA * wv + m) + li, NPP - 1);      // lanes past the region repeat its last pair (same values, same slots)
        const int pr2 = pp / RW;
        pcol[m] = pp - pr2 * RW;
        prow[m] = 2 * pr2 + (lk >> 1);                              // the region row this lane's accumulator half belongs to
        pmid[m] = prow[m] * RW + pcol[m];
        gbase[m] = (2 * pr2 * AW + pcol[m] + 2) * 8;
    }
    float ba[4], bb[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ba[r] = a.bias_a[4 * (lk & 1) + r];
        bb[r] = a.bias_b[4 * (lk & 1) + r];
    }

    if (stager) prefetch(t0);
#pragma unroll 1
    for (int tile = t0; tile < t1; ++tile) {
        const int ty_ = tile / tiles_x;
        const int x0 = (tile - ty_ * tiles_x) * TW, y0 = ty_ * TR;
        if (stager) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                bf16x8 hi, lo;
                split_octet(pa, px, hi, lo);
                *reinterpret_cast<bf16x8*>(&lds_ah[s_lds + px * 8]) = hi;
                if (!kHiOnly) *reinterpret_cast<bf16x8*>(&lds_al[s_lds + px * 8]) = lo;
            }
            if (tile + 1 < t1) prefetch(tile + 1);
        }
        __syncthreads();                                   // input tile complete (and the previous tile's second layer done with mid)

        // ---- first layer on the 18 x 18 region (origin (y0 - 1, x0 - 1)): 12 groups of 16 pair-pixels, 3 per wave ----
        {
            f32x4 acc[NGA];
#pragma unroll
            for (int m = 0; m < NGA; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int s_ = 0; s_ < NKS; ++s_) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[((s_ * 2 + 0) * 64 + lane) * 8]);
                bf16x8 bl = bh;
                if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[((s_ * 2 + 1) * 64 + lane) * 8]);
#pragma unroll
                for (int m = 0; m < NGA; ++m) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&lds_ah[gbase[m] + koffa[s_]]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[m], 0, 0, 0);
                    if (!kHiOnly) {
                        const bf16x8 al = *reinterpret_cast<const bf16x8*>(&lds_al[gbase[m] + koffa[s_]]);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[m], 0, 0, 0);
                    }
                }
            }
            // lane (li, lk): region row prow (upper / lower of the pair by lk >> 1), channels 4 (lk & 1) .. + 3
#pragma unroll
            for (int m = 0; m < NGA; ++m) {
                const int gy = y0 - 1 + prow[m], gx = x0 - 1 + pcol[m];
                const bool pin = (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w);
                f32x4 vf;
#pragma unroll
                for (int r = 0; r < 4; ++r) vf[r] = pin ? fmaxf(acc[m][r] + ba[r], 0.0f) : 0.0f;
                const bf16x4 vh = __builtin_convertvector(vf, bf16x4);
                const bf16x4 vl = __builtin_convertvector(vf - __builtin_convertvector(vh, f32x4), bf16x4);
                *reinterpret_cast<bf16x4*>(&mid_h[pmid[m] * 8 + 4 * (lk & 1)]) = vh;
                if (!kHiOnly) *reinterpret_cast<bf16x4*>(&mid_l[pmid[m] * 8 + 4 * (lk & 1)]) = vl;
            }
        }
        __syncthreads();                                   // mid complete; the input tile may be overwritten

        // ---- second layer on the 16 x 16 tile: wave wv owns rows 4 wv .. 4 wv + 3 as two row pairs ----
        f32x4 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s_ = 0; s_ < NKS; ++s_) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&lds_b[(512 + (s_ * 2 + 0) * 64 + lane) * 8]);
            bf16x8 bl = bh;
            if (!kHiOnly) bl = *reinterpret_cast<const bf16x8*>(&lds_b[(512 + (s_ * 2 + 1) * 64 + lane) * 8]);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(&mid_h[koffb[s_] + m * 2 * RW * 8]);
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[m], 0, 0, 0);
                if (!kHiOnly) {
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(&mid_l[koffb[s_] + m * 2 * RW * 8]);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, acc[m], 0, 0, 0);
                }
            }
        }
        const int x = x0 + li, co0 = 4 * (lk & 1);
        if (x < w && co0 < a.cout) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int y = y0 + 4 * wv + 2 * m + (lk >> 1);
                if (y >= h) continue;
                const long pix = (long)y * w + x;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < a.cout) a.out[(long)(co0 + r) * hw + pix] = fmaxf(acc[m][r] + bb[r], 0.0f);
            }
        }
    }
}

}  // namespace

extern "C" int EFFI_FN(effi_conv2d_k3_twice_bf16x3_f32)(const float* in, int cin, const void* w1_bf16, const float* bias1,
                                               const void* w2_bf16, const float* bias2, int cout, int h, int w, float* out,
                                               effi_stream_t stream) {
    if (!in || !w1_bf16 || !bias1 || !w2_bf16 || !bias2 || !out || cin < 1 || cout < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if ((w & 3) || cin > 8 || cout > 8) return EFFI_ERR_UNSUPPORTED;
    ConvTwiceArgs a;
    a.src = in; a.cin = cin;
    a.w_a = reinterpret_cast<const float*>(w1_bf16); a.bias_a = bias1;
    a.w_b = reinterpret_cast<const float*>(w2_bf16); a.bias_b = bias2;
    a.cout = cout; a.h = h; a.w = w; a.out = out;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    const int tiles_x = effi_cdiv(w, 16), ntiles = tiles_x * effi_cdiv(h, 16);
    const int nwg = ntiles < 1024 ? ntiles : 1024;         // 4 persistent workgroups per CU (38 KB of LDS each)
    hipLaunchKernelGGL(conv2d_k3_twice_oct_kernel, dim3(nwg), dim3(256), 0, effi_s(stream), a, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

extern "C" int EFFI_FN(effi_encoder_tail_bf16x3_f32)(const float* cor1, const float* dfm1, int hd, const void* wc2_bf16, const float* bias_c2,
                                            const void* wd2_bf16, const float* bias_d2, const void* wd_bf16, const float* bias_d,
                                            int cmix, const float* extra, int c_extra, const void* w2pack_bf16, const float* bias2,
                                            int cout2, int h, int w, float* out, effi_stream_t stream) {
    if (!cor1 || !dfm1 || !wc2_bf16 || !bias_c2 || !wd2_bf16 || !bias_d2 || !wd_bf16 || !bias_d || !w2pack_bf16 || !bias2 || !out)
        return EFFI_ERR_BADARG;
    if (hd < 1 || cmix < 1 || cout2 < 1 || h < 1 || w < 1 || c_extra < 0 || (c_extra > 0 && !extra)) return EFFI_ERR_BADARG;
    if ((w & 3) || hd != 16 || cmix > 16 || c_extra > 16 || cout2 > 96) return EFFI_ERR_UNSUPPORTED;
    EncTailArgs a;
    a.src_a = cor1; a.src_b = dfm1;
    a.w_a = reinterpret_cast<const float*>(wc2_bf16); a.bias_a = bias_c2;
    a.w_b = reinterpret_cast<const float*>(wd2_bf16); a.bias_b = bias_d2;
    a.w_d = reinterpret_cast<const float*>(wd_bf16); a.bias_d = bias_d;
    a.extra = c_extra ? extra : bias2; a.c_extra = c_extra;
    a.w2 = reinterpret_cast<const float*>(w2pack_bf16); a.bias2 = bias2; a.cout2 = cout2;
    a.h = h; a.w = w; a.out = out;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    const int tiles_x = effi_cdiv(w, 16), ntiles = tiles_x * effi_cdiv(h, 12);
    hipLaunchKernelGGL((encoder_tail_bf16x3_kernel<1, 1>), dim3(ntiles), dim3(256), 0, effi_s(stream), a, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

extern "C" int EFFI_FN(effi_conv2d_k3_k1_up2x_bf16x3_f32)(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                                 const float* bias, int cout1, const void* w2pack_bf16, const float* bias2,
                                                 const float* inv_depth, const float* disp_range, int n_range, int h, int w,
                                                 float* out_depth, float* out_depth_inv, effi_stream_t stream) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack_bf16 || !bias || !w2pack_bf16 || !bias2 || !inv_depth ||
        !disp_range || n_range < 2 || !out_depth)
        return EFFI_ERR_BADARG;
    if (cout1 < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if ((w & 3) || cout1 > 96) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : srcs[0];
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        if (i + 1 < n_src && (src_channels[i] & 7)) return EFFI_ERR_UNSUPPORTED;
        a.cin += a.ch[i];
    }
    a.kgroups = 1;                          // ReLU between the 3x3 and the 1x1 convolution (models/update.py:110)
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout1;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = EFFI_ACT_NONE;
    a.hd = 0;
    a.aux0 = inv_depth;
    a.aux1 = reinterpret_cast<const float*>(w2pack_bf16);
    a.disp_range = bias2;
    a.n_range = 36;
    a.out0 = out_depth;
    a.out1 = out_depth_inv;
    a.cstride = a.ostride = (long)h * w;
    a.zcount = 0;
    a.zin = n_range;
    a.xptr0 = disp_range;
    hipStream_t st = effi_s(stream);
    switch ((cout1 + 15) / 16) {
        case 2: return launch_bf16x3<2, EFFI_EPI_K1UP>(a, st);
        case 4: return launch_bf16x3<4, EFFI_EPI_K1UP>(a, st);
        case 6: return launch_bf16x3<6, EFFI_EPI_K1UP>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

extern "C" int EFFI_FN(effi_conv3d_k3s1_bf16x3_f32)(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                           const float* bias, int cout, int D, int h, int w, int relu, float* out,
                                           effi_stream_t stream) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack_bf16 || !bias || !out) return EFFI_ERR_BADARG;
    if (cout < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    Conv2dArgs a;                                          // any w: quads that straddle a row end are staged element by element
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : srcs[0];
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        if (i + 1 < n_src && (src_channels[i] & 7)) return EFFI_ERR_UNSUPPORTED;
        a.cin += a.ch[i];
    }
    if (a.cin & 7) return EFFI_ERR_UNSUPPORTED;            // an octet of (plane, channel) lies in one plane
    a.kgroups = 0;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = a.ostride = (long)D * h * w;
    a.zcount = D;
    hipStream_t st = effi_s(stream);
    switch ((cout + 15) / 16) {
        case 1: return launch_bf16x3<1, EFFI_EPI_PLAIN, true>(a, st);
        case 2: return launch_bf16x3<2, EFFI_EPI_PLAIN, true>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

template <int NOCT, int NT>
static int launch_roll(const Conv2dArgs& a, hipStream_t st, const Conv2dArgs* pair = nullptr) {
    // Rows per wave (MR) and planes per workgroup (ZT) from a small cost model fitted to sweeps at the cfg3 shapes
    // (tools/sweep_roll.sh): the chip holds 256 * occ workgroups at a time (occ from the LDS image / register count of the
    // instantiation), a launch takes ceil(workgroups / that) rounds, and a workgroup costs (ZT + 3) plane steps (3 ~ filling
    // the window), 1.3x as much with 4 rows per wave as with 2.  E.g. 8->8 at 48x148x200: MR 4 / ZT 6 (1040 workgroups, 3
    // rounds) 74 us, MR 2 / ZT 16 (741 workgroups, 1 round) 55 us.
    const int cols = effi_cdiv(a.w, 16), D = a.zcount;
    const bool rp = NT == 1 && a.cout <= 8 && effi_option(EFFI_OPT_ROLL_RP) != 0;      // row-pair operand (see conv3d_roll_rp_bf16x3_body)
    int mr = 2, zt = D;
    double best = 1e30;
    for (int m = 2; m <= 4; m += 2) {
        // (row-pair kernels, weights in registers: 3 workgroups per CU with 8 input channels, 2 with 16, whatever MR)
        const int occ = rp ? (NOCT == 1 ? 3 : 2) : (NOCT == 1 && NT == 1) ? (m == 2 ? 3 : 2) : (NOCT == 1 ? 2 : (NT == 1 && m == 2 ? 2 : 1));
        const long tiles_m = (long)cols * effi_cdiv(a.h, 4 * m), slots = 256L * occ;
        for (int nz = 1; nz <= D; ++nz) {
            const int z = effi_cdiv(D, nz);
            const long wgs = tiles_m * effi_cdiv(D, z) * (pair ? 2 : 1);
            const double cost = (double)effi_cdiv(wgs, slots) * (z + 3) * (m == 4 ? 1.3 : 1.0);
            if (cost < best - 1e-9) { best = cost; mr = m; zt = z; }
        }
    }
    const long fm = effi_option(EFFI_OPT_ROLL_MR), fz = effi_option(EFFI_OPT_ROLL_ZT);       // A/B overrides of the rounds model
    if (fm != EFFI_OPT_UNSET) mr = (int)fm;
    if (fz != EFFI_OPT_UNSET) zt = (int)fz;
    const long tiles = (long)cols * effi_cdiv(a.h, 4 * mr);
    const dim3 grid((unsigned)tiles, (unsigned)effi_cdiv(D, zt), pair ? 2 : 1);
    // (option roll_rp = 0, with packing.py: the one-row-per-tile operand, A/B runs)
    if (rp) {
        if (pair) {
            if (mr == 4) hipLaunchKernelGGL((conv3d_roll_rp_bf16x3_pair_kernel<NOCT, 4>), grid, dim3(256), 0, st, a, *pair, cols, (int)tiles, zt);
            else hipLaunchKernelGGL((conv3d_roll_rp_bf16x3_pair_kernel<NOCT, 2>), grid, dim3(256), 0, st, a, *pair, cols, (int)tiles, zt);
        } else {
            if (mr == 4) hipLaunchKernelGGL((conv3d_roll_rp_bf16x3_kernel<NOCT, 4>), grid, dim3(256), 0, st, a, cols, (int)tiles, zt);
            else hipLaunchKernelGGL((conv3d_roll_rp_bf16x3_kernel<NOCT, 2>), grid, dim3(256), 0, st, a, cols, (int)tiles, zt);
        }
        return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
    }
    if (pair) {
        if (mr == 4) hipLaunchKernelGGL((conv3d_roll_bf16x3_pair_kernel<NOCT, NT, 4>), grid, dim3(256), 0, st, a, *pair, cols, (int)tiles, zt);
        else hipLaunchKernelGGL((conv3d_roll_bf16x3_pair_kernel<NOCT, NT, 2>), grid, dim3(256), 0, st, a, *pair, cols, (int)tiles, zt);
    } else {
        if (mr == 4) hipLaunchKernelGGL((conv3d_roll_bf16x3_kernel<NOCT, NT, 4>), grid, dim3(256), 0, st, a, cols, (int)tiles, zt);
        else hipLaunchKernelGGL((conv3d_roll_bf16x3_kernel<NOCT, NT, 2>), grid, dim3(256), 0, st, a, cols, (int)tiles, zt);
    }
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

static int fill_roll_args(Conv2dArgs& a, const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                          const float* bias, int cout, int D, int h, int w, int relu, float* out) {
    if (!srcs || !src_channels || n_src < 1 || n_src > 2 || !wpack_bf16 || !bias || !out) return EFFI_ERR_BADARG;
    if (cout < 1 || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = (i < n_src) ? srcs[i] : nullptr;
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        a.cin += a.ch[i];
    }
    if ((w & 3) || (a.cin != 8 && a.cin != 16) || (a.ch[0] & 7) || cout > 32) return EFFI_ERR_UNSUPPORTED;
    if (n_src == 1) a.src[1] = a.src[0];
    a.kgroups = 0;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = 0;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = out;
    a.out1 = nullptr;
    a.cstride = a.ostride = (long)D * h * w;
    a.zcount = a.zin = D;
    return EFFI_OK;
}

extern "C" int EFFI_FN(effi_conv3d_k3s1_roll_bf16x3_f32)(const float* const* srcs, const int* src_channels, int n_src,
                                                const void* wpack_bf16, const float* bias, int cout, int D, int h, int w,
                                                int relu, float* out, effi_stream_t stream) {
    Conv2dArgs a;
    const int rc = fill_roll_args(a, srcs, src_channels, n_src, wpack_bf16, bias, cout, D, h, w, relu, out);
    if (rc != EFFI_OK) return rc;
    hipStream_t st = effi_s(stream);
    const int nt = (cout + 15) / 16;
    if (a.cin == 8) return nt == 1 ? launch_roll<1, 1>(a, st) : launch_roll<1, 2>(a, st);
    return nt == 1 ? launch_roll<2, 1>(a, st) : launch_roll<2, 2>(a, st);
}

extern "C" int EFFI_FN(effi_conv3d_k3s1_roll_bf16x3_pair_f32)(const float* const* srcs_a, const void* wpack_a, const float* bias_a,
                                                     float* out_a, const float* const* srcs_b, const void* wpack_b,
                                                     const float* bias_b, float* out_b, const int* src_channels, int n_src,
                                                     int cout, int D, int h, int w, int relu, effi_stream_t stream) {
    Conv2dArgs a, b;
    int rc = fill_roll_args(a, srcs_a, src_channels, n_src, wpack_a, bias_a, cout, D, h, w, relu, out_a);
    if (rc != EFFI_OK) return rc;
    rc = fill_roll_args(b, srcs_b, src_channels, n_src, wpack_b, bias_b, cout, D, h, w, relu, out_b);
    if (rc != EFFI_OK) return rc;
    if (cout > 16) return EFFI_ERR_UNSUPPORTED;
    hipStream_t st = effi_s(stream);
    return a.cin == 8 ? launch_roll<1, 1>(a, st, &b) : launch_roll<2, 1>(a, st, &b);
}

// conv0 | conv_cost -> conv1 of two cross-scale blocks over one fine volume x [D][H][W] (csp_gen_roll_rp_kernel): priors [D][h][w],
// h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1, outputs [8][D][h][w].  Bitwise the result of effi_conv3d_k3_pair_f32 (sxy 2 / sxy 1) +
// effi_conv3d_k3s1_roll_bf16x3_pair_f32 on the same weights.
extern "C" int EFFI_FN(effi_csp_gen_roll_bf16x3_pair_f32)(const float* x, int D, int H, int W, const float* prior_a, const float* w0_a,
                                                          const float* b0_a, const float* wc_a, const float* bc_a, const void* w1_a,
                                                          const float* b1_a, float* out_a, const float* prior_b, const float* w0_b,
                                                          const float* b0_b, const float* wc_b, const float* bc_b, const void* w1_b,
                                                          const float* b1_b, float* out_b, effi_stream_t stream) {
    if (!x || !prior_a || !w0_a || !b0_a || !wc_a || !bc_a || !w1_a || !b1_a || !out_a) return EFFI_ERR_BADARG;
    if (!prior_b || !w0_b || !b0_b || !wc_b || !bc_b || !w1_b || !b1_b || !out_b) return EFFI_ERR_BADARG;
    if (D < 1 || H < 1 || W < 1) return EFFI_ERR_BADARG;
    if (D > kCspMaxD || (long)D * H * W >= (1L << 30)) return EFFI_ERR_UNSUPPORTED;
    const int h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1;
    const CspGenCall ca{prior_a, w0_a, b0_a, wc_a, bc_a, reinterpret_cast<const float*>(w1_a), b1_a, out_a};
    const CspGenCall cb{prior_b, w0_b, b0_b, wc_b, bc_b, reinterpret_cast<const float*>(w1_b), b1_b, out_b};
    const int cols = effi_cdiv(w, 16), tiles = cols * effi_cdiv(h, 8);
    hipLaunchKernelGGL(csp_gen_roll_rp_kernel, dim3((unsigned)tiles, 1, 2), dim3(256), 0, effi_s(stream), x, H, W, ca, cb, D, h, w, cols, tiles);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int EFFI_FN(effi_deconv3d_k3s2_bf16x3_f32)(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int D,
                                             int h, int w, int relu, const float* skip, float* out, effi_stream_t stream) {
    if (!in || !wpack_bf16 || !bias || !out || D < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (cin < 16 || (cin & 15) || cout < 1 || cout > 16) return EFFI_ERR_UNSUPPORTED;
    DeconvArgs a;
    a.in = in;
    a.wpack = wpack_bf16;
    a.bias = bias;
    a.skip = skip;
    a.zeros = effi_zero_page();
    if (!a.zeros) return EFFI_ERR_WORKSPACE;
    a.out = out;
    a.cin = cin;
    a.cout = cout;
    a.D = D;
    a.h = h;
    a.w = w;
    a.relu = relu;
    const int tiles_x = effi_cdiv(w, 16);
    const bool al = (w & 3) == 0, half = cout <= 8;       // cout <= 8: packing with 9 fragments per chunk (see the kernel)
    hipStream_t st = effi_s(stream);
    // rows per wave from the same rounds model as launch_roll: 256 * occ workgroups at a time (occ 5 / 3 with cout <= 8, 3 / 2
    // otherwise, for 1 / 2 rows per wave), a workgroup costs (rows per wave + 2); measured 16->8 into 48x148x200: 57 -> 43 us
    const int occ1 = half ? 5 : 3, occ2 = half ? 3 : 2;
    const long wg1 = (long)tiles_x * effi_cdiv(h, 4) * D, wg2 = (long)tiles_x * effi_cdiv(h, 8) * D;
    bool mr2 = effi_cdiv(wg2, 256L * occ2) * 4 < effi_cdiv(wg1, 256L * occ1) * 3;
    const long fdm = effi_option(EFFI_OPT_DECONV_MR);
    if (fdm != EFFI_OPT_UNSET) mr2 = fdm == 2;
    const dim3 grid(tiles_x * effi_cdiv(h, mr2 ? 8 : 4), D);
#define EFFI_DC(MRV, ALV, HFV) \
    hipLaunchKernelGGL((deconv3d_s2_bf16x3_kernel<MRV, ALV, HFV>), grid, dim3(256), 0, st, a, tiles_x, (int)grid.x)
    if (mr2) {
        if (al) { if (half) EFFI_DC(2, true, true); else EFFI_DC(2, true, false); }
        else { if (half) EFFI_DC(2, false, true); else EFFI_DC(2, false, false); }
    } else {
        if (al) { if (half) EFFI_DC(1, true, true); else EFFI_DC(1, true, false); }
        else { if (half) EFFI_DC(1, false, true); else EFFI_DC(1, false, false); }
    }
#undef EFFI_DC
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

#ifndef EFFI_BF16_ONLY
extern "C" int effi_conv2d_c1k7_relu_f32(const float* in, const float* weight, const float* bias, int cout, int h,
                                         int w, float* out, effi_stream_t stream) {
    if (!in || !weight || !bias || !out || h < 1 || w < 1) return EFFI_ERR_BADARG;
    dim3 grid(effi_cdiv(w, EFFI_C1K7_TX), effi_cdiv(h, EFFI_C1K7_TY), cout / 16);
    hipStream_t st = effi_s(stream);
    switch (cout) {
        case 16: hipLaunchKernelGGL(conv2d_c1k7_relu_kernel<16>, grid, dim3(256), 0, st, in, weight, bias, h, w, out); break;
        case 32: hipLaunchKernelGGL(conv2d_c1k7_relu_kernel<32>, grid, dim3(256), 0, st, in, weight, bias, h, w, out); break;
        case 48: hipLaunchKernelGGL(conv2d_c1k7_relu_kernel<48>, grid, dim3(256), 0, st, in, weight, bias, h, w, out); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
#endif
