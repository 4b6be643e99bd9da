// K1+K2+K3: homography warp fused with the correlation cost (the warped C x D x h x w volume of the
// reference, models/module.py:340, is never written).
//
// Data layout: features are channel-last ([h][w][C]) so that one bilinear tap of one pixel is ONE
// contiguous C*4-byte read.  C/4 lanes co-operate on a pixel (8 lanes for C=32): each lane loads one
// float4 (4 channels) per tap, so a wave-instruction fetches whole 128/64/32-byte segments, and the
// per-pixel channel sum is a 1-3 step DPP butterfly inside the lane group (no LDS, no atomics).
// The gather is served by L1/L2; blocks are remapped so that one XCD sweeps a contiguous band of the
// image and neighbouring tiles share that XCD's L2.
//
// Arithmetic follows the reference op for op (models/module.py:318-341 and ATen's grid_sample with
// bilinear / zeros / align_corners=True); built with -ffp-contract=off, FMAs are explicit.  One documented exception: the
// default kernels of the path (stage-1 window kernel, stage-2/3 hypothesis-per-lane kernel) evaluate the four divisions of the
// projection with a refined reciprocal + one residual step (project_xy below: correctly rounded except for rare 1-ulp cases of
// a continuous function); the exact-division forms stay selectable (EFFI_WARP_LDS_KB=-1, EFFI_DYN_SETUP=exact).
#include "common.hpp"
#include <cstdlib>

namespace {

template <int C> struct WarpGeom {
    static constexpr int LPP = C / 4;                 // lanes per pixel
    static constexpr int PIX = 256 / LPP;             // pixels per 256-thread block
    static constexpr int TW = (C == 32) ? 8 : 16;     // tile width  (pixels)
    static constexpr int TH = PIX / TW;               // tile height (4, 4, 8)
};

__device__ __forceinline__ const float* pick_view_list(const EffiPtrList& l, int v) {
    const float* p = l.p[0];
#pragma unroll
    for (int i = 1; i <= EFFI_MAX_VIEWS; ++i)
        if (v == i) p = l.p[i];
    return p;
}
__device__ __forceinline__ const float* pick_view(const EffiPtrList& l, int v) {
    if (l.tbl) return l.tbl[1 + v];                   // view-table form (uniform branch, scalar load)
    return pick_view_list(l, v);
}

struct Taps {
    float w[4];
    int off[4];   // element offsets of the 4 taps' pixel (already multiplied by C)
};

// NO KERNEL OF THE SHIPPED LIBRARY PASSES A TAP SET-UP FROM ONE LANE TO ANOTHER.  Rounds 1-2 let lane j of a group of lanes set up
// hypothesis d0 + j (projection, divisions, bilinear weights) and broadcast the eight results inside the quad (quad_perm DPP moves;
// ds_bpermute was tried too).  Next to kernels of OTHER queues (three hipGraph replays in flight) the stage-2/3 kernel written that
// way produced wrong similarities: single 16-lane rows of single loop iterations, always the ODD rows of a wave (lanes 16-31 /
// 48-63), always too small by one view's contribution; never when a pass ran alone.  Scratch / spills are excluded (every kernel
// of this file has a zero private segment), the DPP source lanes were valid (poison probe), the static code keeps the required
// wait states between the VALU writes and the DPP reads, and no readlane / readfirstlane sits on the tap path (DESIGN.md section 6
// has the evidence and what the ISA comparison of the failing and the passing form shows).  The cause below the ISA was not
// established, so the defect CLASS is closed instead: every lane now computes the set-ups it consumes (hypothesis-per-lane forms
// for the default kernels, plain recomputation for the generic / training kernels); only RESULTS (sums, maxima) cross lanes.
// The exchange forms are compiled only with -DEFFI_DIAG_LANE_EXCHANGE (diagnostic builds; never the shipped library).
#ifdef EFFI_DIAG_LANE_EXCHANGE
template <int G, int J>
__device__ __forceinline__ int quad_bcast_i(int v) {
    // source lane inside the quad for destination lanes 0..3
    constexpr int s0 = J, s1 = J, s2 = (G == 4) ? J : 2 + J, s3 = (G == 4) ? J : 2 + J;
    constexpr int ctrl = s0 | (s1 << 2) | (s2 << 4) | (s3 << 6);
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, true);
}
template <int G, int J>
__device__ __forceinline__ void taps_bcast(const Taps& mine, Taps& out) {
#ifdef EFFI_DPP_POISON
    // diagnostic build (tools/diag_in_flight8.py with EFFI_MVS_LIB): bound_ctrl off and a poisoned `old` operand -- a destination lane
    // whose source lane the hardware treats as invalid / disabled keeps the poison (weight 1000) instead of reading 0
    constexpr int s0 = J, s1 = J, s2 = (G == 4) ? J : 2 + J, s3 = (G == 4) ? J : 2 + J;
    constexpr int ctrl = s0 | (s1 << 2) | (s2 << 4) | (s3 << 6);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out.w[k] = __int_as_float(__builtin_amdgcn_update_dpp(0x447A0000, __float_as_int(mine.w[k]), ctrl, 0xF, 0xF, false));
        out.off[k] = __builtin_amdgcn_update_dpp(0, mine.off[k], ctrl, 0xF, 0xF, false);
    }
    return;
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out.w[k] = __int_as_float(quad_bcast_i<G, J>(__float_as_int(mine.w[k])));
        out.off[k] = quad_bcast_i<G, J>(mine.off[k]);
    }
}
#endif   // EFFI_DIAG_LANE_EXCHANGE

// (X, Y, Z) in the source camera -> 4 bilinear taps (weights zeroed when out of bounds).
__device__ __forceinline__ void make_taps(float X, float Y, float Z, int W, int H, int C, Taps& t) {
    if (Z == 0.0f) Z = Z + 1e-8f;                                   // models/module.py:328-329
    const float px = X / Z, py = Y / Z;                             // :330
    const float gx = px / ((float)(W - 1) / 2.0f) - 1.0f;           // :336
    const float gy = py / ((float)(H - 1) / 2.0f) - 1.0f;           // :337
    float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);               // grid_sampler_unnormalize
    float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    // neutral clamp: beyond these both taps of that axis are out of bounds anyway; also maps NaN to
    // "outside" and keeps the float->int conversion in range.
    ix = fminf(fmaxf(ix, -2.0f), (float)W + 1.0f);
    iy = fminf(fmaxf(iy, -2.0f), (float)H + 1.0f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = ix - x0f, wx0 = (x0f + 1.0f) - ix;
    const float wy1 = iy - y0f, wy0 = (y0f + 1.0f) - iy;
    const bool vx0 = (x0 >= 0) & (x0 <= W - 1), vx1 = (x0 + 1 >= 0) & (x0 + 1 <= W - 1);
    const bool vy0 = (y0 >= 0) & (y0 <= H - 1), vy1 = (y0 + 1 >= 0) & (y0 + 1 <= H - 1);
    const int xc0 = min(max(x0, 0), W - 1), xc1 = min(max(x0 + 1, 0), W - 1);
    const int yc0 = min(max(y0, 0), H - 1), yc1 = min(max(y0 + 1, 0), H - 1);
    t.w[0] = (vx0 & vy0) ? wx0 * wy0 : 0.0f;   // nw
    t.w[1] = (vx1 & vy0) ? wx1 * wy0 : 0.0f;   // ne
    t.w[2] = (vx0 & vy1) ? wx0 * wy1 : 0.0f;   // sw
    t.w[3] = (vx1 & vy1) ? wx1 * wy1 : 0.0f;   // se
    t.off[0] = (yc0 * W + xc0) * C;
    t.off[1] = (yc0 * W + xc1) * C;
    t.off[2] = (yc1 * W + xc0) * C;
    t.off[3] = (yc1 * W + xc1) * C;
}

__device__ __forceinline__ float dot4(const float4 a, const float4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// this lane's share (4 channels) of  sum_c ref[c] * bilinear(src[c])
__device__ __forceinline__ float sample_dot(const float* __restrict__ src, const Taps& t, int sub4, const float4 r4) {
    const float4 a = *reinterpret_cast<const float4*>(src + t.off[0] + sub4);
    const float4 b = *reinterpret_cast<const float4*>(src + t.off[1] + sub4);
    const float4 c = *reinterpret_cast<const float4*>(src + t.off[2] + sub4);
    const float4 d = *reinterpret_cast<const float4*>(src + t.off[3] + sub4);
    float s = t.w[0] * dot4(a, r4);
    s = fmaf(t.w[1], dot4(b, r4), s);
    s = fmaf(t.w[2], dot4(c, r4), s);
    s = fmaf(t.w[3], dot4(d, r4), s);
    return s;
}

template <int C>
__device__ __forceinline__ bool tile_pixel(int bid, int nblk, int h, int w, int& x, int& y, int& sub) {
    using G = WarpGeom<C>;
    const int tiles_x = (w + G::TW - 1) / G::TW;
    const int t = effi_xcd_remap(bid, nblk);
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int g = threadIdx.x / G::LPP;
    sub = threadIdx.x % G::LPP;
    x = tx * G::TW + (g % G::TW);
    y = ty * G::TH + (g / G::TW);
    return (x < w) & (y < h);
}

// ------------------------------------------------------------------------------------------------
// stage 1: per-view similarity volume + softmax entropy over D   (grid.y = source view)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void warpcorr_views_kernel(const float* ref_arg, EffiPtrList srcs,
                                                             const float* __restrict__ rt_all,
                                                             const float* __restrict__ depth, long dds, long dps,
                                                             int h, int w, int D, float* sim_views,
                                                             float* __restrict__ entropy) {
    using G = WarpGeom<C>;
    const float* __restrict__ ref = effi_resolve_views(ref_arg, srcs);
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int view = blockIdx.y;
    const float* __restrict__ src = pick_view(srcs, view);
    const float* __restrict__ rt = rt_all + view * 12;
    const int hw = h * w, pix = y * w + x, sub4 = 4 * sub;
    const float4 r4 = *reinterpret_cast<const float4*>(ref + (long)pix * C + sub4);
    const float fx = (float)x, fy = (float)y;
    const float rx = rt[0] * fx + rt[1] * fy + rt[2];               // rot . (x, y, 1)   module.py:324
    const float ry = rt[3] * fx + rt[4] * fy + rt[5];
    const float rz = rt[6] * fx + rt[7] * fy + rt[8];
    const float tx = rt[9], ty = rt[10], tz = rt[11];
    float* simv = sim_views + (long)view * D * hw + pix;
    const float* dp = depth + (long)pix * dps;
    float m = -INFINITY;
    // every lane sets up every hypothesis itself (no set-up crosses lanes: see the note at the top of this file); this is the generic
    // kernel (C != 32, per-pixel hypotheses, EFFI option warp_lds_kb = -1), not the cascade's default
    for (int d = 0; d < D; ++d) {
        const float dep = dp[d * dds];
        Taps t;
        make_taps(rx * dep + tx, ry * dep + ty, rz * dep + tz, w, h, C, t);        // module.py:325-327
        const float s = effi_group_sum<G::LPP>(sample_dot(src, t, sub4, r4)) / (float)C;   // mean over C, :40
        if ((d % G::LPP) == sub) simv[(long)d * hw] = s;
        m = fmaxf(m, s);
    }
    // softmax over D and entropy (models/Effi_MVS_plus.py:43-44); lane `sub` owns d = sub, sub+LPP, ...
    float z = 0.0f;
    for (int d = sub; d < D; d += G::LPP) z = z + expf(simv[(long)d * hw] - m);
    z = effi_group_sum<G::LPP>(z);
    float e = 0.0f;
    for (int d = sub; d < D; d += G::LPP) {
        const float p = expf(simv[(long)d * hw] - m) / z;
        e = e + (-p) * logf(p + 1e-7f);
    }
    e = effi_group_sum<G::LPP>(e);
    if (sub == 0) entropy[(long)view * hw + pix] = e;
}

// ------------------------------------------------------------------------------------------------
// stage 1, windowed form (C = 32, hypotheses shared by all pixels): the taps are served from LDS instead of the L1 / texture
// path.  The direct-gather kernel above issues S*D*h*w*4 wave-level 128-byte tap reads (2.9 GB at 148x200, D = 48, S = 4),
// which costs ~74 us at the L1's 64 B/clk/CU, and spends as long again on vector instructions (the per-hypothesis projection
// set-up is recomputed by both quads of a pixel, 8 lanes reduce a 4-channel partial dot product each).  Here:
//   * a 512-thread workgroup owns a 16 x 8 pixel tile of the reference map and ONE source view; 4 lanes per pixel, 8
//     channels per lane (two float4 per tap), the 4 lanes of a pixel each set up a different hypothesis and exchange the
//     taps with quad DPP moves -- no set-up is computed twice, half the cross-lane reduction steps.
//   * hypotheses are processed in chunks.  For a chunk [da, db) the source positions of the tile are confined to the convex
//     hull of the tile's four corners projected at the chunk's first and last depth (for a fixed depth the map is a
//     homography: convex -> convex while Z > 0; for a fixed pixel the position moves monotonically along the epipolar line;
//     Z is multilinear in (pixel, depth), so Z > 0 at the 8 corner cases means Z > 0 throughout).  Wave 0 evaluates EIGHT
//     candidate chunk lengths at once (8 lanes per candidate, one corner case each) and picks the longest whose bounding
//     box (+1 pixel of slack each side, clipped to the image) fits the LDS window; all rows of the box are then copied with
//     coalesced 128-byte-per-pixel reads (channel-last layout: a window row is one contiguous run) and the chunk is
//     sampled with ds_read_b128 (256 B/clk/CU against the L1's 64).
//   * a chunk whose box does not fit (extreme geometry, Z <= 0 inside the tile) is sampled from global memory by the same
//     code: identical results bit for bit, which is what tests/test_gpu_kernels.py checks (EFFI_WARP_LDS_KB=0 forces it).
//   Measured (148x200, D = 48, S = 4): 92-95 us against 116 us for the direct-gather kernel and 125 us for this kernel on
//   global loads.  Ablation on the same box: blend + dot 20 us, per-hypothesis set-up 8 us, window copies 15 us (exposed:
//   the two workgroups of a CU run in step), the remaining skeleton (LDS reads at 2.9 GB / 124 TB/s = 27 us with 17 %
//   bank conflicts, tap exchange, address arithmetic, the similarity stores, the entropy epilogue) 59 us; the SIMDs issue
//   ~100 % of the time (SQ_ACTIVE_INST_ANY ~ kernel cycles per SIMD), i.e. the kernel is now bound by instruction issue.
//   * LDS layout: window pixel P holds its 32 channels as 8 float4; float4 q sits at slot (8P + q) ^ ((P >> 1) & 1).  Lane
//     `sub` reads logical q = 2sub and 2sub + 1, i.e. slots A = (8P + 2sub) ^ s and A ^ 1.  A ds_read_b128 is served in
//     groups of 16 lanes = 4 pixels; pixel P's four lanes touch banks 32(P&1) + 8sub + 4s + {0..3}, so the four pixels of a
//     group are conflict-free when their P differ mod 4 -- true for the near-unit-scale maps of neighbouring views.
// Arithmetic: as the reference op for op, except that the four divisions of the projection (X/Z, Y/Z, and the grid
// normalisation by (W-1)/2, (H-1)/2) use a refined reciprocal and one FMA residual step (3 instructions each instead of the
// ~10 of the IEEE sequence): correctly rounded except for rare 1-ulp cases, i.e. ~1e-7 relative in a CONTINUOUS function of
// the coordinate (bilinear sampling with zeros padding has no jumps), far inside the kernel tolerance.
// ------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float effi_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) effi_f4* effi_gptr4;      // a pointer KNOWN to be global memory (global_load, not flat_load)
typedef short effi_s16x2 __attribute__((ext_vector_type(2)));

struct WinTaps {
    float w[4];
    int a[4];     // LDS form: byte address of the tap pixel's swizzled slot 0, ((8P) ^ s) * 16; global form: pixel index y*W + x
};

__device__ __forceinline__ float div_by(float a, float b, float rb) {      // a / b given rb ~ 1/b (refined reciprocal)
    const float q = a * rb;
    return fmaf(fmaf(-q, b, a), rb, q);
}
__device__ __forceinline__ float refined_rcp(float b) {
    const float r0 = __builtin_amdgcn_rcpf(b);
    return fmaf(fmaf(-b, r0, 1.0f), r0, r0);
}

// Unclamped source coordinates (ix, iy) of a reference pixel at one depth -- models/module.py:325-337 + grid_sampler_unnormalize
__device__ __forceinline__ void project_xy(float X, float Y, float Z, float hw2, float rhw2, float hh2, float rhh2, float wm1,
                                           float hm1, float& ix, float& iy) {
    if (Z == 0.0f) Z = Z + 1e-8f;                                   // :328-329
    const float rz = refined_rcp(Z);
    const float px = div_by(X, Z, rz), py = div_by(Y, Z, rz);       // :330
    const float gx = div_by(px, hw2, rhw2) - 1.0f;                  // :336
    const float gy = div_by(py, hh2, rhh2) - 1.0f;                  // :337
    ix = ((gx + 1.0f) * 0.5f) * wm1;
    iy = ((gy + 1.0f) * 0.5f) * hm1;
}

template <bool LDSWIN>
__device__ __forceinline__ void make_taps_win(float ix, float iy, int W, int H, int x_lo, int y_lo, int ww, int wh, WinTaps& t) {
    ix = fminf(fmaxf(ix, -2.0f), (float)W + 1.0f);                  // neutral clamp (see make_taps); NaN -> outside
    iy = fminf(fmaxf(iy, -2.0f), (float)H + 1.0f);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    // 1-D weights, zeroed when their tap column / row is outside the image (zeros padding).  The product of two masked
    // factors equals the masked product: the factors lie in [0, 1], so a zero factor gives +0.
    const float wx0 = ((unsigned)x0 < (unsigned)W) ? (x0f + 1.0f) - ix : 0.0f;
    const float wx1 = ((unsigned)(x0 + 1) < (unsigned)W) ? ix - x0f : 0.0f;
    const float wy0 = ((unsigned)y0 < (unsigned)H) ? (y0f + 1.0f) - iy : 0.0f;
    const float wy1 = ((unsigned)(y0 + 1) < (unsigned)H) ? iy - y0f : 0.0f;
    t.w[0] = wx0 * wy0;
    t.w[1] = wx1 * wy0;
    t.w[2] = wx0 * wy1;
    t.w[3] = wx1 * wy1;
    if (LDSWIN) {
        // window-relative, clamped INTO the window: a no-op for every tap that carries weight (the window covers them),
        // and zero-weight taps then read finite staged data (0 * finite = 0), never uninitialised LDS
        const int xa = min(max(x0 - x_lo, 0), ww - 1), xb = min(max(x0 + 1 - x_lo, 0), ww - 1);
        const int ya = __mul24(min(max(y0 - y_lo, 0), wh - 1), ww), yb = __mul24(min(max(y0 + 1 - y_lo, 0), wh - 1), ww);
        const int p0 = ya + xa, p1 = ya + xb, p2 = yb + xa, p3 = yb + xb;
        t.a[0] = ((p0 << 3) ^ ((p0 >> 1) & 1)) << 4;
        t.a[1] = ((p1 << 3) ^ ((p1 >> 1) & 1)) << 4;
        t.a[2] = ((p2 << 3) ^ ((p2 >> 1) & 1)) << 4;
        t.a[3] = ((p3 << 3) ^ ((p3 >> 1) & 1)) << 4;
    } else {
        const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1);
        const int ya = __mul24(min(max(y0, 0), H - 1), W), yb = __mul24(min(max(y0 + 1, 0), H - 1), W);
        t.a[0] = ya + xa; t.a[1] = ya + xb; t.a[2] = yb + xa; t.a[3] = yb + xb;
    }
}

// The 32 reference channels of a pixel as this lane sees them: ROTATED by the lane's position in its pixel's quad -- position i
// holds channels 8 ((sub + i) & 3) .. + 7 -- so that at step i the four lanes of a pixel read four DIFFERENT 32-byte pieces of
// their taps: the LDS image and its bank behaviour are those of the channel-split form (a ds_read_b128 lane group = 4 pixels x 4
// lanes is conflict-free when the four tap pixels differ mod 4), although every lane now walks all 32 channels of ITS OWN hypothesis.
struct RefRot {
    float4 lo[4], hi[4];
};

constexpr int WIN_TW = 16, WIN_TH = 8, WIN_THREADS = 512;

// all threads of the workgroup (after hyp[] is visible): are the D <= 256 hypotheses monotone (non-decreasing or non-increasing)?
__device__ __forceinline__ bool win_hyp_monotone(const float* __restrict__ hyp, int D, int tid) {
    const bool mine = tid + 1 < D;
    const float a = mine ? hyp[tid] : 0.0f, b = mine ? hyp[tid + 1] : 0.0f;
    const int inc = __syncthreads_and(!mine || b >= a), dec = __syncthreads_and(!mine || b <= a);
    return inc || dec;
}

struct WinProj {                // per-thread projection state of its pixel and view
    float rx, ry, rz, tx, ty, tz, hw2, rhw2, hh2, rhh2, wm1, hm1;
};

// One chunk of hypotheses for this lane's pixel, HYPOTHESIS-PER-LANE: lane `sub` of the pixel's quad owns hypothesis d0 + sub with
// all 32 channels -- its set-up (projection, weights, tap addresses) is computed once, by the lane that uses it, and nothing but the
// finished similarity leaves the lane.  (Rounds 1-2: the four lanes split the CHANNELS of every hypothesis and passed the set-ups
// round with quad_perm DPP moves; see the note at the top of this file.)  Per hypothesis a lane issues 32 ds_read_b128 (4 taps x 8
// float4) -- the same LDS traffic per pixel as before -- and 80 packed FMAs; gone are the 32 DPP moves and 8 DPP adds per group of
// four hypotheses.  Blend first, as grid_sample does, then the product with the reference features.
template <bool LDSWIN>
__device__ __forceinline__ void win_sample_chunk(const char* __restrict__ win, const float* __restrict__ src, const WinProj& P,
                                                 const float* __restrict__ hyp, int da, int db, int D, int w, int h,
                                                 int x_lo, int y_lo, int ww, int wh, int sub, const RefRot& R,
                                                 bool valid, float* __restrict__ simv, int hw, float& m) {
    int rot[4];                                   // byte offset of this lane's i-th 32-byte piece inside a pixel's 128 bytes
#pragma unroll
    for (int i = 0; i < 4; ++i) rot[i] = ((sub + i) & 3) * 32;
    for (int d0 = da; d0 < db; d0 += 4) {
        const float dep = hyp[d0 + sub];      // this lane's hypothesis; hyp[] is padded with the last one to a multiple of 4
        float ix, iy;
        project_xy(P.rx * dep + P.tx, P.ry * dep + P.ty, P.rz * dep + P.tz, P.hw2, P.rhw2, P.hh2, P.rhh2, P.wm1, P.hm1, ix, iy);
        WinTaps t;
        make_taps_win<LDSWIN>(ix, iy, w, h, x_lo, y_lo, ww, wh, t);
        f32x2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // this step's 8 channels of the four taps, two taps at a time (16 registers of tap data in flight, not 32: with the 32
            // reference registers the kernel must stay under the 128-register cap of two 512-thread workgroups per CU; the other
            // three waves of the SIMD hide the LDS latency); blend in tap order, as grid_sample does
            f32x2 v[4];
#pragma unroll
            for (int kp = 0; kp < 4; kp += 2) {
                float4 lo[2], hi[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (LDSWIN) {
                        // logical float4 q of window pixel P sits at slot (8P + q) ^ s: even q at t.a + 16 q, its odd partner at that ^ 16
                        const int a = t.a[kp + k] + rot[i];
                        lo[k] = *reinterpret_cast<const float4*>(win + a);
                        hi[k] = *reinterpret_cast<const float4*>(win + (a ^ 16));
                    } else {
                        const float4* q = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(src) + ((long)t.a[kp + k] * 128 + rot[i]));
                        lo[k] = q[0];
                        hi[k] = q[1];
                    }
                }
#define EFFI_BLEND2(dst, SRC, F0, F1)                                                                                   \
                dst = (kp == 0) ? f32x2{SRC[0].F0, SRC[0].F1} * f32x2{t.w[0], t.w[0]}                                          \
                                : __builtin_elementwise_fma(f32x2{SRC[0].F0, SRC[0].F1}, f32x2{t.w[kp], t.w[kp]}, dst);        \
                dst = __builtin_elementwise_fma(f32x2{SRC[1].F0, SRC[1].F1}, f32x2{t.w[kp + 1], t.w[kp + 1]}, dst)
                EFFI_BLEND2(v[0], lo, x, y);
                EFFI_BLEND2(v[1], lo, z, w);
                EFFI_BLEND2(v[2], hi, x, y);
                EFFI_BLEND2(v[3], hi, z, w);
#undef EFFI_BLEND2

            }
            acc = (i == 0) ? v[0] * f32x2{R.lo[0].x, R.lo[0].y} : __builtin_elementwise_fma(v[0], f32x2{R.lo[i].x, R.lo[i].y}, acc);
            acc = __builtin_elementwise_fma(v[1], f32x2{R.lo[i].z, R.lo[i].w}, acc);
            acc = __builtin_elementwise_fma(v[2], f32x2{R.hi[i].x, R.hi[i].y}, acc);
            acc = __builtin_elementwise_fma(v[3], f32x2{R.hi[i].z, R.hi[i].w}, acc);
        }
        const float sv = (acc.x + acc.y) * (1.0f / 32.0f);          // mean over C (:40); x 2^-5 is exact
        if (d0 + sub < D) {
            if (valid) simv[(long)(d0 + sub) * hw] = sv;
            m = fmaxf(m, sv);
        }
    }
}

// wave 0: candidate k = lane >> 3 proposes the chunk [da, da + cs_k); its 8 lanes project the tile's corners at both ends of it;
// the longest candidate whose box fits the window is published in par[0..5] = {x_lo, y_lo, ww, wh, db, use_lds}
// mono: the hypotheses are monotone (checked once per workgroup, win_hyp_monotone): only then do the chunk's two END depths bound
// the positions of the depths between them; otherwise every chunk is sampled from global memory (same arithmetic, any order of
// hypotheses -- the reference accepts any depth_values)
__device__ __forceinline__ void win_choose_chunk(const float* __restrict__ rt, const float* __restrict__ hyp, const WinProj& P,
                                                 int txi, int tyi, int da, int D, int w, int h, int lds_px, int lane, int* par,
                                                 bool mono) {
    const int ng = (D + 3) >> 2;
    const int k = lane >> 3, corner = lane & 7;
    const int rem_g = (D - da + 3) >> 2;                        // groups of 4 hypotheses still to do
    const int cs = 4 * max(1, (rem_g * (8 - k) + 7) >> 3);      // 8/8 ... 1/8 of what is left, in whole groups
    const int db = min(D, da + cs);
    const float dep = hyp[(corner & 4) ? db - 1 : da];
    const float cxf = (float)min(txi * WIN_TW + ((corner & 1) ? WIN_TW - 1 : 0), w - 1);
    const float cyf = (float)min(tyi * WIN_TH + ((corner & 2) ? WIN_TH - 1 : 0), h - 1);
    const float X = (rt[0] * cxf + rt[1] * cyf + rt[2]) * dep + rt[9];
    const float Y = (rt[3] * cxf + rt[4] * cyf + rt[5]) * dep + rt[10];
    const float Z = (rt[6] * cxf + rt[7] * cyf + rt[8]) * dep + rt[11];
    float ix, iy;
    project_xy(X, Y, Z, P.hw2, P.rhw2, P.hh2, P.rhh2, P.wm1, P.hm1, ix, iy);
    int ok = (Z > 0.0f) & (ix == ix) & (iy == iy);
    ix = fminf(fmaxf(ix, -4.0f), (float)w + 3.0f);              // monotonic: bounding boxes survive the clamp
    iy = fminf(fmaxf(iy, -4.0f), (float)h + 3.0f);
    float mnx = ix, mxx = ix, mny = iy, mxy = iy;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o));
        mny = fminf(mny, __shfl_xor(mny, o)); mxy = fmaxf(mxy, __shfl_xor(mxy, o));
        ok &= __shfl_xor(ok, o);
    }
    // taps are floor(.) and floor(.) + 1; one more pixel of slack on every side for rounding
    int x_lo = (int)floorf(mnx) - 1, x_hi = (int)floorf(mxx) + 2;
    int y_lo = (int)floorf(mny) - 1, y_hi = (int)floorf(mxy) + 2;
    x_lo = min(max(x_lo, 0), w - 1); x_hi = max(min(x_hi, w - 1), x_lo);
    y_lo = min(max(y_lo, 0), h - 1); y_hi = max(min(y_hi, h - 1), y_lo);
    const int ww = x_hi - x_lo + 1, wh = y_hi - y_lo + 1;
    const bool fits = ok && mono && (ww * wh <= lds_px);
    const unsigned long long mask = __ballot(fits && corner == 0);
    const int pick = mask ? (int)(__ffsll((long long)mask) - 1) : 56;      // longest fitting chunk, else the shortest
    if (lane == pick) {
        par[0] = x_lo; par[1] = y_lo; par[2] = ww; par[3] = wh; par[4] = db; par[5] = mask ? 1 : 0;
    }
    (void)ng;
}

// The kernel's body takes its two maps as restrict-qualified PARAMETERS: both launch forms (pointers in the kernel arguments /
// pointers read from a device table, TBL) then compile to the same code.  With the table's loads feeding plain locals this kernel,
// which sits at its 128-register cap, lost the no-alias information and spilled 96 bytes per lane.
template <int MAXPX>
__device__ __forceinline__ void warpcorr_views_win_body(const float* __restrict__ ref, const float* __restrict__ src,
                                                        const float* __restrict__ rt_all,
                                                        const float* __restrict__ depth, long dds,
                                                        int h, int w, int D, float* sim_views,
                                                        float* __restrict__ entropy, int lds_px) {
    constexpr int C = 32, MAXD = 256;
    __shared__ float4 win4[(MAXPX > 0 ? MAXPX : 1) * 8];
    __shared__ float hyp[MAXD + 4];
    __shared__ int wpar[2][8];
    const char* win = reinterpret_cast<const char*>(win4);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (w + WIN_TW - 1) / WIN_TW;
    const int tl = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tyi = tl / tiles_x, txi = tl - tyi * tiles_x;
    const int g = tid >> 2, sub = tid & 3;
    const int x = txi * WIN_TW + (g & (WIN_TW - 1)), y = tyi * WIN_TH + (g >> 4);
    const bool valid = (x < w) & (y < h);
    const int xs = min(x, w - 1), ys = min(y, h - 1);          // out-of-image lanes shadow the border pixel (no stores)
    const int view = blockIdx.y;
    const float* __restrict__ rt = rt_all + view * 12;
    const int hw = h * w, pix = ys * w + xs;
    // the hypotheses (shared by all pixels) once into LDS, padded with the last one to a multiple of 4
    for (int d = tid; d < ((D + 3) & ~3); d += WIN_THREADS) hyp[d] = depth[(long)min(d, D - 1) * dds];
    WinProj P;
    {
        const float fx = (float)xs, fy = (float)ys;
        P.rx = rt[0] * fx + rt[1] * fy + rt[2];                  // rot . (x, y, 1)   module.py:324
        P.ry = rt[3] * fx + rt[4] * fy + rt[5];
        P.rz = rt[6] * fx + rt[7] * fy + rt[8];
        P.tx = rt[9]; P.ty = rt[10]; P.tz = rt[11];
        P.wm1 = (float)(w - 1); P.hm1 = (float)(h - 1);
        P.hw2 = P.wm1 / 2.0f; P.hh2 = P.hm1 / 2.0f;
        P.rhw2 = 1.0f / P.hw2; P.rhh2 = 1.0f / P.hh2;           // IEEE divisions: correctly rounded reciprocals
    }
    float* simv = sim_views + (long)view * D * hw + pix;
    float m = -INFINITY;
    __syncthreads();                                             // hyp[] visible
    const bool mono = win_hyp_monotone(hyp, D, tid);
    if (wv == 0) win_choose_chunk(rt, hyp, P, txi, tyi, 0, D, w, h, lds_px, lane, wpar[0], mono);
    __syncthreads();
    int da = 0, pb = 0;
    while (da < D) {
        const int x_lo = wpar[pb][0], y_lo = wpar[pb][1], ww = wpar[pb][2], wh = wpar[pb][3], db = wpar[pb][4], use_lds = wpar[pb][5];
        if (use_lds) {
            // copy the window: rows are contiguous runs of ww * 8 float4 in the channel-last map; up to 5 loads in flight per thread
            const int n4 = ww * 8, total = n4 * wh;
            const float inv_n4 = 1.0f / (float)n4;
            const float4* __restrict__ gsrc = reinterpret_cast<const float4*>(src + ((long)y_lo * w + x_lo) * C);
            for (int i0 = tid; i0 < total; i0 += 5 * WIN_THREADS) {
                float4 v[5];
                int slot[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int i = min(i0 + j * WIN_THREADS, total - 1);
                    int row = (int)(((float)i + 0.5f) * inv_n4);           // i < 2^13: exact up to +-1, fixed below
                    row -= (row * n4 > i);
                    row += ((row + 1) * n4 <= i);
                    const int c = i - row * n4;
                    const int Pp = row * ww + (c >> 3);
                    slot[j] = (Pp * 8 + (c & 7)) ^ ((Pp >> 1) & 1);
                    v[j] = gsrc[(long)row * w * 8 + c];
                }
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (i0 + j * WIN_THREADS < total) win4[slot[j]] = v[j];
            }
            __syncthreads();
        }
        // the next chunk's window is chosen by wave 0 before it joins the sampling (published by the barrier below)
        if (wv == 0 && db < D) win_choose_chunk(rt, hyp, P, txi, tyi, db, D, w, h, lds_px, lane, wpar[pb ^ 1], mono);
        // the pixel's reference features (rotated, see RefRot) are fetched per chunk, AFTER the window copy: 32 registers that must not
        // stay live across the copy loop under the 128-register cap (128 L1-resident bytes per lane, 1-3 times per workgroup)
        RefRot R;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* rp = ref + (long)pix * C + ((sub + i) & 3) * 8;
            R.lo[i] = *reinterpret_cast<const float4*>(rp);
            R.hi[i] = *reinterpret_cast<const float4*>(rp + 4);
        }
        if (use_lds)
            win_sample_chunk<true>(win, src, P, hyp, da, db, D, w, h, x_lo, y_lo, ww, wh, sub, R, valid, simv, hw, m);
        else
            win_sample_chunk<false>(win, src, P, hyp, da, db, D, w, h, 0, 0, w, h, sub, R, valid, simv, hw, m);
        __syncthreads();           // the window is rewritten by the next chunk; wpar[pb ^ 1] is published
        da = db;
        pb ^= 1;
    }
    m = effi_group_max<4>(m);          // a lane saw its own hypotheses only: maximum over the pixel's four lanes (a RESULT crosses lanes)
    if (!valid) return;
    // softmax over D and entropy (models/Effi_MVS_plus.py:43-44); lane `sub` owns (and wrote) d = sub, sub + 4, ...
    // exp(s - m) is evaluated once and parked in the (now free) window, [i][thread]: conflict-free, no second exp pass
    float* ebuf = reinterpret_cast<float*>(win4) + tid;
    const bool park = ((D + 3) >> 2) * WIN_THREADS <= (MAXPX > 0 ? MAXPX : 1) * 32;
    float z = 0.0f;
    for (int d = sub, i = 0; d < D; d += 4, ++i) {
        const float ex = expf(simv[(long)d * hw] - m);
        if (park) ebuf[i * WIN_THREADS] = ex;
        z = z + ex;
    }
    z = effi_group_sum<4>(z);
    float e = 0.0f;
    for (int d = sub, i = 0; d < D; d += 4, ++i) {
        const float p = (park ? ebuf[i * WIN_THREADS] : expf(simv[(long)d * hw] - m)) / z;
        e = e + (-p) * logf(p + 1e-7f);
    }
    e = effi_group_sum<4>(e);
    if (sub == 0) entropy[(long)view * hw + pix] = e;
}

template <int MAXPX, bool TBL>
__global__ __launch_bounds__(WIN_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void warpcorr_views_win_kernel(
    const float* __restrict__ ref_arg, EffiPtrList srcs, const float* __restrict__ rt_all, const float* __restrict__ depth, long dds,
    int h, int w, int D, float* sim_views, float* __restrict__ entropy, int lds_px) {
    const int view = blockIdx.y;
    warpcorr_views_win_body<MAXPX>(TBL ? srcs.tbl[0] : ref_arg, TBL ? srcs.tbl[1 + view] : pick_view_list(srcs, view), rt_all, depth, dds,
                                   h, w, D, sim_views, entropy, lds_px);
}

// ------------------------------------------------------------------------------------------------
// stage 1 in SPLIT precision, CORRELATE FIRST (round 4; C = 32, hypotheses shared by all pixels, precision "split" / "bf16" only --
// the exact-fp32 mode and training keep the kernel above).  BUILT, CORRECT (tests/test_gpu_kernels.py::test_warpcorr_views_matrix_core_form),
// AND SLOWER THAN THE WINDOW KERNEL: 147 vs 93 us at 148x200, D = 48, S = 4 -- NOT the default (Python option warp_x3 = 1 selects it).
// Why: the box of a chunk is dominated by the 16-pixel width of the row segment, not by the chunk's disparity
// spread, so every chunk re-reads and re-splits almost the same source pixels (36-96 blocks of 16 x 16 per row segment and view
// instead of the ~10 one box over all hypotheses would need -- which does not fit the LDS region once the epipolar line is slanted);
// the per-block split (24 instructions) then costs as much per hypothesis as the window kernel's blend.  What it would take: source
// features split to bf16 hi / lo once per image, and a sheared box that follows the epipolar line.  Correlation is linear in the warped feature:
//     sum_c ref[c] * (sum_k w_k src[tap_k][c])  =  sum_k w_k * (sum_c ref[c] * src[tap_k][c])  =  sum_k w_k G[p][tap_k],
// and the 192 taps of a pixel's 48 hypotheses in one view lie on one short epipolar segment whose ~100 distinct source pixels are
// shared with the pixel's neighbours.  So a wave takes a row segment of 16 reference pixels and, per chunk of 16 hypotheses:
//   A. every lane (pixel p = lane & 15, hypothesis slot lane >> 4) projects its four hypotheses (the arithmetic of the kernel above);
//      the wave reduces the bounding box of the tap pixels (packed 16-bit min / max: DPP inside a row of lanes, shuffles across);
//   B. G[16 reference pixels][box pixels] on the matrix cores: per 16 box pixels of one source row ONE coalesced 2-KB read (a lane
//      fetches 8 consecutive channels of one pixel: 16 pixels x 128 B contiguous in the channel-last map), split into bf16 hi + lo
//      in registers, and three v_mfma_f32_16x16x32_bf16 (hi*hi + lo*hi + hi*lo, K = 32 = all channels, fp32 accumulation: the
//      precision of the path's convolutions); the 16 x 16 block goes to the wave's own LDS region -- no source window in LDS at all;
//   C. a hypothesis is then 4 LDS dwords and 4 FMAs.
// Per (pixel, hypothesis, view) this is ~75 vector instructions and 16 LDS bytes against ~366 instructions and 512 LDS bytes of the
// window kernel.  A chunk whose box does not fit the wave's LDS region (a steep epipolar line: extreme geometry) is evaluated
// directly from global memory (blend of the four taps, then the product -- fp32).  No workgroup barrier inside the chunk loop: the
// four waves of a workgroup only share the hypothesis table.
// ------------------------------------------------------------------------------------------------
constexpr int MM_NQ = 144;             // box pixels per wave and chunk (16 x 145 floats = 9.3 KB per wave: four workgroups per CU)
constexpr int MM_CH = 16;              // hypotheses per full chunk (four per lane); a chunk shrinks to 8 / 4 when its box does not fit

__device__ __forceinline__ int mm_dpp_min16(int v, int o) { effi_s16x2 a = __builtin_bit_cast(effi_s16x2, v), b = __builtin_bit_cast(effi_s16x2, o); return __builtin_bit_cast(int, __builtin_elementwise_min(a, b)); }
__device__ __forceinline__ int mm_dpp_max16(int v, int o) { effi_s16x2 a = __builtin_bit_cast(effi_s16x2, v), b = __builtin_bit_cast(effi_s16x2, o); return __builtin_bit_cast(int, __builtin_elementwise_max(a, b)); }
// wave-wide minimum / maximum of two packed 16-bit pairs (every lane receives the result)
__device__ __forceinline__ void mm_wave_box(int& lo, int& hi) {
#define EFFI_MM_STEP(MOVE) { const int lo_o = MOVE(lo), hi_o = MOVE(hi); lo = mm_dpp_min16(lo, lo_o); hi = mm_dpp_max16(hi, hi_o); }
#define EFFI_MM_B1(v) __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true)
#define EFFI_MM_4E(v) __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true)
#define EFFI_MM_141(v) __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true)
#define EFFI_MM_140(v) __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true)
#define EFFI_MM_S16(v) __shfl_xor(v, 16)
#define EFFI_MM_S32(v) __shfl_xor(v, 32)
    EFFI_MM_STEP(EFFI_MM_B1) EFFI_MM_STEP(EFFI_MM_4E) EFFI_MM_STEP(EFFI_MM_141) EFFI_MM_STEP(EFFI_MM_140) EFFI_MM_STEP(EFFI_MM_S16) EFFI_MM_STEP(EFFI_MM_S32)
#undef EFFI_MM_STEP
#undef EFFI_MM_B1
#undef EFFI_MM_4E
#undef EFFI_MM_141
#undef EFFI_MM_140
#undef EFFI_MM_S16
#undef EFFI_MM_S32
}

typedef __bf16 mm_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mm_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mm_split8(const effi_f4& a, const effi_f4& b, mm_bf16x8& hi, mm_bf16x8& lo) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2 v = {x[e], x[e + 1]};
        const mm_bf16x2 h2 = __builtin_convertvector(v, mm_bf16x2);
        const mm_bf16x2 l2 = __builtin_convertvector(v - __builtin_convertvector(h2, f32x2), mm_bf16x2);
        hi[e] = h2[0]; hi[e + 1] = h2[1];
        lo[e] = l2[0]; lo[e + 1] = l2[1];
    }
}

template <bool HI_ONLY>
__device__ __forceinline__ void warpcorr_views_mm_body(const float* __restrict__ ref, const float* __restrict__ src,
                                                       const float* __restrict__ rt_all, const float* __restrict__ depth, long dds,
                                                       int h, int w, int D, float* sim_views, float* __restrict__ entropy) {
    constexpr int C = 32, MAXD = 256, GS = MM_NQ + 1;                // G row stride per reference pixel (odd: bank spread)
    __shared__ float gbuf[4][16 * GS];
    __shared__ float hyp[MAXD + MM_CH];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lk = lane >> 4;
    const int tiles_x = (w + 15) / 16;
    const int tl = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tyi = tl / tiles_x, txi = tl - tyi * tiles_x;
    const int x = txi * 16 + li, y = tyi * 4 + wvi;
    const bool valid = (x < w) & (y < h);
    const int xs = min(x, w - 1), ys = min(y, h - 1);          // out-of-image lanes shadow the border pixel (no stores)
    const int view = blockIdx.y;
    const float* __restrict__ rt = rt_all + view * 12;
    const int hw = h * w, pix = ys * w + xs;
    for (int d = tid; d < ((D + MM_CH - 1) / MM_CH) * MM_CH; d += 256) hyp[d] = depth[(long)min(d, D - 1) * dds];
    WinProj P;
    {
        const float fx = (float)xs, fy = (float)ys;
        P.rx = rt[0] * fx + rt[1] * fy + rt[2];
        P.ry = rt[3] * fx + rt[4] * fy + rt[5];
        P.rz = rt[6] * fx + rt[7] * fy + rt[8];
        P.tx = rt[9]; P.ty = rt[10]; P.tz = rt[11];
        P.wm1 = (float)(w - 1); P.hm1 = (float)(h - 1);
        P.hw2 = P.wm1 / 2.0f; P.hh2 = P.hm1 / 2.0f;
        P.rhw2 = 1.0f / P.hw2; P.rhh2 = 1.0f / P.hh2;
    }
    // A operand: this lane's reference pixel li, channels 8 lk .. 8 lk + 7, as bf16 hi + lo
    mm_bf16x8 rh, rl;
    {
        const effi_f4* rp = reinterpret_cast<const effi_f4*>(ref + (long)pix * C + 8 * lk);
        mm_split8(rp[0], rp[1], rh, rl);
    }
    float* simv = sim_views + (long)view * D * hw + pix;
    float* G = gbuf[wvi];
    const effi_gptr4 sb4 = (effi_gptr4)src;
    float m = -INFINITY;
    __syncthreads();                                             // hyp[] visible (the only workgroup barrier)
    constexpr int NG = MM_CH / 4, MAXB = MM_NQ / 16;              // hypothesis groups of a full chunk; 16 x 16 blocks of a full box
    for (int d0 = 0; d0 < D;) {
        // ---- A: positions of this lane's hypotheses d0 + 4 g + lk; the LONGEST prefix of groups whose box fits the wave's region
        float ix[NG], iy[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float dep = hyp[d0 + 4 * g + lk];                  // padded with the last hypothesis to a whole chunk
            project_xy(P.rx * dep + P.tx, P.ry * dep + P.ty, P.rz * dep + P.tz, P.hw2, P.rhw2, P.hh2, P.rhh2, P.wm1, P.hm1, ix[g], iy[g]);
            ix[g] = fminf(fmaxf(ix[g], -2.0f), (float)w + 1.0f);       // neutral clamp (make_taps); NaN -> outside
            iy[g] = fminf(fmaxf(iy[g], -2.0f), (float)h + 1.0f);
        }
        int ng = min(NG, (D - d0 + 3) >> 2), x_lo = 0, y_lo = 0, nrows = 1, nb = 1;
        bool fits = false;
        for (;;) {
            float mnx = ix[0], mxx = ix[0], mny = iy[0], mxy = iy[0];
#pragma unroll
            for (int g = 1; g < NG; ++g)
                if (g < ng) { mnx = fminf(mnx, ix[g]); mxx = fmaxf(mxx, ix[g]); mny = fminf(mny, iy[g]); mxy = fmaxf(mxy, iy[g]); }
            const effi_s16x2 lo2 = {(short)(int)floorf(mnx), (short)(int)floorf(mny)};
            const effi_s16x2 hi2 = {(short)((int)floorf(mxx) + 1), (short)((int)floorf(mxy) + 1)};
            int lo_i = __builtin_bit_cast(int, lo2), hi_i = __builtin_bit_cast(int, hi2);
            mm_wave_box(lo_i, hi_i);
            const int lo_u = __builtin_amdgcn_readfirstlane(lo_i), hi_u = __builtin_amdgcn_readfirstlane(hi_i);
            x_lo = min(max((int)(short)(lo_u & 0xffff), 0), w - 1);
            const int x_hi = max(min((int)(short)(hi_u & 0xffff), w - 1), x_lo);
            y_lo = min(max(lo_u >> 16, 0), h - 1);
            const int y_hi = max(min(hi_u >> 16, h - 1), y_lo);
            nrows = y_hi - y_lo + 1;
            nb = (x_hi - x_lo + 16) >> 4;
            fits = nrows * nb <= MAXB;
            if (fits || ng == 1) break;
            ng >>= 1;                                                // (the positions of the dropped groups are set up again next round)
        }
        const int pitch = nb * 16;
        if (fits) {
            // ---- B: G[reference pixel][box pixel] = sum_c ref * src, 16 x 16 blocks on the matrix cores; ALL reads of the box first
            effi_f4 s0[MAXB], s1[MAXB];
            const int nblk = nrows * nb;
            {
                int r = 0, b = 0;
#pragma unroll
                for (int k = 0; k < MAXB; ++k) {
                    if (k < nblk) {
                        const int qx = min(x_lo + 16 * b + li, w - 1);   // beyond the box / the image: a finite value nobody weights
                        const effi_gptr4 q = sb4 + ((long)(y_lo + r) * w + qx) * (C / 4) + 2 * lk;
                        s0[k] = q[0];
                        s1[k] = q[1];
                    }
                    if (++b == nb) { b = 0; ++r; }
                }
            }
            {
                int r = 0, b = 0;
#pragma unroll
                for (int k = 0; k < MAXB; ++k) {
                    if (k < nblk) {
                        mm_bf16x8 sh, sl;
                        mm_split8(s0[k], s1[k], sh, sl);
                        effi_f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rh, sh, acc, 0, 0, 0);
                        if (!HI_ONLY) {
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rl, sh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rh, sl, acc, 0, 0, 0);
                        }
                        float* gp = G + (4 * lk) * GS + r * pitch + 16 * b + li;       // D[reference pixel 4 lk + i][box pixel li]
#pragma unroll
                        for (int i = 0; i < 4; ++i) gp[i * GS] = acc[i];
                    }
                    if (++b == nb) { b = 0; ++r; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- C: this lane's hypotheses of the accepted groups
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= ng) break;
            const int d = d0 + 4 * g + lk;
            const float x0f = floorf(ix[g]), y0f = floorf(iy[g]);
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float wx0 = ((unsigned)x0 < (unsigned)w) ? (x0f + 1.0f) - ix[g] : 0.0f;
            const float wx1 = ((unsigned)(x0 + 1) < (unsigned)w) ? ix[g] - x0f : 0.0f;
            const float wy0 = ((unsigned)y0 < (unsigned)h) ? (y0f + 1.0f) - iy[g] : 0.0f;
            const float wy1 = ((unsigned)(y0 + 1) < (unsigned)h) ? iy[g] - y0f : 0.0f;
            const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
            float sv;
            if (fits) {
                // clamped INTO the box: a no-op for every tap that carries weight
                const int xa = min(max(x0 - x_lo, 0), pitch - 1), xb = min(max(x0 + 1 - x_lo, 0), pitch - 1);
                const int ra = min(max(y0 - y_lo, 0), nrows - 1) * pitch, rb = min(max(y0 + 1 - y_lo, 0), nrows - 1) * pitch;
                const float* gq = G + li * GS;
                float a = w00 * gq[ra + xa];
                a = fmaf(w01, gq[ra + xb], a);
                a = fmaf(w10, gq[rb + xa], a);
                a = fmaf(w11, gq[rb + xb], a);
                sv = a * (1.0f / 32.0f);
            } else {
                // direct form (rare): blend the four taps, then the product with the reference features, all channels in this lane
                const int xa = min(max(x0, 0), w - 1), xb = min(max(x0 + 1, 0), w - 1);
                const int ya = min(max(y0, 0), h - 1) * w, yb = min(max(y0 + 1, 0), h - 1) * w;
                const effi_gptr4 t0 = sb4 + (long)(ya + xa) * (C / 4), t1 = sb4 + (long)(ya + xb) * (C / 4);
                const effi_gptr4 t2 = sb4 + (long)(yb + xa) * (C / 4), t3 = sb4 + (long)(yb + xb) * (C / 4);
                const effi_f4* rq = reinterpret_cast<const effi_f4*>(ref + (long)pix * C);
                float a = 0.0f;
#pragma unroll 2
                for (int q = 0; q < C / 4; ++q) {
                    const effi_f4 v = w00 * t0[q] + w01 * t1[q] + w10 * t2[q] + w11 * t3[q];
                    const effi_f4 rr = rq[q];
                    a = fmaf(v.x, rr.x, a); a = fmaf(v.y, rr.y, a); a = fmaf(v.z, rr.z, a); a = fmaf(v.w, rr.w, a);
                }
                sv = a * (1.0f / 32.0f);
            }
            if (d < D) {
                if (valid) simv[(long)d * hw] = sv;
                m = fmaxf(m, sv);
            }
        }
        if (fits) {                                              // G is rewritten by the next chunk
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        d0 += 4 * ng;
    }
    // ---- softmax over D and entropy (models/Effi_MVS_plus.py:43-44); lane slot lk owns (and wrote) d = lk, lk + 4, ...
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float z = 0.0f;
    for (int d = lk; d < D; d += 4) z = z + expf((valid ? simv[(long)d * hw] : 0.0f) - m);
    z = z + __shfl_xor(z, 16);
    z = z + __shfl_xor(z, 32);
    float e = 0.0f;
    for (int d = lk; d < D; d += 4) {
        const float p = expf((valid ? simv[(long)d * hw] : 0.0f) - m) / z;
        e = e + (-p) * logf(p + 1e-7f);
    }
    e = e + __shfl_xor(e, 16);
    e = e + __shfl_xor(e, 32);
    if (valid && lk == 0) entropy[(long)view * hw + pix] = e;
}

template <bool TBL, bool HI_ONLY>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void warpcorr_views_mm_kernel(const float* __restrict__ ref_arg, EffiPtrList srcs,
                                                                const float* __restrict__ rt_all, const float* __restrict__ depth,
                                                                long dds, int h, int w, int D, float* sim_views,
                                                                float* __restrict__ entropy) {
    const int view = blockIdx.y;
    warpcorr_views_mm_body<HI_ONLY>(TBL ? srcs.tbl[0] : ref_arg, TBL ? srcs.tbl[1 + view] : pick_view_list(srcs, view), rt_all, depth, dds,
                                    h, w, D, sim_views, entropy);
}

// ------------------------------------------------------------------------------------------------
// Backward of the stage-1 warp + correlation with the scatter PRIVATISED in LDS (scope row n2): same tiles, chunks and windows as
// warpcorr_views_win_kernel.  grad_src[tap][c] += g * w_tap * ref[c] is accumulated with LDS atomics (ds_add_f32) into a window of
// the source-gradient map and flushed to global memory ONCE per (tile, chunk) -- 576 x 32 global atomics instead of
// 128 pixels x 20 hypotheses x 4 taps x 32 channels, an 18-fold reduction of what bound the direct kernel (0.73 G global atomic adds
// = 8.6 ms at 148x200, D = 48, S = 4, on par with PyTorch-ROCm's grid_sample backward).  grad_ref[p][c] = sum g * warp(src)[c] reads
// its taps from global memory (L1 path) and is added atomically per view (grid.y = view; zero on entry).  A chunk whose window
// does not fit falls back to global atomics for the scatter.
// ------------------------------------------------------------------------------------------------
struct WinTapsB {
    float w[4];
    int a[4];     // index of the tap pixel in the LDS window (or -1: fallback to global atomics)
    int g[4];     // pixel index y*W + x in the source map
};

template <int MAXPX>
__global__ __launch_bounds__(WIN_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void warpcorr_views_bwd_win_kernel(
    const float* __restrict__ ref, EffiPtrList srcs, const float* __restrict__ rt_all, const float* __restrict__ depth, long dds, int h,
    int w, int D, const float* __restrict__ grad_sim, float* __restrict__ grad_ref, EffiOutList grad_srcs, int lds_px) {
    constexpr int C = 32, MAXD = 256;
    // the gradient window, CHANNEL-PLANAR [32][npad] with npad = 2 (mod 8): the 64 lanes of a ds_add_f32 (16 neighbouring pixels x the
    // 4 lanes of a pixel, channels 8 apart) hit 64 different banks.  Measured: the layout does not matter (pixel-major [pixel][32],
    // where 8 of the 16 pixels share banks, runs in the same 4.0 ms) -- the kernel is bound by the LDS atomic unit itself, about
    // 180 clocks per ds_add_f32 wave instruction; without the adds the whole kernel takes 0.25 ms, without the flush or without the
    // global tap reads it is unchanged (tools/prof_warp_bwd.py)
    __shared__ float4 win4[(MAXPX > 0 ? MAXPX : 1) * 8 + 64];
    __shared__ float hyp[MAXD + 4];
    __shared__ int wpar[2][8];
    float* winf = reinterpret_cast<float*>(win4);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tiles_x = (w + WIN_TW - 1) / WIN_TW;
    const int tl = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tyi = tl / tiles_x, txi = tl - tyi * tiles_x;
    const int gidx = tid >> 2, sub = tid & 3;
    const int x = txi * WIN_TW + (gidx & (WIN_TW - 1)), y = tyi * WIN_TH + (gidx >> 4);
    const bool valid = (x < w) & (y < h);
    const int xs = min(x, w - 1), ys = min(y, h - 1);
    const int view = blockIdx.y;
    const float* __restrict__ src = pick_view(srcs, view);
    float* gs = grad_srcs.p[0];
#pragma unroll
    for (int i = 1; i <= EFFI_MAX_VIEWS; ++i)
        if (view == i) gs = grad_srcs.p[i];
    const float* __restrict__ rt = rt_all + view * 12;
    const int hw = h * w, pix = ys * w + xs;
    for (int d = tid; d < ((D + 3) & ~3); d += WIN_THREADS) hyp[d] = depth[(long)min(d, D - 1) * dds];
    const float4 rlo = *reinterpret_cast<const float4*>(ref + (long)pix * C + sub * 8);
    const float4 rhi = *reinterpret_cast<const float4*>(ref + (long)pix * C + sub * 8 + 4);
    const float rr[8] = {rlo.x, rlo.y, rlo.z, rlo.w, rhi.x, rhi.y, rhi.z, rhi.w};
    WinProj P;
    {
        const float fx = (float)xs, fy = (float)ys;
        P.rx = rt[0] * fx + rt[1] * fy + rt[2];
        P.ry = rt[3] * fx + rt[4] * fy + rt[5];
        P.rz = rt[6] * fx + rt[7] * fy + rt[8];
        P.tx = rt[9]; P.ty = rt[10]; P.tz = rt[11];
        P.wm1 = (float)(w - 1); P.hm1 = (float)(h - 1);
        P.hw2 = P.wm1 / 2.0f; P.hh2 = P.hm1 / 2.0f;
        P.rhw2 = 1.0f / P.hw2; P.rhh2 = 1.0f / P.hh2;
    }
    const float* gsim = grad_sim + (long)view * D * hw + pix;
    float gr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) gr[c] = 0.0f;
    __syncthreads();
    const bool mono = win_hyp_monotone(hyp, D, tid);
    if (wv == 0) win_choose_chunk(rt, hyp, P, txi, tyi, 0, D, w, h, lds_px, lane, wpar[0], mono);
    __syncthreads();
    int da = 0, pb = 0;
    while (da < D) {
        const int x_lo = wpar[pb][0], y_lo = wpar[pb][1], ww = wpar[pb][2], wh = wpar[pb][3], db = wpar[pb][4], use_lds = wpar[pb][5];
        const int npx = ww * wh;
        const int npad = ((npx + 5) & ~7) + 2;             // >= npx, = 2 (mod 8); 32 * npad <= 32 * MAXPX + 256 floats
        if (use_lds) {
            for (int i = tid; i < 8 * npad; i += WIN_THREADS) win4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            __syncthreads();
        }
        if (wv == 0 && db < D) win_choose_chunk(rt, hyp, P, txi, tyi, db, D, w, h, lds_px, lane, wpar[pb ^ 1], mono);
        for (int d0 = da; d0 < db; d0 += 4) {
            // every lane sets up all four hypotheses of the group itself (no set-up crosses lanes: note at the top of this file); the
            // set-up is ~3 % of this kernel, which is bound by the LDS atomic unit
#define EFFI_ONE(J)                                                                                                     \
            {                                                                                                           \
                const float dep = hyp[d0 + J];                                                                          \
                float ix, iy;                                                                                           \
                project_xy(P.rx * dep + P.tx, P.ry * dep + P.ty, P.rz * dep + P.tz, P.hw2, P.rhw2, P.hh2, P.rhh2, P.wm1, P.hm1, ix, iy); \
                WinTaps tw, tg;                                                                                         \
                make_taps_win<true>(ix, iy, w, h, x_lo, y_lo, ww, wh, tw);                                              \
                make_taps_win<false>(ix, iy, w, h, 0, 0, w, h, tg);                                                     \
                WinTapsB t;                                                                                             \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) { t.w[k] = tw.w[k]; t.a[k] = use_lds ? (tw.a[k] >> 7) : -1; t.g[k] = tg.a[k]; } \
                const float g = (valid && d0 + J < D) ? gsim[(long)(d0 + J) * hw] * (1.0f / 32.0f) : 0.0f;     /* x 1/C */ \
                _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                         \
                    const float gw = g * t.w[k];                                                                        \
                    if (gw == 0.0f) continue;              /* out-of-bounds taps, masked pixels: nothing to add */      \
                    const float4* q = reinterpret_cast<const float4*>(src + (long)t.g[k] * 32 + sub * 8);               \
                    const float4 s0 = q[0], s1 = q[1];                                                                  \
                    gr[0] = fmaf(gw, s0.x, gr[0]); gr[1] = fmaf(gw, s0.y, gr[1]); gr[2] = fmaf(gw, s0.z, gr[2]); gr[3] = fmaf(gw, s0.w, gr[3]); \
                    gr[4] = fmaf(gw, s1.x, gr[4]); gr[5] = fmaf(gw, s1.y, gr[5]); gr[6] = fmaf(gw, s1.z, gr[6]); gr[7] = fmaf(gw, s1.w, gr[7]); \
                    if (t.a[k] >= 0) {                                                                                  \
                        float* wp_ = winf + (sub * 8) * npad + t.a[k];                                                   \
                        _Pragma("unroll") for (int c = 0; c < 8; ++c)                                                    \
                            __hip_atomic_fetch_add(wp_ + c * npad, gw * rr[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     \
                    } else {                                                                                             \
                        float* o = gs + (long)t.g[k] * 32 + sub * 8;                                                     \
                        _Pragma("unroll") for (int c = 0; c < 8; ++c) unsafeAtomicAdd(o + c, gw * rr[c]);                \
                    }                                                                                                    \
                }                                                                                                       \
            }
            EFFI_ONE(0) EFFI_ONE(1) EFFI_ONE(2) EFFI_ONE(3)
#undef EFFI_ONE
        }
        __syncthreads();                   // every add of this chunk has landed in the window
        if (use_lds) {
            // flush: item i = (window pixel Pw, channel quad q); a wave covers 16 pixels x 4 quads -> 16-byte runs in the nhwc map
            for (int i = tid; i < npx * 8; i += WIN_THREADS) {
                const int Pw = i >> 3, q = i & 7;
                const float* wp_ = winf + (q * 4) * npad + Pw;
                const float v0 = wp_[0], v1 = wp_[npad], v2 = wp_[2 * npad], v3 = wp_[3 * npad];
                if (v0 == 0.0f && v1 == 0.0f && v2 == 0.0f && v3 == 0.0f) continue;
                const int wy = Pw / ww, wx = Pw - wy * ww;
                float* o = gs + ((long)(y_lo + wy) * w + x_lo + wx) * 32 + q * 4;
                unsafeAtomicAdd(o + 0, v0);
                unsafeAtomicAdd(o + 1, v1);
                unsafeAtomicAdd(o + 2, v2);
                unsafeAtomicAdd(o + 3, v3);
            }
            __syncthreads();               // before the next chunk clears the window
        }
        da = db;
        pb ^= 1;
    }
    if (!valid) return;
    float* o = grad_ref + (long)pix * C + sub * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) unsafeAtomicAdd(o + c, gr[c]);
}

// ------------------------------------------------------------------------------------------------
// stages 2/3: hypotheses around the current depth, all views, view-weighted aggregate
// ------------------------------------------------------------------------------------------------
template <int C, bool NODPP = true, bool SHFL = false, bool FAST = false>
__global__ __launch_bounds__(256) void warpcorr_dyn_kernel(const float* ref_arg, EffiPtrList srcs, int S,
                                                           const float* __restrict__ rt_all,
                                                           const float* __restrict__ cur_depth,
                                                           const float* __restrict__ interval,
                                                           const float* __restrict__ view_w, int vw_shift,
                                                           int h, int w, int D, float* __restrict__ sim,
                                                           float* __restrict__ samples) {
    using G = WarpGeom<C>;
    const float* __restrict__ ref = effi_resolve_views(ref_arg, srcs);
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int hw = h * w, pix = y * w + x, sub4 = 4 * sub;
    const float4 r4 = *reinterpret_cast<const float4*>(ref + (long)pix * C + sub4);
    const float fx = (float)x, fy = (float)y;
    // get_cur_depth_range_samples in inverse depth (models/module.py:554-570, Effi_MVS_plus.py:194-207)
    const float inv = 1.0f / cur_depth[pix];
    const float half = (float)(D / 2) * interval[0];
    const float smin = fmaxf(inv - half, 1e-4f);
    const float smax = fminf(fmaxf(inv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(D - 1);
    // view weights, nearest-upsampled from the stage-1 map (Effi_MVS_plus.py:497)
    const int vh = h >> vw_shift, vw = w >> vw_shift;
    const int vpix = (y >> vw_shift) * vw + (x >> vw_shift);
    float wsum = 0.0f;
    for (int v = 0; v < S; ++v) wsum = wsum + view_w[(long)v * vh * vw + vpix];
    const float den = wsum + 1e-6f;
    constexpr int GS = (G::LPP >= 4) ? 4 : 2;
#ifdef EFFI_DIAG_LANE_EXCHANGE
    const int gj = threadIdx.x % GS;
#endif
    // FAST set-up (the default, see effi_warpcorr_dyn_f32): the kernel is bound by vector-ALU issue -- a hypothesis' set-up is ~130
    // instructions with the four IEEE divisions of the projection, and without a lane exchange every lane sets up every hypothesis.
    // This form uses the stage-1 window kernel's projection (project_xy: refined reciprocal + one residual step per division,
    // correctly rounded except for rare 1-ulp cases of a continuous function) and its tap arithmetic (make_taps_win<false>), and
    // addresses the taps as 32-bit byte offsets from the view's (uniform) base pointer.
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const float hw2 = wm1 / 2.0f, hh2 = hm1 / 2.0f;
    const float rhw2 = 1.0f / hw2, rhh2 = 1.0f / hh2;              // IEEE divisions: correctly rounded reciprocals
    for (int d0 = 0; d0 < D; d0 += GS) {
#ifdef EFFI_DIAG_LANE_EXCHANGE
        const int dm = min(d0 + gj, D - 1);
        const float my_dep = 1.0f / fmaxf(smin + (float)dm * step, 1e-5f);
#endif
        float acc[GS], dep[GS];
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            acc[j] = 0.0f;
            dep[j] = 1.0f / fmaxf(smin + (float)min(d0 + j, D - 1) * step, 1e-5f);
        }
        for (int v = 0; v < S; ++v) {
            const float* __restrict__ src = pick_view(srcs, v);
            const float* __restrict__ rt = rt_all + v * 12;
            const float rx = rt[0] * fx + rt[1] * fy + rt[2];
            const float ry = rt[3] * fx + rt[4] * fy + rt[5];
            const float rz = rt[6] * fx + rt[7] * fy + rt[8];
#ifdef EFFI_DIAG_LANE_EXCHANGE
            Taps mine;
#endif
            Taps t;
            const float wv = view_w[(long)v * vh * vw + vpix];
            if (NODPP && !SHFL && FAST) {
                const char* __restrict__ sb = reinterpret_cast<const char*>(src) + sub4 * 4;
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    float ix, iy;
                    project_xy(rx * dep[j] + rt[9], ry * dep[j] + rt[10], rz * dep[j] + rt[11], hw2, rhw2, hh2, rhh2, wm1, hm1, ix, iy);
                    WinTaps tw;
                    make_taps_win<false>(ix, iy, w, h, 0, 0, w, h, tw);
                    const float4 a = *reinterpret_cast<const float4*>(sb + (unsigned)(tw.a[0] * (C * 4)));
                    const float4 b = *reinterpret_cast<const float4*>(sb + (unsigned)(tw.a[1] * (C * 4)));
                    const float4 c = *reinterpret_cast<const float4*>(sb + (unsigned)(tw.a[2] * (C * 4)));
                    const float4 e = *reinterpret_cast<const float4*>(sb + (unsigned)(tw.a[3] * (C * 4)));
                    float sd = tw.w[0] * dot4(a, r4);
                    sd = fmaf(tw.w[1], dot4(b, r4), sd);
                    sd = fmaf(tw.w[2], dot4(c, r4), sd);
                    sd = fmaf(tw.w[3], dot4(e, r4), sd);
                    acc[j] = fmaf(wv, sd, acc[j]);
                }
                continue;
            }
#ifdef EFFI_DIAG_LANE_EXCHANGE
            if (NODPP && SHFL) {
                // A/B form (EFFI_DYN_XCHG=shfl): each lane of a group sets up ONE hypothesis and the taps go round through ds_bpermute.
                // Fails next to concurrent replays exactly like the DPP form below
                make_taps(rx * my_dep + rt[9], ry * my_dep + rt[10], rz * my_dep + rt[11], w, h, C, mine);
                const int lane0 = (int)(threadIdx.x & 63) & ~(GS - 1);
#pragma unroll
                for (int j = 0; j < GS; ++j) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        t.w[k] = __shfl(mine.w[k], lane0 + j);
                        t.off[k] = __shfl(mine.off[k], lane0 + j);
                    }
                    acc[j] = fmaf(wv, sample_dot(src, t, sub4, r4), acc[j]);
                }
                continue;
            }
#endif
            if (NODPP) {
                // NO cross-lane exchange (the default): every lane sets up all GS hypotheses itself.  The exchange form below (each lane
                // of a group sets up ONE hypothesis, the taps go round by quad_perm DPP moves; 1.5 % faster per view) gives wrong
                // similarities for single 16-lane rows of single iterations -- one to five replays in a hundred -- as soon as kernels
                // of OTHER hipGraph replays run on the GPU at the same time (three views in flight), never when a pass runs alone;
                // see DESIGN.md section 6 and tools/diag_in_flight*.py.  EFFI_DYN_XCHG=dpp selects it for A/B runs.  The same happens when
                // the taps go round through ds_bpermute instead (EFFI_DYN_XCHG=shfl): it is the exchange of the set-up between lanes
                // that fails, not one instruction.
#pragma unroll
                for (int j = 0; j < GS; ++j) {
                    make_taps(rx * dep[j] + rt[9], ry * dep[j] + rt[10], rz * dep[j] + rt[11], w, h, C, t);
                    acc[j] = fmaf(wv, sample_dot(src, t, sub4, r4), acc[j]);
                }
                continue;
            }
#ifdef EFFI_DIAG_LANE_EXCHANGE
            make_taps(rx * my_dep + rt[9], ry * my_dep + rt[10], rz * my_dep + rt[11], w, h, C, mine);
            taps_bcast<GS, 0>(mine, t);
            acc[0] = fmaf(wv, sample_dot(src, t, sub4, r4), acc[0]);
            taps_bcast<GS, 1>(mine, t);
            acc[1] = fmaf(wv, sample_dot(src, t, sub4, r4), acc[1]);
            if (GS == 4) {
                taps_bcast<GS, 2>(mine, t);
                acc[2] = fmaf(wv, sample_dot(src, t, sub4, r4), acc[2]);
                taps_bcast<GS, 3>(mine, t);
                acc[3] = fmaf(wv, sample_dot(src, t, sub4, r4), acc[3]);
            }
#endif
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int d = d0 + j;
            float gsum;
            if (NODPP) {
                gsum = acc[j];
#pragma unroll
                for (int o = 1; o < G::LPP; o <<= 1) gsum += __shfl_xor(gsum, o);
            } else {
                gsum = effi_group_sum<G::LPP>(acc[j]);
            }
            const float total = (gsum / (float)C) / den;
            // every lane recomputes the hypothesis value of d (two ops) so that the owner lane can store it
            const float dep_d = 1.0f / fmaxf(smin + (float)min(d, D - 1) * step, 1e-5f);
            if (d < D && (d % G::LPP) == sub) {
                sim[(long)d * hw + pix] = total;
                samples[(long)d * hw + pix] = dep_d;
            }
        }
    }
}

// sum_c tap[c] * ref[c] over C channels as TWO interleaved chains (even / odd channels) of packed FMAs (v_pk_fma_f32: C/2 instructions
// + one add instead of C); used by the hypothesis-per-lane gather kernel and by the LDS-window kernel, which therefore agree bit for bit
template <int Q>
__device__ __forceinline__ float dyn_dot(const effi_f4 (&tv)[Q], const float4 (&r)[Q]) {
    f32x2 a = f32x2{tv[0].x, tv[0].y} * f32x2{r[0].x, r[0].y};
    a = __builtin_elementwise_fma(f32x2{tv[0].z, tv[0].w}, f32x2{r[0].z, r[0].w}, a);
#pragma unroll
    for (int q = 1; q < Q; ++q) {
        a = __builtin_elementwise_fma(f32x2{tv[q].x, tv[q].y}, f32x2{r[q].x, r[q].y}, a);
        a = __builtin_elementwise_fma(f32x2{tv[q].z, tv[q].w}, f32x2{r[q].z, r[q].w}, a);
    }
    return a.x + a.y;
}

// Hypothesis-per-lane form of the same operator (the default for C = 8 / 16): the C/4 lanes of a pixel no longer split the channels of every
// hypothesis -- which makes every lane set up every hypothesis once the set-up cannot be exchanged between lanes -- but the
// hypotheses: lane `sub` owns d = sub, sub + C/4, ... with ALL C channels of the pixel (the reference features sit in C registers),
// so a pixel's D x S set-ups are computed exactly once, and nothing crosses lanes (no exchange, no reduction).  Same FAST set-up
// arithmetic as above; the channel sum runs over C in one lane, as two chains of packed FMAs (dyn_dot; a different association than 4 + butterfly: ~1e-7 relative).
template <int C>
__global__ __launch_bounds__(256) void warpcorr_dyn_hyp_kernel(const float* ref_arg, EffiPtrList srcs, int S,
                                                               const float* __restrict__ rt_all,
                                                               const float* __restrict__ cur_depth,
                                                               const float* __restrict__ interval,
                                                               const float* __restrict__ view_w, int vw_shift,
                                                               int h, int w, int D, float* __restrict__ sim,
                                                               float* __restrict__ samples) {
    using G = WarpGeom<C>;
    const float* __restrict__ ref = effi_resolve_views(ref_arg, srcs);
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int hw = h * w, pix = y * w + x;
    float4 r[C / 4];
#pragma unroll
    for (int q = 0; q < C / 4; ++q) r[q] = *reinterpret_cast<const float4*>(ref + (long)pix * C + 4 * q);
    const float fx = (float)x, fy = (float)y;
    const float inv = 1.0f / cur_depth[pix];
    const float half = (float)(D / 2) * interval[0];
    const float smin = fmaxf(inv - half, 1e-4f);
    const float smax = fminf(fmaxf(inv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(D - 1);
    const int vh = h >> vw_shift, vw = w >> vw_shift;
    const int vpix = (y >> vw_shift) * vw + (x >> vw_shift);
    float wsum = 0.0f;
    for (int v = 0; v < S; ++v) wsum = wsum + view_w[(long)v * vh * vw + vpix];
    const float den = wsum + 1e-6f;
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const float hw2 = wm1 / 2.0f, hh2 = hm1 / 2.0f;
    const float rhw2 = 1.0f / hw2, rhh2 = 1.0f / hh2;
    for (int d = sub; d < D; d += G::LPP) {
        const float dep = 1.0f / fmaxf(smin + (float)d * step, 1e-5f);
        float acc = 0.0f;
        for (int v = 0; v < S; ++v) {                 // (forced unroll factors 1 / 2 / 4 measured: none better than the compiler's choice)
            const char* __restrict__ sb = reinterpret_cast<const char*>(pick_view(srcs, v));
            const float* __restrict__ rt = rt_all + v * 12;
            const float rx = rt[0] * fx + rt[1] * fy + rt[2];
            const float ry = rt[3] * fx + rt[4] * fy + rt[5];
            const float rz = rt[6] * fx + rt[7] * fy + rt[8];
            const float wv = view_w[(long)v * vh * vw + vpix];
            float ix, iy;
            project_xy(rx * dep + rt[9], ry * dep + rt[10], rz * dep + rt[11], hw2, rhw2, hh2, rhh2, wm1, hm1, ix, iy);
            WinTaps tw;
            make_taps_win<false>(ix, iy, w, h, 0, 0, w, h, tw);
            effi_f4 tv[4][C / 4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const char* tp = sb + (unsigned)(tw.a[k] * (C * 4));
#pragma unroll
                for (int q = 0; q < C / 4; ++q) tv[k][q] = *reinterpret_cast<const effi_f4*>(tp + 16 * q);
            }
            float sd = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dk = dyn_dot<C / 4>(tv[k], r);
                sd = (k == 0) ? tw.w[0] * dk : fmaf(tw.w[k], dk, sd);
            }
            acc = fmaf(wv, sd, acc);
        }
        sim[(long)d * hw + pix] = (acc / (float)C) / den;
        samples[(long)d * hw + pix] = dep;
    }
}

// LDS-WINDOW form of the hypothesis-per-lane kernel (round 4; the default for C = 8 / 16, D <= 8).  The kernel above is bound by the
// texture path: 1.9 M wave-level 16-byte gathers at 592x800 (tools/microbench/ta_gather.hip: 80 us for the access shape alone),
// although the source positions of a tile of reference pixels are confined to a small box -- median 24 x 10 source pixels for the
// 16 x 8 tile's 128 pixels x 8 hypotheses (tools/probe_dyn_windows.py on the benchmark rig, whose depth maps are noisy).  Here, per
// source view:
//   * every lane sets up its hypotheses (projection as above) and the workgroup reduces the bounding box of the tap pixels: per-lane
//     min / max, packed as two 16-bit pairs, a wave butterfly (RESULTS cross lanes, never set-ups), one LDS store per wave; every
//     thread combines the four waves' boxes after the barrier (no atomics);
//   * the box (clipped to the image; channel-last rows are contiguous runs) is copied into LDS by global_load_lds_dwordx4 -- an image
//     of 16 / 12 rows with a fixed pitch of 32 pixels (C = 8: 1 KB = one wave-instruction per row; tap addresses are shifts) -- and
//     the taps are read with ds_read_b128 (128 B/clk/CU, no tag look-ups) instead of through the L1;
//   * a view whose box exceeds the image (wider than 32 pixels or taller than the image's rows: a depth discontinuity inside the tile,
//     extreme geometry) is sampled from global memory by the same code: identical arithmetic, bit for bit (option dyn_win = 0 forces
//     it for every view: the cross-check of the tests).
// Software pipeline over the views, ONE barrier per view: set-ups + box of view v + 1, barrier, the window of v + 1 requested (LDS-DMA:
// no registers), view v sampled from its image.  Same operations in the same order as warpcorr_dyn_hyp_kernel: bitwise the same
// similarities.  Instruction count per wave and view (C = 8): ~600 as the gather kernel, which it replaces the L1 path of.
struct DynTaps {
    float w[4];
    unsigned a[4];     // byte offset of the tap pixel's C channels: inside the LDS image / inside the source map
};
constexpr int DYN_WIN_PX = 32;         // pixels per image row (C = 8: 64 float4 = one wave-instruction of the copy; C = 16: 128 = two)

// MODE 0: taps in the source map; 1: in the LDS image of a window that was clipped at the image border; 2: in the LDS image of an
// INTERIOR window (the view's whole box lies inside the image: every tap is inside the image and inside the window -- no validity
// tests, no clamps, and the four tap addresses are ONE register + immediate offsets).  Same weights in every mode, bit for bit.
template <int MODE, int C>
__device__ __forceinline__ void make_taps_dyn(float ix, float iy, int W, int H, int x_lo, int y_lo, int ww, int wh, DynTaps& t) {
    // ix, iy: already clamped to [-2, W + 1] x [-2, H + 1] (make_taps_win's neutral clamp, applied by the caller)
    const float x0f = floorf(ix), y0f = floorf(iy);
    const int x0 = (int)x0f, y0 = (int)y0f;
    if (MODE == 2) {
        const float wx0 = (x0f + 1.0f) - ix, wx1 = ix - x0f, wy0 = (y0f + 1.0f) - iy, wy1 = iy - y0f;
        t.w[0] = wx0 * wy0;
        t.w[1] = wx1 * wy0;
        t.w[2] = wx0 * wy1;
        t.w[3] = wx1 * wy1;
        t.a[0] = (unsigned)(y0 - y_lo) * (DYN_WIN_PX * C * 4) + (unsigned)(x0 - x_lo) * (C * 4);
        t.a[1] = t.a[0] + C * 4;
        t.a[2] = t.a[0] + DYN_WIN_PX * C * 4;
        t.a[3] = t.a[0] + DYN_WIN_PX * C * 4 + C * 4;
        return;
    }
    const float wx0 = ((unsigned)x0 < (unsigned)W) ? (x0f + 1.0f) - ix : 0.0f;
    const float wx1 = ((unsigned)(x0 + 1) < (unsigned)W) ? ix - x0f : 0.0f;
    const float wy0 = ((unsigned)y0 < (unsigned)H) ? (y0f + 1.0f) - iy : 0.0f;
    const float wy1 = ((unsigned)(y0 + 1) < (unsigned)H) ? iy - y0f : 0.0f;
    t.w[0] = wx0 * wy0;
    t.w[1] = wx1 * wy0;
    t.w[2] = wx0 * wy1;
    t.w[3] = wx1 * wy1;
    // window-relative and clamped INTO the window (MODE 1: x_lo, y_lo, ww, wh describe it; MODE 0: the whole map): a no-op for every
    // tap that carries weight, and zero-weight taps read finite staged data
    const int xa = min(max(x0 - x_lo, 0), ww - 1), xb = min(max(x0 + 1 - x_lo, 0), ww - 1);
    const int ra = min(max(y0 - y_lo, 0), wh - 1), rb = min(max(y0 + 1 - y_lo, 0), wh - 1);
    if (MODE == 1) {
        const unsigned ya = (unsigned)ra * (DYN_WIN_PX * C * 4), yb = (unsigned)rb * (DYN_WIN_PX * C * 4);
        t.a[0] = ya + (unsigned)xa * (C * 4);
        t.a[1] = ya + (unsigned)xb * (C * 4);
        t.a[2] = yb + (unsigned)xa * (C * 4);
        t.a[3] = yb + (unsigned)xb * (C * 4);
    } else {
        const int ya = __mul24(ra, ww), yb = __mul24(rb, ww);
        t.a[0] = (unsigned)(ya + xa) * (C * 4);
        t.a[1] = (unsigned)(ya + xb) * (C * 4);
        t.a[2] = (unsigned)(yb + xa) * (C * 4);
        t.a[3] = (unsigned)(yb + xb) * (C * 4);
    }
}


// one hypothesis of one view: sum_c ref[c] * bilinear(src[c]) with the taps from the LDS window (LDSWIN) or from the source map
template <int MODE, int C>
__device__ __forceinline__ float dyn_sample_hyp(const char* __restrict__ wb, effi_gptr4 sb, float ix, float iy, int w, int h,
                                                int x_lo, int y_lo, int ww, int wh, const float4 (&r)[C / 4]) {
    constexpr int Q = C / 4;
    DynTaps t;
    constexpr bool LDSWIN = MODE != 0;
    make_taps_dyn<MODE, C>(ix, iy, w, h, x_lo, y_lo, ww, wh, t);
    // taps in groups of KG (C = 16: two at a time -- 32 instead of 64 registers of tap data in flight; the other waves hide the latency)
    constexpr int KG = (Q > 2) ? 2 : 4;
    float sd = 0.0f;
#pragma unroll
    for (int k0 = 0; k0 < 4; k0 += KG) {
        effi_f4 tv[KG][Q];
#pragma unroll
        for (int k = 0; k < KG; ++k)
#pragma unroll
            for (int q = 0; q < Q; ++q)
                tv[k][q] = LDSWIN ? *reinterpret_cast<const effi_f4*>(wb + t.a[k0 + k] + 16 * q) : sb[(t.a[k0 + k] >> 4) + q];
#pragma unroll
        for (int k = 0; k < KG; ++k) {
            const float dk = dyn_dot<Q>(tv[k], r);
            sd = (k0 + k == 0) ? t.w[0] * dk : fmaf(t.w[k0 + k], dk, sd);
        }
        if (KG < 4) __builtin_amdgcn_sched_barrier(0);
    }
    return sd;
}

template <int C, int HPL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C == 8 ? 4 : 3, C == 8 ? 4 : 3))) void warpcorr_dyn_win_kernel(
    const float* ref_arg, EffiPtrList srcs, int S, const float* __restrict__ rt_all, const float* __restrict__ cur_depth,
    const float* __restrict__ interval, const float* __restrict__ view_w, int vw_shift, int h, int w, int D, float* __restrict__ sim,
    float* __restrict__ samples, int lds_px) {
    using G = WarpGeom<C>;
    constexpr int LPP = G::LPP, Q = C / 4;
    constexpr int PITCH = DYN_WIN_PX * Q, PARTS = PITCH / 64;   // float4 per image row; wave-instructions of the copy per row
    constexpr int ROWS = (C == 8) ? 16 : 12;                     // 16 / 24 KB per image, two images: four / three workgroups per CU (as the registers allow)
    constexpr int MAXW = DYN_WIN_PX;
    __shared__ effi_f4 win[2][ROWS * PITCH];
    __shared__ int bbw[EFFI_MAX_VIEWS][4][2];                // per view and wave: packed (min x, min y), (max x, max y) of the tap pixels
    const float* __restrict__ ref = effi_resolve_views(ref_arg, srcs);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvi = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave index as a SCALAR: window rows / LDS bases of the copy are then scalar arithmetic
    const int tiles_x = (w + G::TW - 1) / G::TW;
    const int tl = effi_xcd_remap(blockIdx.x, gridDim.x);
    const int tyi = tl / tiles_x, txi = tl - tyi * tiles_x;
    const int g = tid / LPP, sub = tid % LPP;
    const int x = txi * G::TW + (g % G::TW), y = tyi * G::TH + (g / G::TW);
    const bool valid = (x < w) & (y < h);
    const int xs = min(x, w - 1), ys = min(y, h - 1);          // out-of-image lanes shadow the border pixel (no stores)
    const int hw = h * w, pix = ys * w + xs;
    float4 r[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) r[q] = *reinterpret_cast<const float4*>(ref + (long)pix * C + 4 * q);
    const float fx = (float)xs, fy = (float)ys;
    const float inv = 1.0f / cur_depth[pix];
    const float half = (float)(D / 2) * interval[0];
    const float smin = fmaxf(inv - half, 1e-4f);
    const float smax = fminf(fmaxf(inv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(D - 1);
    const int vh = h >> vw_shift, vw = w >> vw_shift;
    const int vpix = (ys >> vw_shift) * vw + (xs >> vw_shift);
    float wsum = 0.0f;
    for (int v = 0; v < S; ++v) wsum = wsum + view_w[(long)v * vh * vw + vpix];
    const float den = wsum + 1e-6f;
    const float wm1 = (float)(w - 1), hm1 = (float)(h - 1);
    const float hw2 = wm1 / 2.0f, hh2 = hm1 / 2.0f;
    const float rhw2 = 1.0f / hw2, rhh2 = 1.0f / hh2;
    float dep[HPL], acc[HPL];
#pragma unroll
    for (int j = 0; j < HPL; ++j) {                             // lane `sub` owns d = sub, sub + LPP, ... (beyond D: a shadow of D - 1)
        dep[j] = 1.0f / fmaxf(smin + (float)min(sub + j * LPP, D - 1) * step, 1e-5f);
        acc[j] = 0.0f;
    }

    // set-ups of one view for this lane's hypotheses + the wave's share of the view's bounding box
    auto setup = [&](int v, float (&ix)[HPL], float (&iy)[HPL]) {
        const float* __restrict__ rt = rt_all + v * 12;
        const float rx = rt[0] * fx + rt[1] * fy + rt[2];
        const float ry = rt[3] * fx + rt[4] * fy + rt[5];
        const float rz = rt[6] * fx + rt[7] * fy + rt[8];
        float mnx = 0.0f, mxx = 0.0f, mny = 0.0f, mxy = 0.0f;
#pragma unroll
        for (int j = 0; j < HPL; ++j) {
            project_xy(rx * dep[j] + rt[9], ry * dep[j] + rt[10], rz * dep[j] + rt[11], hw2, rhw2, hh2, rhh2, wm1, hm1, ix[j], iy[j]);
            ix[j] = fminf(fmaxf(ix[j], -2.0f), (float)w + 1.0f);       // make_taps_win's neutral clamp; NaN -> outside
            iy[j] = fminf(fmaxf(iy[j], -2.0f), (float)h + 1.0f);
            mnx = j ? fminf(mnx, ix[j]) : ix[j]; mxx = j ? fmaxf(mxx, ix[j]) : ix[j];
            mny = j ? fminf(mny, iy[j]) : iy[j]; mxy = j ? fmaxf(mxy, iy[j]) : iy[j];
        }
        // tap pixels are floor(.) and floor(.) + 1; coordinates lie in [-2, 32767) (the launch checks the map size): 16-bit pairs
        effi_s16x2 lo = {(short)(int)floorf(mnx), (short)(int)floorf(mny)};
        effi_s16x2 hi = {(short)((int)floorf(mxx) + 1), (short)((int)floorf(mxy) + 1)};
        // inside a row of 16 lanes: four DPP moves (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: full-rate, no LDS round
        // trip); across the four rows: two shuffles
#define EFFI_BOX_STEP(MOVE)                                                                                              \
        {                                                                                                                \
            const int lo_i = __builtin_bit_cast(int, lo), hi_i = __builtin_bit_cast(int, hi);                           \
            const int lo_o = MOVE(lo_i), hi_o = MOVE(hi_i);                                                              \
            lo = __builtin_elementwise_min(lo, __builtin_bit_cast(effi_s16x2, lo_o));                                    \
            hi = __builtin_elementwise_max(hi, __builtin_bit_cast(effi_s16x2, hi_o));                                    \
        }
#define EFFI_DPP_B1(v) __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true)
#define EFFI_DPP_4E(v) __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true)
#define EFFI_DPP_141(v) __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true)
#define EFFI_DPP_140(v) __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true)
#define EFFI_SHFL_16(v) __shfl_xor(v, 16)
#define EFFI_SHFL_32(v) __shfl_xor(v, 32)
        EFFI_BOX_STEP(EFFI_DPP_B1)
        EFFI_BOX_STEP(EFFI_DPP_4E)
        EFFI_BOX_STEP(EFFI_DPP_141)
        EFFI_BOX_STEP(EFFI_DPP_140)
        EFFI_BOX_STEP(EFFI_SHFL_16)
        EFFI_BOX_STEP(EFFI_SHFL_32)
#undef EFFI_BOX_STEP
#undef EFFI_DPP_B1
#undef EFFI_DPP_4E
#undef EFFI_DPP_141
#undef EFFI_DPP_140
#undef EFFI_SHFL_16
#undef EFFI_SHFL_32
        if (lane == 0) {
            bbw[v][wvi][0] = __builtin_bit_cast(int, lo);
            bbw[v][wvi][1] = __builtin_bit_cast(int, hi);
        }
    };
    // the window of a view (uniform): the box clipped to the image, or ww = 0 when it does not fit the LDS image
    struct Win { int x_lo, y_lo, ww, wh, interior; };
    auto window_of = [&](int v) {
        effi_s16x2 lo = __builtin_bit_cast(effi_s16x2, bbw[v][0][0]), hi = __builtin_bit_cast(effi_s16x2, bbw[v][0][1]);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            lo = __builtin_elementwise_min(lo, __builtin_bit_cast(effi_s16x2, bbw[v][k][0]));
            hi = __builtin_elementwise_max(hi, __builtin_bit_cast(effi_s16x2, bbw[v][k][1]));
        }
        const int lo_u = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lo));
        const int hi_u = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hi));
        Win W_;
        const int x_lo = min(max((int)(short)(lo_u & 0xffff), 0), w - 1);
        const int x_hi = max(min((int)(short)(hi_u & 0xffff), w - 1), x_lo);
        const int y_lo = min(max(lo_u >> 16, 0), h - 1);
        const int y_hi = max(min(hi_u >> 16, h - 1), y_lo);
        W_.x_lo = x_lo; W_.y_lo = y_lo; W_.ww = x_hi - x_lo + 1; W_.wh = y_hi - y_lo + 1;
        // interior: the unclipped box of the tap pixels lies inside the image (every tap is a real pixel of the window)
        W_.interior = ((int)(short)(lo_u & 0xffff) >= 0) & ((lo_u >> 16) >= 0) & ((int)(short)(hi_u & 0xffff) <= w - 1) & ((hi_u >> 16) <= h - 1);
        if (W_.ww > MAXW || W_.wh > ROWS || W_.ww * W_.wh > lds_px) W_.ww = 0;
        return W_;
    };
    // window copy: global -> LDS directly (global_load_lds_dwordx4: no registers, no ds_write pass).  Wave k copies rows k, k + 4, ...:
    // one wave-instruction per row -- lane i reads float4 min(i, ww Q - 1) of the row and lands in slot i of the row's 64.  The copy
    // is drained by the next barrier (vmcnt(0)); no ordinary vector load is consumed while it is in flight (the view's weight and base
    // pointer are fetched with the set-ups).
    auto stage = [&](const float* __restrict__ src, int buf, const Win& W_) {
        if (W_.ww == 0) return;
        const effi_gptr4 gsrc = (effi_gptr4)(src + ((long)W_.y_lo * w + W_.x_lo) * C);
        const int n4 = W_.ww * Q;
#pragma unroll
        for (int j = 0; j < (ROWS * PARTS + 3) / 4; ++j) {
            const int item = wvi + 4 * j, row = item / PARTS, part = item % PARTS;     // (scalar arithmetic)
            if (row >= W_.wh) break;                             // wave-uniform
            if (part * 64 >= n4) continue;
            __builtin_amdgcn_global_load_lds(gsrc + ((long)row * w * Q + min(part * 64 + lane, n4 - 1)),
                                             (__attribute__((address_space(3))) effi_f4*)(&win[buf][row * PITCH + part * 64]), 16, 0, 0);
        }
    };
    auto sample = [&](const float* __restrict__ src, float wv, int buf, const Win& W_, const float (&ix)[HPL], const float (&iy)[HPL]) {
        const char* __restrict__ wb = reinterpret_cast<const char*>(win[buf]);
        const effi_gptr4 sb = (effi_gptr4)src;
        // one hypothesis at a time (sched_barrier: the scheduler otherwise hoists the tap reads of all HPL hypotheses to the top --
        // 128 registers of tap data at C = 8); the uniform window / global choice is the OUTER branch
        if (W_.ww != 0 && W_.interior) {
#pragma unroll
            for (int j = 0; j < HPL; ++j) {
                acc[j] = fmaf(wv, dyn_sample_hyp<2, C>(wb, sb, ix[j], iy[j], w, h, W_.x_lo, W_.y_lo, W_.ww, W_.wh, r), acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (W_.ww != 0) {
#pragma unroll
            for (int j = 0; j < HPL; ++j) {
                acc[j] = fmaf(wv, dyn_sample_hyp<1, C>(wb, sb, ix[j], iy[j], w, h, W_.x_lo, W_.y_lo, W_.ww, W_.wh, r), acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < HPL; ++j) {
                acc[j] = fmaf(wv, dyn_sample_hyp<0, C>(wb, sb, ix[j], iy[j], w, h, 0, 0, w, h, r), acc[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    float ixc[HPL], iyc[HPL], ixn[HPL], iyn[HPL];
    setup(0, ixc, iyc);
    const float* srcc = pick_view(srcs, 0);
    const float* srcn = srcc;
    float wvc = view_w[vpix], wvn = wvc;
    __syncthreads();                                             // the waves' boxes of view 0
    Win wc = window_of(0), wn = wc;
    stage(srcc, 0, wc);
    for (int v = 0; v < S; ++v) {
        const bool more = v + 1 < S;
        if (more) {
            setup(v + 1, ixn, iyn);
            srcn = pick_view(srcs, v + 1);
            wvn = view_w[(long)(v + 1) * vh * vw + vpix];
        }
        // This wave's share of window v must have LANDED before the barrier lets other waves read it.  The compiler's fence in front
        // of the barrier does not wait for the LDS-DMA by itself (seen in the ISA: lgkmcnt(0) only, when no ordinary load happened to
        // be outstanding): explicit vmcnt(0).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();             // window v has landed, the boxes of v + 1 are stored, nobody samples image (v + 1) & 1 any more
        if (more) {
            wn = window_of(v + 1);
            stage(srcn, (v + 1) & 1, wn);
        }
        sample(srcc, wvc, v & 1, wc, ixc, iyc);
        if (more) {
#pragma unroll
            for (int j = 0; j < HPL; ++j) { ixc[j] = ixn[j]; iyc[j] = iyn[j]; }
            wc = wn; srcc = srcn; wvc = wvn;
        }
    }
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < HPL; ++j) {
        const int d = sub + j * LPP;
        if (d < D) {
            sim[(long)d * hw + pix] = (acc[j] / (float)C) / den;
            samples[(long)d * hw + pix] = dep[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// full warped volume (API parity for homo_warping_new only; not on the fused path)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void homo_warp_kernel(const float* __restrict__ src, const float* __restrict__ rt,
                                                        const float* __restrict__ depth, long dds, long dps,
                                                        int h, int w, int D, float* __restrict__ out) {
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int hw = h * w, pix = y * w + x, sub4 = 4 * sub;
    const float fx = (float)x, fy = (float)y;
    const float rx = rt[0] * fx + rt[1] * fy + rt[2];
    const float ry = rt[3] * fx + rt[4] * fy + rt[5];
    const float rz = rt[6] * fx + rt[7] * fy + rt[8];
    const float* dp = depth + (long)pix * dps;
    for (int d = 0; d < D; ++d) {
        const float dep = dp[d * dds];
        Taps t;
        make_taps(rx * dep + rt[9], ry * dep + rt[10], rz * dep + rt[11], w, h, C, t);
        const float4 a = *reinterpret_cast<const float4*>(src + t.off[0] + sub4);
        const float4 b = *reinterpret_cast<const float4*>(src + t.off[1] + sub4);
        const float4 c = *reinterpret_cast<const float4*>(src + t.off[2] + sub4);
        const float4 e = *reinterpret_cast<const float4*>(src + t.off[3] + sub4);
        float4 r;
        r.x = a.x * t.w[0] + b.x * t.w[1] + c.x * t.w[2] + e.x * t.w[3];
        r.y = a.y * t.w[0] + b.y * t.w[1] + c.y * t.w[2] + e.y * t.w[3];
        r.z = a.z * t.w[0] + b.z * t.w[1] + c.z * t.w[2] + e.z * t.w[3];
        r.w = a.w * t.w[0] + b.w * t.w[1] + c.w * t.w[2] + e.w * t.w[3];
        float* o = out + ((long)sub4 * D + d) * hw + pix;
        o[0] = r.x;
        o[(long)D * hw] = r.y;
        o[2L * D * hw] = r.z;
        o[3L * D * hw] = r.w;
    }
}

// ------------------------------------------------------------------------------------------------
// Scope row n2 (first piece): backward of the stage-1 warp + correlation,  sim[v][d][p] = mean_c ref[p][c] * warp_v(src_v)[c][d][p]
// (models/module.py:303-344 + models/Effi_MVS_plus.py:38-40; the sampling grid carries no gradient, module.py:313).
//   grad_ref[p][c]       = sum_{v,d} g * sum_t w_t * src_v[tap_t][c]            (plain store: a lane group owns its pixel)
//   grad_src_v[tap_t][c] += g * w_t * ref[p][c]                                  (scatter: fp32 atomic adds)
// with g = grad_sim[v][d][p] / C.  Same lane layout as the forward kernel (C/4 lanes per pixel, 4 channels per lane), taps
// recomputed from the projection, so nothing but grad_sim is read beyond the forward's inputs.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void warpcorr_views_bwd_kernel(const float* __restrict__ ref, EffiPtrList srcs, int S,
                                                                 const float* __restrict__ rt_all,
                                                                 const float* __restrict__ depth, long dds, long dps,
                                                                 int h, int w, int D, const float* __restrict__ grad_sim,
                                                                 float* __restrict__ grad_ref, EffiOutList grad_srcs) {
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int hw = h * w, pix = y * w + x, sub4 = 4 * sub;
    const float4 r4 = *reinterpret_cast<const float4*>(ref + (long)pix * C + sub4);
    const float fx = (float)x, fy = (float)y;
    const float* dp = depth + (long)pix * dps;
    float4 gr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int v = 0; v < S; ++v) {
        const float* __restrict__ src = pick_view(srcs, v);
        float* gs = grad_srcs.p[0];
#pragma unroll
        for (int i = 1; i <= EFFI_MAX_VIEWS; ++i)
            if (v == i) gs = grad_srcs.p[i];
        const float* __restrict__ rt = rt_all + v * 12;
        const float rx = rt[0] * fx + rt[1] * fy + rt[2];
        const float ry = rt[3] * fx + rt[4] * fy + rt[5];
        const float rz = rt[6] * fx + rt[7] * fy + rt[8];
        for (int d = 0; d < D; ++d) {
            const float dep = dp[d * dds];
            Taps t;
            make_taps(rx * dep + rt[9], ry * dep + rt[10], rz * dep + rt[11], w, h, C, t);
            const float g = grad_sim[((long)v * D + d) * hw + pix] / (float)C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (t.w[k] == 0.0f) continue;                        // out-of-bounds taps carry no gradient
                const float4 sv = *reinterpret_cast<const float4*>(src + t.off[k] + sub4);
                const float gw = g * t.w[k];
                gr.x = fmaf(gw, sv.x, gr.x);
                gr.y = fmaf(gw, sv.y, gr.y);
                gr.z = fmaf(gw, sv.z, gr.z);
                gr.w = fmaf(gw, sv.w, gr.w);
                float* o = gs + t.off[k] + sub4;
                unsafeAtomicAdd(o + 0, gw * r4.x);
                unsafeAtomicAdd(o + 1, gw * r4.y);
                unsafeAtomicAdd(o + 2, gw * r4.z);
                unsafeAtomicAdd(o + 3, gw * r4.w);
            }
        }
    }
    *reinterpret_cast<float4*>(grad_ref + (long)pix * C + sub4) = gr;
}

// Backward of homo_warp_kernel w.r.t. the source features (scope row n2): grad_src[tap][c] += w_tap * grad_out[c][d][p]; the grid
// is constant (models/module.py:313).  Same lane layout as the forward kernel; fp32 atomic adds.
template <int C>
__global__ __launch_bounds__(256) void homo_warp_bwd_kernel(const float* __restrict__ rt, const float* __restrict__ depth,
                                                            long dds, long dps, int h, int w, int D,
                                                            const float* __restrict__ grad_out, float* __restrict__ grad_src) {
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int hw = h * w, pix = y * w + x, sub4 = 4 * sub;
    const float fx = (float)x, fy = (float)y;
    const float rx = rt[0] * fx + rt[1] * fy + rt[2];
    const float ry = rt[3] * fx + rt[4] * fy + rt[5];
    const float rz = rt[6] * fx + rt[7] * fy + rt[8];
    const float* dp = depth + (long)pix * dps;
    for (int d = 0; d < D; ++d) {
        const float dep = dp[d * dds];
        Taps t;
        make_taps(rx * dep + rt[9], ry * dep + rt[10], rz * dep + rt[11], w, h, C, t);
        const float* g = grad_out + ((long)sub4 * D + d) * hw + pix;
        const float g0 = g[0], g1 = g[(long)D * hw], g2 = g[2L * D * hw], g3 = g[3L * D * hw];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (t.w[k] == 0.0f) continue;
            float* o = grad_src + t.off[k] + sub4;
            unsafeAtomicAdd(o + 0, t.w[k] * g0);
            unsafeAtomicAdd(o + 1, t.w[k] * g1);
            unsafeAtomicAdd(o + 2, t.w[k] * g2);
            unsafeAtomicAdd(o + 3, t.w[k] * g3);
        }
    }
}

// Backward of warpcorr_dyn_kernel (scope row n2; models/Effi_MVS_plus.py:184-251 under autograd): with s_vd = mean_c ref_c *
// warp_v(src_v)_c at hypothesis d and sim_d = sum_v w_v s_vd / den, den = sum_v w_v + 1e-6, and g_d the incoming gradient:
//   grad ref_c        = sum_{d,v} g_d w_v / (den C) * warp_vdc                       (plain store: the lane group owns its pixel)
//   grad src_v[tap]_c += g_d w_v / (den C) * w_tap * ref_c                           (scatter, fp32 atomics; zero on entry)
//   grad w_v (coarse) += sum_d g_d (s_vd - sim_d) / den                              (atomics into the 1/2^k-resolution map)
// The hypotheses come from the detached current depth (models/Effi_MVS_plus.py:495): no gradient there.
template <int C>
__global__ __launch_bounds__(256) void warpcorr_dyn_bwd_kernel(const float* __restrict__ ref, EffiPtrList srcs, int S,
                                                               const float* __restrict__ rt_all, const float* __restrict__ cur_depth,
                                                               const float* __restrict__ interval, const float* __restrict__ view_w,
                                                               int vw_shift, int h, int w, int D, const float* __restrict__ sim,
                                                               const float* __restrict__ gsim, float* __restrict__ grad_ref,
                                                               EffiOutList grad_srcs, double* __restrict__ grad_vw) {
    using G = WarpGeom<C>;
    int x, y, sub;
    if (!tile_pixel<C>(blockIdx.x, gridDim.x, h, w, x, y, sub)) return;
    const int hw = h * w, pix = y * w + x, sub4 = 4 * sub;
    const float4 r4 = *reinterpret_cast<const float4*>(ref + (long)pix * C + sub4);
    const float fx = (float)x, fy = (float)y;
    const float inv = 1.0f / cur_depth[pix];
    const float half = (float)(D / 2) * interval[0];
    const float smin = fmaxf(inv - half, 1e-4f);
    const float smax = fminf(fmaxf(inv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(D - 1);
    const int vh = h >> vw_shift, vw = w >> vw_shift;
    const int vpix = (y >> vw_shift) * vw + (x >> vw_shift);
    float wsum = 0.0f;
    for (int v = 0; v < S; ++v) wsum = wsum + view_w[(long)v * vh * vw + vpix];
    const float den = wsum + 1e-6f;
    float4 gr = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int v = 0; v < S; ++v) {
        const float* __restrict__ src = pick_view(srcs, v);
        float* gs = grad_srcs.p[0];
#pragma unroll
        for (int i = 1; i <= EFFI_MAX_VIEWS; ++i)
            if (v == i) gs = grad_srcs.p[i];
        const float* __restrict__ rt = rt_all + v * 12;
        const float rx = rt[0] * fx + rt[1] * fy + rt[2];
        const float ry = rt[3] * fx + rt[4] * fy + rt[5];
        const float rz = rt[6] * fx + rt[7] * fy + rt[8];
        const float wv = view_w[(long)v * vh * vw + vpix];
        // gradient of the view weight: a sum over hypotheses (here) and over the 4 / 16 fine pixels of a weight cell (atomics below) of
        // differences (s_vd - sim_d) that mostly cancel; accumulated in DOUBLE, atomics included, so that the order of the adds does
        // not show in fp32 (in fp32 the view-weight net's gradients moved by 3e-3 of their peak between two runs of one step)
        double gwv = 0.0;
        for (int d = 0; d < D; ++d) {
            const float dep = 1.0f / fmaxf(smin + (float)d * step, 1e-5f);
            Taps t;
            make_taps(rx * dep + rt[9], ry * dep + rt[10], rz * dep + rt[11], w, h, C, t);
            float4 tv[4];
            float4 wp = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // warped source, this lane's 4 channels
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                tv[k] = *reinterpret_cast<const float4*>(src + t.off[k] + sub4);
                wp.x = fmaf(t.w[k], tv[k].x, wp.x);
                wp.y = fmaf(t.w[k], tv[k].y, wp.y);
                wp.z = fmaf(t.w[k], tv[k].z, wp.z);
                wp.w = fmaf(t.w[k], tv[k].w, wp.w);
            }
            const float s_vd = effi_group_sum<G::LPP>(dot4(wp, r4)) / (float)C;
            const float g = gsim[(long)d * hw + pix];
            const float coef = g * wv / (den * (float)C);
            gr.x = fmaf(coef, wp.x, gr.x);
            gr.y = fmaf(coef, wp.y, gr.y);
            gr.z = fmaf(coef, wp.z, gr.z);
            gr.w = fmaf(coef, wp.w, gr.w);
            gwv += (double)g * ((double)s_vd - (double)sim[(long)d * hw + pix]) / (double)den;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (t.w[k] == 0.0f) continue;
                const float cw = coef * t.w[k];
                float* o = gs + t.off[k] + sub4;
                unsafeAtomicAdd(o + 0, cw * r4.x);
                unsafeAtomicAdd(o + 1, cw * r4.y);
                unsafeAtomicAdd(o + 2, cw * r4.z);
                unsafeAtomicAdd(o + 3, cw * r4.w);
            }
        }
        if (sub == 0) unsafeAtomicAdd(&grad_vw[(long)v * vh * vw + vpix], gwv);
    }
    *reinterpret_cast<float4*>(grad_ref + (long)pix * C + sub4) = gr;
}

template <int C> int grid_blocks(int h, int w) {
    using G = WarpGeom<C>;
    return ((w + G::TW - 1) / G::TW) * ((h + G::TH - 1) / G::TH);
}

bool fill_views(const float* const* src, int S, EffiPtrList& l) {
    l.tbl = nullptr;
    if (!src || S < 1 || S > EFFI_MAX_VIEWS) return false;
    for (int i = 0; i <= EFFI_MAX_VIEWS; ++i) l.p[i] = nullptr;
    for (int i = 0; i < S; ++i) {
        if (!src[i]) return false;
        l.p[i] = src[i];
    }
    return true;
}

}  // namespace

extern "C" int effi_homo_warp_f32(const float* src_nhwc, const float* rt, const float* depth, long dds, long dps,
                                  int C, int h, int w, int D, float* out, effi_stream_t stream) {
    if (!src_nhwc || !rt || !depth || !out || h < 2 || w < 2 || D < 1) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    switch (C) {
        case 32: hipLaunchKernelGGL(homo_warp_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, src_nhwc, rt, depth, dds, dps, h, w, D, out); break;
        case 16: hipLaunchKernelGGL(homo_warp_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, src_nhwc, rt, depth, dds, dps, h, w, D, out); break;
        case 8:  hipLaunchKernelGGL(homo_warp_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, src_nhwc, rt, depth, dds, dps, h, w, D, out); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_homo_warp_bwd_f32(const float* rt, const float* depth, long dds, long dps, int C, int h, int w, int D,
                                      const float* grad_out, float* grad_src_nhwc, effi_stream_t stream) {
    if (!rt || !depth || !grad_out || !grad_src_nhwc || h < 2 || w < 2 || D < 1) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    switch (C) {
        case 32: hipLaunchKernelGGL(homo_warp_bwd_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, rt, depth, dds, dps, h, w, D, grad_out, grad_src_nhwc); break;
        case 16: hipLaunchKernelGGL(homo_warp_bwd_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, rt, depth, dds, dps, h, w, D, grad_out, grad_src_nhwc); break;
        case 8:  hipLaunchKernelGGL(homo_warp_bwd_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, rt, depth, dds, dps, h, w, D, grad_out, grad_src_nhwc); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

// Window capacity of the stage-1 LDS kernel in pixels (128 B each): 72 KB leaves room for two 512-thread workgroups per CU.
constexpr int WIN_MAXPX = 576;

// Option warp_lds_kb (effi_set_option; tests and A/B runs only): unset = the windowed kernel with its full 72 KB window;
// 0 = the windowed kernel with every chunk sampled from global memory (same arithmetic, the bitwise cross-check);
// -1 = the direct-gather kernel (also what C != 32 and per-pixel hypotheses use).
static int warp_lds_px() {
    const long kb = effi_option(EFFI_OPT_WARP_LDS_KB);
    if (kb == EFFI_OPT_UNSET) return WIN_MAXPX;
    if (kb < 0) return -1;
    return min(WIN_MAXPX, (int)(kb * 1024 / 128));
}

// A view table (device memory, EFFI_MAX_VIEWS + 2 pointers: reference, sources, nulls) as the kernels' view list
static bool fill_table(const float* const* table_dev, int S, EffiPtrList& l) {
    if (!table_dev || S < 1 || S > EFFI_MAX_VIEWS) return false;
    for (int i = 0; i <= EFFI_MAX_VIEWS; ++i) l.p[i] = nullptr;
    l.tbl = table_dev;
    return true;
}

static int launch_warpcorr_views(const float* ref_nhwc, const EffiPtrList& l, int S, const float* rt, const float* depth, long dds,
                                 long dps, int C, int h, int w, int D, float* sim_views, float* entropy, effi_stream_t stream) {
    if (!rt || !depth || !sim_views || !entropy) return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 1) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    const int lds_px = warp_lds_px();
    if (C == 32 && dps == 0 && lds_px >= 0 && D <= 256) {
        // hypotheses shared by all pixels (the cascade's stage 1): taps served from an LDS window
        const int tiles = ((w + WIN_TW - 1) / WIN_TW) * ((h + WIN_TH - 1) / WIN_TH);
        if (l.tbl)
            hipLaunchKernelGGL((warpcorr_views_win_kernel<WIN_MAXPX, true>), dim3(tiles, S), dim3(WIN_THREADS), 0, s, ref_nhwc, l, rt, depth,
                               dds, h, w, D, sim_views, entropy, lds_px);
        else
            hipLaunchKernelGGL((warpcorr_views_win_kernel<WIN_MAXPX, false>), dim3(tiles, S), dim3(WIN_THREADS), 0, s, ref_nhwc, l, rt, depth,
                               dds, h, w, D, sim_views, entropy, lds_px);
        EFFI_LAUNCH_CHECK();
        return EFFI_OK;
    }
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_views_kernel<32>, dim3(grid_blocks<32>(h, w), S), dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, dps, h, w, D, sim_views, entropy); break;
        case 16: hipLaunchKernelGGL(warpcorr_views_kernel<16>, dim3(grid_blocks<16>(h, w), S), dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, dps, h, w, D, sim_views, entropy); break;
        case 8:  hipLaunchKernelGGL(warpcorr_views_kernel<8>, dim3(grid_blocks<8>(h, w), S), dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, dps, h, w, D, sim_views, entropy); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_warpcorr_views_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                       const float* depth, long dds, long dps, int C, int h, int w, int D,
                                       float* sim_views, float* entropy, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc) return EFFI_ERR_BADARG;
    return launch_warpcorr_views(ref_nhwc, l, S, rt, depth, dds, dps, C, h, w, D, sim_views, entropy, stream);
}

// Split-precision stage-1 form (warpcorr_views_mm_kernel): C = 32 and hypotheses shared by all pixels, else the exact kernels above.
// hi_only: bf16 operands (precision "bf16").
static int launch_warpcorr_views_x3(const float* ref_nhwc, const EffiPtrList& l, int S, const float* rt, const float* depth, long dds,
                                    long dps, int C, int h, int w, int D, float* sim_views, float* entropy, int hi_only,
                                    effi_stream_t stream) {
    if (!rt || !depth || !sim_views || !entropy) return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 1) return EFFI_ERR_BADARG;
    if (!(C == 32 && dps == 0 && D <= 256 && h < 32000 && w < 32000))
        return launch_warpcorr_views(ref_nhwc, l, S, rt, depth, dds, dps, C, h, w, D, sim_views, entropy, stream);
    hipStream_t s = effi_s(stream);
    const dim3 grid(((w + 15) / 16) * ((h + 3) / 4), S);
#define EFFI_MM(TBL, HO) hipLaunchKernelGGL((warpcorr_views_mm_kernel<TBL, HO>), grid, dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, h, w, D, sim_views, entropy)
    if (l.tbl) { if (hi_only) EFFI_MM(true, true); else EFFI_MM(true, false); }
    else { if (hi_only) EFFI_MM(false, true); else EFFI_MM(false, false); }
#undef EFFI_MM
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_warpcorr_views_x3_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                          const float* depth, long dds, long dps, int C, int h, int w, int D,
                                          float* sim_views, float* entropy, int hi_only, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc) return EFFI_ERR_BADARG;
    return launch_warpcorr_views_x3(ref_nhwc, l, S, rt, depth, dds, dps, C, h, w, D, sim_views, entropy, hi_only, stream);
}

extern "C" int effi_warpcorr_views_x3_tbl_f32(const float* const* view_table_dev, int S, const float* rt, const float* depth, long dds,
                                              long dps, int C, int h, int w, int D, float* sim_views, float* entropy, int hi_only,
                                              effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_table(view_table_dev, S, l)) return EFFI_ERR_BADARG;
    return launch_warpcorr_views_x3(nullptr, l, S, rt, depth, dds, dps, C, h, w, D, sim_views, entropy, hi_only, stream);
}

extern "C" int effi_warpcorr_views_tbl_f32(const float* const* view_table_dev, int S, const float* rt, const float* depth, long dds,
                                           long dps, int C, int h, int w, int D, float* sim_views, float* entropy,
                                           effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_table(view_table_dev, S, l)) return EFFI_ERR_BADARG;
    return launch_warpcorr_views(nullptr, l, S, rt, depth, dds, dps, C, h, w, D, sim_views, entropy, stream);
}

extern "C" int effi_warpcorr_views_bwd_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                           const float* depth, long dds, long dps, int C, int h, int w, int D,
                                           const float* grad_sim, float* grad_ref_nhwc, float* const* grad_src_nhwc,
                                           effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc || !rt || !depth || !grad_sim || !grad_ref_nhwc || !grad_src_nhwc) return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 1) return EFFI_ERR_BADARG;
    EffiOutList g;
    for (int i = 0; i <= EFFI_MAX_VIEWS; ++i) g.p[i] = (i < S) ? grad_src_nhwc[i] : nullptr;
    for (int i = 0; i < S; ++i)
        if (!g.p[i]) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    const int lds_px = warp_lds_px();
    if (C == 32 && dps == 0 && lds_px >= 0 && D <= 256) {
        // hypotheses shared by all pixels: scatter privatised in an LDS window; grad_ref is ACCUMULATED per view (zero on entry)
        const int tiles = ((w + WIN_TW - 1) / WIN_TW) * ((h + WIN_TH - 1) / WIN_TH);
        hipLaunchKernelGGL(warpcorr_views_bwd_win_kernel<WIN_MAXPX>, dim3(tiles, S), dim3(WIN_THREADS), 0, s, ref_nhwc, l, rt, depth, dds, h,
                           w, D, grad_sim, grad_ref_nhwc, g, lds_px);
        EFFI_LAUNCH_CHECK();
        return EFFI_OK;
    }
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_views_bwd_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, depth, dds, dps, h, w, D, grad_sim, grad_ref_nhwc, g); break;
        case 16: hipLaunchKernelGGL(warpcorr_views_bwd_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, depth, dds, dps, h, w, D, grad_sim, grad_ref_nhwc, g); break;
        case 8:  hipLaunchKernelGGL(warpcorr_views_bwd_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, depth, dds, dps, h, w, D, grad_sim, grad_ref_nhwc, g); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

static int launch_warpcorr_dyn(const float* ref_nhwc, const EffiPtrList& l, int S, const float* rt, const float* cur_depth,
                               const float* interval, const float* view_w, int vw_shift, int C, int h, int w, int D, float* sim,
                               float* samples, effi_stream_t stream) {
    if (!rt || !cur_depth || !interval || !view_w || !sim || !samples) return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 2 || vw_shift < 0 || vw_shift > 4) return EFFI_ERR_BADARG;
    if ((h >> vw_shift) << vw_shift != h || (w >> vw_shift) << vw_shift != w) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    // (An 8-channels-per-lane form of this kernel -- one lane per pixel at C = 8, no exchange, no reduction, cheaper projection --
    // was built and measured at 592x800: 121 us against 109 us; at 296x400, C = 16: 63 against 58.  These kernels are bound by the
    // number of distinct cache lines a wave-instruction touches in the L1 / texture path, not by instruction issue.)
#define EFFI_DYN(CC, ...) hipLaunchKernelGGL((warpcorr_dyn_kernel<CC, __VA_ARGS__>), dim3(grid_blocks<CC>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples)
#ifdef EFFI_DIAG_LANE_EXCHANGE
    // diagnostic builds only: the forms that pass set-ups between lanes (option dyn_xchg: 1 = quad_perm DPP moves, 2 = ds_bpermute)
    const long xchg = effi_option(EFFI_OPT_DYN_XCHG);
    if (xchg == 1 || xchg == 2) {
        const bool dpp = xchg == 1;
        switch (C) {
            case 32: if (dpp) EFFI_DYN(32, false, false); else EFFI_DYN(32, true, true); break;
            case 16: if (dpp) EFFI_DYN(16, false, false); else EFFI_DYN(16, true, true); break;
            case 8:  if (dpp) EFFI_DYN(8, false, false); else EFFI_DYN(8, true, true); break;
            default: return EFFI_ERR_UNSUPPORTED;
        }
        EFFI_LAUNCH_CHECK();
        return EFFI_OK;
    }
#endif
    // Default for C = 8 / 16 (stages 2 / 3): the hypothesis-per-lane form (91 vs 107 us at 592x800, 60 vs 75 us at 296x400);
    // option dyn_form = 1 selects the channel-split form below, dyn_setup_exact = 1 the reference's IEEE divisions (A/B runs, tests)
    const bool lanes = effi_option(EFFI_OPT_DYN_FORM) == 1;
    const bool exact = effi_option(EFFI_OPT_DYN_SETUP_EXACT) == 1;
    if (!exact && !lanes && (C == 8 || C == 16) && (long)h * w * C * 4 < (1L << 31)) {
        // LDS-window form (warpcorr_dyn_win_kernel): D <= 8 (a lane holds the set-ups of its 4 / 2 hypotheses for two views);
        // option dyn_win: -1 = the gather kernel below, 0 = the window kernel with every view sampled from global memory
        const long dw = effi_option(EFFI_OPT_DYN_WIN);
        if (dw != -1 && D <= 8 && S <= EFFI_MAX_VIEWS && h < 32000 && w < 32000) {
            // option value n > 0: windows of at most n pixels (tests: most boxes then overflow and mix with LDS-served views)
            const int lds_px = dw == EFFI_OPT_UNSET ? (1 << 20) : (int)max(dw, 0L);
            if (C == 8)
                hipLaunchKernelGGL((warpcorr_dyn_win_kernel<8, 4>), dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth,
                                   interval, view_w, vw_shift, h, w, D, sim, samples, lds_px);
            else
                hipLaunchKernelGGL((warpcorr_dyn_win_kernel<16, 2>), dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt,
                                   cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples, lds_px);
            EFFI_LAUNCH_CHECK();
            return EFFI_OK;
        }
        if (C == 8) hipLaunchKernelGGL(warpcorr_dyn_hyp_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples);
        else hipLaunchKernelGGL(warpcorr_dyn_hyp_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples);
        EFFI_LAUNCH_CHECK();
        return EFFI_OK;
    }
    // set-up arithmetic of the default (exchange-free) form: "fast" (see the kernel) unless EFFI_DYN_SETUP=exact asks for the
    // reference's IEEE divisions op for op, or the map is too large for 32-bit byte offsets
    if (!exact && (long)h * w * C * 4 < (1L << 31)) {
        switch (C) {
            case 32: EFFI_DYN(32, true, false, true); break;
            case 16: EFFI_DYN(16, true, false, true); break;
            case 8:  EFFI_DYN(8, true, false, true); break;
            default: return EFFI_ERR_UNSUPPORTED;
        }
        EFFI_LAUNCH_CHECK();
        return EFFI_OK;
    }
#undef EFFI_DYN
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_dyn_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples); break;
        case 16: hipLaunchKernelGGL(warpcorr_dyn_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples); break;
        case 8:  hipLaunchKernelGGL(warpcorr_dyn_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_warpcorr_dyn_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                     const float* cur_depth, const float* interval, const float* view_w, int vw_shift,
                                     int C, int h, int w, int D, float* sim, float* samples, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc) return EFFI_ERR_BADARG;
    return launch_warpcorr_dyn(ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, C, h, w, D, sim, samples, stream);
}

extern "C" int effi_warpcorr_dyn_tbl_f32(const float* const* view_table_dev, int S, const float* rt, const float* cur_depth,
                                         const float* interval, const float* view_w, int vw_shift, int C, int h, int w, int D,
                                         float* sim, float* samples, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_table(view_table_dev, S, l)) return EFFI_ERR_BADARG;
    return launch_warpcorr_dyn(nullptr, l, S, rt, cur_depth, interval, view_w, vw_shift, C, h, w, D, sim, samples, stream);
}

// effi_view_table_set: n pointers, by value, into a device table (stream-ordered: the replay that follows on the same stream reads them)
struct EffiTableArgs { const void* p[EFFI_VIEW_TABLE_MAX]; };
__global__ void view_table_set_kernel(const void** __restrict__ table, EffiTableArgs a, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const void* v = a.p[0];
#pragma unroll
        for (int k = 1; k < EFFI_VIEW_TABLE_MAX; ++k)
            if (i == k) v = a.p[k];
        table[i] = v;
    }
}
extern "C" int effi_view_table_set(const void** table_dev, const void* const* ptrs, int n, effi_stream_t stream) {
    if (!table_dev || !ptrs || n < 1 || n > EFFI_VIEW_TABLE_MAX) return EFFI_ERR_BADARG;
    EffiTableArgs a;
    for (int i = 0; i < EFFI_VIEW_TABLE_MAX; ++i) a.p[i] = (i < n) ? ptrs[i] : nullptr;
    hipStream_t s = effi_s(stream);
    hipLaunchKernelGGL(view_table_set_kernel, dim3(1), dim3(64), 0, s, table_dev, a, n);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_warpcorr_dyn_bwd_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                         const float* cur_depth, const float* interval, const float* view_w, int vw_shift, int C, int h,
                                         int w, int D, const float* sim, const float* grad_sim, float* grad_ref_nhwc,
                                         float* const* grad_src_nhwc, double* grad_view_w, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc || !rt || !cur_depth || !interval || !view_w || !sim || !grad_sim || !grad_ref_nhwc ||
        !grad_src_nhwc || !grad_view_w)
        return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 2 || vw_shift < 0 || vw_shift > 4) return EFFI_ERR_BADARG;
    if ((h >> vw_shift) << vw_shift != h || (w >> vw_shift) << vw_shift != w) return EFFI_ERR_BADARG;
    EffiOutList g;
    for (int i = 0; i <= EFFI_MAX_VIEWS; ++i) g.p[i] = (i < S) ? grad_src_nhwc[i] : nullptr;
    for (int i = 0; i < S; ++i)
        if (!g.p[i]) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_dyn_bwd_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, grad_sim, grad_ref_nhwc, g, grad_view_w); break;
        case 16: hipLaunchKernelGGL(warpcorr_dyn_bwd_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, grad_sim, grad_ref_nhwc, g, grad_view_w); break;
        case 8:  hipLaunchKernelGGL(warpcorr_dyn_bwd_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, grad_sim, grad_ref_nhwc, g, grad_view_w); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
