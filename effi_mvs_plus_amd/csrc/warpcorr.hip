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
// bilinear / zeros / align_corners=True); built with -ffp-contract=off, FMAs are explicit.
#include "common.hpp"

namespace {

template <int C> struct WarpGeom {
    static constexpr int LPP = C / 4;                 // lanes per pixel
    static constexpr int PIX = 256 / LPP;             // pixels per 256-thread block
    static constexpr int TW = (C == 32) ? 8 : 16;     // tile width  (pixels)
    static constexpr int TH = PIX / TW;               // tile height (4, 4, 8)
};

__device__ __forceinline__ const float* pick_view(const EffiPtrList& l, int v) {
    const float* p = l.p[0];
#pragma unroll
    for (int i = 1; i <= EFFI_MAX_VIEWS; ++i)
        if (v == i) p = l.p[i];
    return p;
}

struct Taps {
    float w[4];
    int off[4];   // element offsets of the 4 taps' pixel (already multiplied by C)
};

// Lanes of one pixel share the per-hypothesis set-up (projection, two divisions, bilinear weights): lane j of
// every aligned group of G = min(lanes-per-pixel, 4) lanes computes hypothesis d0 + j, and the 8 results are
// broadcast inside the quad with quad_perm DPP moves (full-rate VALU, no LDS).  For C = 32 the two quads of a
// pixel do this redundantly (2x instead of 8x), for C = 8 a quad holds two pixels and G = 2.
template <int G, int J>
__device__ __forceinline__ int quad_bcast_i(int v) {
    // source lane inside the quad for destination lanes 0..3
    constexpr int s0 = J, s1 = J, s2 = (G == 4) ? J : 2 + J, s3 = (G == 4) ? J : 2 + J;
    constexpr int ctrl = s0 | (s1 << 2) | (s2 << 4) | (s3 << 6);
    return __builtin_amdgcn_update_dpp(0, v, ctrl, 0xF, 0xF, true);
}
template <int G, int J>
__device__ __forceinline__ void taps_bcast(const Taps& mine, Taps& out) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out.w[k] = __int_as_float(quad_bcast_i<G, J>(__float_as_int(mine.w[k])));
        out.off[k] = quad_bcast_i<G, J>(mine.off[k]);
    }
}

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
__global__ __launch_bounds__(256) void warpcorr_views_kernel(const float* __restrict__ ref, EffiPtrList srcs,
                                                             const float* __restrict__ rt_all,
                                                             const float* __restrict__ depth, long dds, long dps,
                                                             int h, int w, int D, float* sim_views,
                                                             float* __restrict__ entropy) {
    using G = WarpGeom<C>;
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
    constexpr int GS = (G::LPP >= 4) ? 4 : 2;                    // lanes sharing the set-up
    const int gj = threadIdx.x % GS;
    for (int d0 = 0; d0 < D; d0 += GS) {
        // this lane's hypothesis (clamped: the tail group recomputes the last one and discards it)
        const int dm = min(d0 + gj, D - 1);
        const float dep = dp[dm * dds];
        Taps mine;
        make_taps(rx * dep + tx, ry * dep + ty, rz * dep + tz, w, h, C, mine);     // module.py:325-327
        auto one = [&](const Taps& t, int d) {
            const float s = effi_group_sum<G::LPP>(sample_dot(src, t, sub4, r4)) / (float)C;   // mean over C, :40
            if (d < D) {
                if ((d % G::LPP) == sub) simv[(long)d * hw] = s;
                m = fmaxf(m, s);
            }
        };
        Taps t;
        taps_bcast<GS, 0>(mine, t);
        one(t, d0);
        taps_bcast<GS, 1>(mine, t);
        one(t, d0 + 1);
        if (GS == 4) {
            taps_bcast<GS, 2>(mine, t);
            one(t, d0 + 2);
            taps_bcast<GS, 3>(mine, t);
            one(t, d0 + 3);
        }
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
// stages 2/3: hypotheses around the current depth, all views, view-weighted aggregate
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void warpcorr_dyn_kernel(const float* __restrict__ ref, EffiPtrList srcs, int S,
                                                           const float* __restrict__ rt_all,
                                                           const float* __restrict__ cur_depth,
                                                           const float* __restrict__ interval,
                                                           const float* __restrict__ view_w, int vw_shift,
                                                           int h, int w, int D, float* __restrict__ sim,
                                                           float* __restrict__ samples) {
    using G = WarpGeom<C>;
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
    const int gj = threadIdx.x % GS;
    for (int d0 = 0; d0 < D; d0 += GS) {
        const int dm = min(d0 + gj, D - 1);
        const float my_dep = 1.0f / fmaxf(smin + (float)dm * step, 1e-5f);
        float acc[GS];
#pragma unroll
        for (int j = 0; j < GS; ++j) acc[j] = 0.0f;
        for (int v = 0; v < S; ++v) {
            const float* __restrict__ src = pick_view(srcs, v);
            const float* __restrict__ rt = rt_all + v * 12;
            const float rx = rt[0] * fx + rt[1] * fy + rt[2];
            const float ry = rt[3] * fx + rt[4] * fy + rt[5];
            const float rz = rt[6] * fx + rt[7] * fy + rt[8];
            Taps mine, t;
            make_taps(rx * my_dep + rt[9], ry * my_dep + rt[10], rz * my_dep + rt[11], w, h, C, mine);
            const float wv = view_w[(long)v * vh * vw + vpix];
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
        }
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            const int d = d0 + j;
            const float total = (effi_group_sum<G::LPP>(acc[j]) / (float)C) / den;
            // every lane recomputes the hypothesis value of d (two ops) so that the owner lane can store it
            const float dep_d = 1.0f / fmaxf(smin + (float)min(d, D - 1) * step, 1e-5f);
            if (d < D && (d % G::LPP) == sub) {
                sim[(long)d * hw + pix] = total;
                samples[(long)d * hw + pix] = dep_d;
            }
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

template <int C> int grid_blocks(int h, int w) {
    using G = WarpGeom<C>;
    return ((w + G::TW - 1) / G::TW) * ((h + G::TH - 1) / G::TH);
}

bool fill_views(const float* const* src, int S, EffiPtrList& l) {
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

extern "C" int effi_warpcorr_views_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                       const float* depth, long dds, long dps, int C, int h, int w, int D,
                                       float* sim_views, float* entropy, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc || !rt || !depth || !sim_views || !entropy) return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 1) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_views_kernel<32>, dim3(grid_blocks<32>(h, w), S), dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, dps, h, w, D, sim_views, entropy); break;
        case 16: hipLaunchKernelGGL(warpcorr_views_kernel<16>, dim3(grid_blocks<16>(h, w), S), dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, dps, h, w, D, sim_views, entropy); break;
        case 8:  hipLaunchKernelGGL(warpcorr_views_kernel<8>, dim3(grid_blocks<8>(h, w), S), dim3(256), 0, s, ref_nhwc, l, rt, depth, dds, dps, h, w, D, sim_views, entropy); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
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
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_views_bwd_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, depth, dds, dps, h, w, D, grad_sim, grad_ref_nhwc, g); break;
        case 16: hipLaunchKernelGGL(warpcorr_views_bwd_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, depth, dds, dps, h, w, D, grad_sim, grad_ref_nhwc, g); break;
        case 8:  hipLaunchKernelGGL(warpcorr_views_bwd_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, depth, dds, dps, h, w, D, grad_sim, grad_ref_nhwc, g); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_warpcorr_dyn_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                     const float* cur_depth, const float* interval, const float* view_w, int vw_shift,
                                     int C, int h, int w, int D, float* sim, float* samples, effi_stream_t stream) {
    EffiPtrList l;
    if (!fill_views(src_nhwc, S, l) || !ref_nhwc || !rt || !cur_depth || !interval || !view_w || !sim || !samples)
        return EFFI_ERR_BADARG;
    if (h < 2 || w < 2 || D < 2 || vw_shift < 0 || vw_shift > 4) return EFFI_ERR_BADARG;
    if ((h >> vw_shift) << vw_shift != h || (w >> vw_shift) << vw_shift != w) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    switch (C) {
        case 32: hipLaunchKernelGGL(warpcorr_dyn_kernel<32>, dim3(grid_blocks<32>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples); break;
        case 16: hipLaunchKernelGGL(warpcorr_dyn_kernel<16>, dim3(grid_blocks<16>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples); break;
        case 8:  hipLaunchKernelGGL(warpcorr_dyn_kernel<8>, dim3(grid_blocks<8>(h, w)), dim3(256), 0, s, ref_nhwc, l, S, rt, cur_depth, interval, view_w, vw_shift, h, w, D, sim, samples); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
