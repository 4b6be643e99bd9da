// Per-pixel / element-wise kernels of the stage loop: layout change, hypothesis generation,
// view aggregation, softmax + soft-argmin + confidence, 1-D volume lookups, convex upsampling.
// All are HBM-bound streaming kernels: one thread per pixel, lanes along x (coalesced), 256-thread
// blocks.  Compiled with -ffp-contract=off so that the element-wise formulas round exactly like the
// reference's separate torch ops; fused multiply-adds are written explicitly where wanted.
#include <cstdint>
#include "common.hpp"

namespace {

constexpr int TPB = 256;

// ------------------------------------------------------------------------------------------------
// planar [C][HW] -> nhwc [HW][C] through an LDS tile of 64 pixels (coalesced on both sides)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void planar_to_nhwc_kernel(EffiPtrList srcs, EffiOutList dsts, int C, int HW) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [C][65]
    const float* __restrict__ src = srcs.p[blockIdx.y];
    float* __restrict__ dst = dsts.p[blockIdx.y];
    const long p0 = (long)blockIdx.x * 64;
    const int n = C * 64;
    for (int e = threadIdx.x; e < n; e += TPB) {
        const int c = e >> 6, p = e & 63;
        const long gp = p0 + p;
        tile[c * 65 + p] = (gp < HW) ? src[(long)c * HW + gp] : 0.0f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += TPB) {
        const int p = e / C, c = e - p * C;
        const long gp = p0 + p;
        if (gp < HW) dst[gp * C + c] = tile[c * 65 + p];
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void split_tanh_relu_kernel(const float* __restrict__ ctx, int hd, int cd, int hw,
                                       float* __restrict__ hidden, float* __restrict__ inp) {
    const long n = (long)(hd + cd) * hw;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const float v = ctx[i];
        if (i < (long)hd * hw) hidden[i] = tanhf(v);
        else inp[i - (long)hd * hw] = fmaxf(v, 0.0f);
    }
}

// hw % 4 == 0: four values per thread (float4 loads / stores; a quad never straddles the hidden / input boundary)
__global__ __launch_bounds__(TPB) void split_tanh_relu4_kernel(const float4* __restrict__ ctx, long nh4, long n4, float4* __restrict__ hidden,
                                                               float4* __restrict__ inp) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= n4) return;
    const float4 v = ctx[i];
    if (i < nh4) hidden[i] = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
    else inp[i - nh4] = make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
}

// the same for up to 4 context maps (all stages of the cascade) in ONE launch: blocks [first[k], first[k+1]) work on map k
struct SplitStages {
    const float* ctx[4];
    float* hidden[4];
    float* inp[4];
    long nh[4], n[4];          // elements of the hidden part / of the whole map
    int first[5];              // first block of each map
    int vec4;                  // all maps movable as float4
};
__global__ __launch_bounds__(TPB) void split_tanh_relu_stages_kernel(SplitStages a) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) k += ((int)blockIdx.x >= a.first[j]) ? 1 : 0;
    const float* __restrict__ ctx = a.ctx[0];
    float* __restrict__ hidden = a.hidden[0];
    float* __restrict__ inp = a.inp[0];
    long nh = a.nh[0], n = a.n[0];
    int b0 = a.first[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (k == j) { ctx = a.ctx[j]; hidden = a.hidden[j]; inp = a.inp[j]; nh = a.nh[j]; n = a.n[j]; b0 = a.first[j]; }
    const long i = 4 * ((long)(blockIdx.x - b0) * TPB + threadIdx.x);          // 4 consecutive values per thread
    if (a.vec4) {               // every map: element counts multiples of 4, pointers 16-byte aligned -> one quad never straddles
        if (i >= n) return;     // the hidden / input boundary and moves as one 16-byte access
        const float4 v = *reinterpret_cast<const float4*>(ctx + i);
        if (i < nh) *reinterpret_cast<float4*>(hidden + i) = make_float4(tanhf(v.x), tanhf(v.y), tanhf(v.z), tanhf(v.w));
        else *reinterpret_cast<float4*>(inp + (i - nh)) = make_float4(fmaxf(v.x, 0.0f), fmaxf(v.y, 0.0f), fmaxf(v.z, 0.0f), fmaxf(v.w, 0.0f));
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i + e >= n) break;
        const float v = ctx[i + e];
        if (i + e < nh) hidden[i + e] = tanhf(v);
        else inp[i + e - nh] = fmaxf(v, 0.0f);
    }
}

__global__ void depth_to_inv_kernel(const float* __restrict__ depth, const float* __restrict__ disp_range,
                                    int n_range, int n, float* __restrict__ inv) {
    const float lo = disp_range[0], hi = disp_range[n_range - 1];
    // depth_to_disp(depth, depth_min_, depth_max_) with depth_max_ = 1/lo, depth_min_ = 1/hi
    const float max_depth = 1.0f / lo, min_depth = 1.0f / hi;
    const float min_disp = 1.0f / max_depth, max_disp = 1.0f / min_depth;
    const float den = (max_disp - min_disp) + 1e-10f;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
        const float s = 1.0f / depth[i];
        inv[i] = (s - min_disp) / den;
    }
}

__global__ void stage1_hypotheses_kernel(const float* __restrict__ disp_range, int n_range, int D,
                                         float* __restrict__ depths, float* __restrict__ intervals) {
    const float lo = disp_range[0], hi = disp_range[n_range - 1];
    const float step = (hi - lo) / (float)(D - 1);               // models/module.py:578-583
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float s = lo + (float)d * step;
        depths[d] = 1.0f / s;                                    // models/Effi_MVS_plus.py:473-474
    }
    if (threadIdx.x < 3) {
        const float base = (hi - lo) / (float)n_range;           // models/Effi_MVS_plus.py:424
        const float ratio = (threadIdx.x == 0) ? 4.0f : (threadIdx.x == 1 ? 2.0f : 1.0f);   // :316
        intervals[threadIdx.x] = base * ratio;
    } else if (threadIdx.x == 3) {
        intervals[3] = 1.0f / hi;                                // depth_min_ (:414)
    } else if (threadIdx.x == 4) {
        intervals[4] = 1.0f / lo;                                // depth_max_ (:413)
    }
}

__global__ void upsample_nearest_kernel(const float* __restrict__ in, int C, int h, int w, int f,
                                        float* __restrict__ out) {
    const int W = w * f, H = h * f;
    const long n = (long)C * H * W;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const int x = (int)(i % W);
        const long t = i / W;
        const int y = (int)(t % H);
        const int c = (int)(t / H);
        out[i] = in[((long)c * h + y / f) * w + x / f];
    }
}

// ------------------------------------------------------------------------------------------------
// Tail of the depth head (models/update.py:15,21 + the update of BasicUpdateBlock.forward, :125-127): conv2 is a 3x3
// convolution with ONE output channel, so  conv2(hid)[p] = sum_tap s_tap[p + tap]  with  s_tap[q] = sum_c w2[c][tap] hid[q][c]
// -- nine 1x1 projections of the hidden map, which the producer of ``hid`` (conv1 + ReLU) applies in its epilogue
// (effi_conv2d_k3_k1_bf16x3_f32 with cout2 = 9): the hidden map itself (16-48 channels) never reaches HBM, this kernel reads the
// nine partial-sum planes.  Zero padding of conv2 = partial sums of pixels outside the map count as zero.
//   inv_new = inv + tanh(sum + b2);  depth = inv_to_depth(inv_new)          4 pixels along x per thread (w % 4 == 0)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void head_update_kernel(const float* __restrict__ s9, const float* __restrict__ bias2,
                                                          const float* __restrict__ inv, const float* __restrict__ disp_range,
                                                          int n_range, int h, int w, float* __restrict__ out_inv,
                                                          float* __restrict__ out_depth) {
    const int wq = w >> 2;
    const long q = (long)blockIdx.x * TPB + threadIdx.x;
    if (q >= (long)h * wq) return;
    const int y = (int)(q / wq), x = (int)(q - (long)y * wq) * 4;
    const long hw = (long)h * w;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if (yy < 0 || yy >= h) continue;
        const float* row = s9 + (long)yy * w + x;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* pl = row + (long)(ky * 3 + kx) * hw;
            const float4 c = *reinterpret_cast<const float4*>(pl);
            if (kx == 0) {
                const float l = (x > 0) ? pl[-1] : 0.0f;
                acc[0] = acc[0] + l; acc[1] = acc[1] + c.x; acc[2] = acc[2] + c.y; acc[3] = acc[3] + c.z;
            } else if (kx == 1) {
                acc[0] = acc[0] + c.x; acc[1] = acc[1] + c.y; acc[2] = acc[2] + c.z; acc[3] = acc[3] + c.w;
            } else {
                const float r = (x + 4 < w) ? pl[4] : 0.0f;
                acc[0] = acc[0] + c.y; acc[1] = acc[1] + c.z; acc[2] = acc[2] + c.w; acc[3] = acc[3] + r;
            }
        }
    }
    const float b = bias2[0], lo = disp_range[0], hi = disp_range[n_range - 1];
    const long pix = (long)y * w + x;
    const float4 iv = *reinterpret_cast<const float4*>(inv + pix);
    float4 o, d;
    o.x = iv.x + tanhf(acc[0] + b); o.y = iv.y + tanhf(acc[1] + b); o.z = iv.z + tanhf(acc[2] + b); o.w = iv.w + tanhf(acc[3] + b);
    d.x = effi_inv_to_depth(o.x, lo, hi); d.y = effi_inv_to_depth(o.y, lo, hi);
    d.z = effi_inv_to_depth(o.z, lo, hi); d.w = effi_inv_to_depth(o.w, lo, hi);
    *reinterpret_cast<float4*>(out_inv + pix) = o;
    *reinterpret_cast<float4*>(out_depth + pix) = d;
}

// ------------------------------------------------------------------------------------------------
// K3: sim = sum_v sim_v * w_v / (sum_v w_v + 1e-6)      (models/Effi_MVS_plus.py:48-53,67)
// ------------------------------------------------------------------------------------------------
__global__ void view_aggregate_kernel(const float* __restrict__ sim_views, const float* __restrict__ weights,
                                      int S, int D, int hw, float* __restrict__ out) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    const int d0 = blockIdx.y * 8;
    if (!weights) {                                     // pixel_wise_net = None (models/Effi_MVS_plus.py:55-58,70): sum / (N - 1)
        for (int d = d0; d < min(D, d0 + 8); ++d) {
            float acc = 0.0f;
            for (int v = 0; v < S; ++v) acc = acc + sim_views[((long)v * D + d) * hw + p];
            out[(long)d * hw + p] = acc / (float)S;
        }
        return;
    }
    float wv[EFFI_MAX_VIEWS];
    float wsum = 0.0f;
#pragma unroll
    for (int v = 0; v < EFFI_MAX_VIEWS; ++v) {          // constant trip count keeps wv[] in registers
        wv[v] = (v < S) ? weights[(long)v * hw + p] : 0.0f;
        if (v < S) wsum = wsum + wv[v];
    }
    const float den = wsum + 1e-6f;
    for (int d = d0; d < min(D, d0 + 8); ++d) {
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < EFFI_MAX_VIEWS; ++v)
            if (v < S) acc = acc + sim_views[((long)v * D + d) * hw + p] * wv[v];
        out[(long)d * hw + p] = acc / den;
    }
}

// ------------------------------------------------------------------------------------------------
// K7: softmax over D, depth regression, 4-window confidence (models/Effi_MVS_plus.py:79-88)
// ------------------------------------------------------------------------------------------------
__global__ void softmax_regress_conf_kernel(const float* __restrict__ logits, const float* __restrict__ depth,
                                            long dds, long dps, int D, int hw,
                                            float* __restrict__ out_depth, float* __restrict__ out_conf,
                                            const float* __restrict__ disp_range, int n_range, float* __restrict__ out_dinv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;     // launched with 64-thread blocks
    if (p >= hw) return;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, logits[(long)d * hw + p]);
    float sum = 0.0f;
    for (int d = 0; d < D; ++d) sum = sum + expf(logits[(long)d * hw + p] - m);
    float dep = 0.0f, idxf = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float pr = expf(logits[(long)d * hw + p] - m) / sum;
        dep = dep + pr * depth[d * dds + p * dps];
        idxf = idxf + pr * (float)d;
    }
    int idx = (int)idxf;                       // .long(): truncation
    idx = max(0, min(D - 1, idx));
    float s4 = 0.0f;                           // pad (1,2) + avg_pool3d(4) * 4 == sum over idx-1 .. idx+2
    for (int k = idx - 1; k <= idx + 2; ++k) {
        const float pr = (k >= 0 && k < D) ? expf(logits[(long)k * hw + p] - m) / sum : 0.0f;
        s4 = s4 + pr;
    }
    out_depth[p] = dep;
    out_conf[p] = 4.0f * (s4 / 4.0f);
    if (out_dinv) out_dinv[p] = effi_depth_to_inv(dep, disp_range[0], disp_range[n_range - 1]);   // models/Effi_MVS_plus.py:538
}

// Same arithmetic, same order, for the depth counts the cascade uses: the D logits of a pixel are loaded ONCE into registers
// (all loads in flight together) and every exponential is evaluated once; the generic kernel walks the volume three times with
// dependent expf / division chains (27 us at 48 x 148 x 200 for 5.7 MB).
template <int DT>
__global__ __launch_bounds__(64) void softmax_regress_conf_reg_kernel(const float* __restrict__ logits, const float* __restrict__ depth,
                                                                      long dds, long dps, int hw, float* __restrict__ out_depth,
                                                                      float* __restrict__ out_conf, const float* __restrict__ disp_range,
                                                                      int n_range, float* __restrict__ out_dinv) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= hw) return;
    float e[DT], dv[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) e[d] = logits[(long)d * hw + p];
#pragma unroll
    for (int d = 0; d < DT; ++d) dv[d] = depth[d * dds + p * dps];
    float m = -INFINITY;
#pragma unroll
    for (int d = 0; d < DT; ++d) m = fmaxf(m, e[d]);
    float sum = 0.0f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        e[d] = expf(e[d] - m);
        sum = sum + e[d];
    }
    float dep = 0.0f, idxf = 0.0f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        e[d] = e[d] / sum;
        dep = dep + e[d] * dv[d];
        idxf = idxf + e[d] * (float)d;
    }
    int idx = (int)idxf;
    idx = max(0, min(DT - 1, idx));
    float s4 = 0.0f;
#pragma unroll
    for (int d = 0; d < DT; ++d) {                 // sum over idx-1 .. idx+2 in ascending order, as the generic kernel does
        const bool in = (d >= idx - 1) & (d <= idx + 2);
        s4 = in ? s4 + e[d] : s4;
    }
    out_depth[p] = dep;
    out_conf[p] = 4.0f * (s4 / 4.0f);
    if (out_dinv) out_dinv[p] = effi_depth_to_inv(dep, disp_range[0], disp_range[n_range - 1]);   // models/Effi_MVS_plus.py:538
}

// ------------------------------------------------------------------------------------------------
// K8: 1-D lookup in a per-pixel vector (pro_bilinear_sampler, models/Effi_MVS_plus.py:102-134)
// ------------------------------------------------------------------------------------------------
// (lookup1d_index / lookup1d_fetch / effi_getcost_pixel: common.hpp -- shared with the generated-input convolution of conv2d_sr.hip)
__device__ __forceinline__ float lookup1d(const float* __restrict__ vol, long dstride, int Dp,
                                          float q_depth, float dmin, float dmax) {
    int x0;
    float w0, w1;
    lookup1d_index(Dp, q_depth, dmin, dmax, x0, w0, w1);
    return lookup1d_fetch(vol, dstride, Dp, x0, w0, w1);
}

// bilinear_sampler (models/Effi_MVS_plus.py:102-117): rows of an [N][C][1][W] image sampled at PIXEL x coordinates, zeros outside,
// align_corners=True; the row index is 0 whatever ygrid says because H == 1 (grid_sample un-normalises y to 0 * (H - 1) / 2).
// One thread per (n, query): the sampling position is shared by the C channels.
__global__ void bilinear_sampler1d_kernel(const float* __restrict__ img, int C, int W, const float* __restrict__ coords, long nq,
                                          long total, float* __restrict__ out, float* __restrict__ mask) {
    const long i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= total) return;
    const long n = i / nq, q = i - n * nq;
    const float xp = coords[2 * i], yp = coords[2 * i + 1];
    const float wm1 = (float)(W - 1);
    const float g = 2.0f * xp / wm1 - 1.0f;                            // :107
    float ix = ((g + 1.0f) / 2.0f) * wm1;                              // grid_sample, align_corners=True
    ix = fminf(fmaxf(ix, -2.0f), wm1 + 2.0f);                          // neutral: both taps stay out of range
    const float x0f = floorf(ix);
    const int x0 = (int)x0f;
    const float w1 = ix - x0f, w0 = (x0f + 1.0f) - ix;
    const float* row = img + n * (long)C * W;
    for (int c = 0; c < C; ++c) out[(n * C + c) * nq + q] = lookup1d_fetch(row + (long)c * W, 1, W, x0, w0, w1);
    if (mask) mask[i] = ((g > -1.0f) & (yp > -1.0f) & (g < 1.0f) & (yp < 1.0f)) ? 1.0f : 0.0f;   // :113
}

// blockIdx.y = 1 (pair launch, effi_vol_lookup1d_pair_f32): the same queries into a second volume of the same shape
__global__ void vol_lookup1d_kernel(const float* __restrict__ vol, long vds, long vps, int Dp,
                                    const float* __restrict__ query, long qds, long qys, long qxs, int nq,
                                    const float* __restrict__ dmin, const float* __restrict__ dmax, long rps,
                                    int h, int w, float* __restrict__ out, const float* __restrict__ vol_b,
                                    float* __restrict__ out_b) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= h * w) return;
    if (blockIdx.y) {
        vol = vol_b;
        out = out_b;
    }
    const int y = p / w, x = p - y * w;
    const float lo = dmin[p * rps], hi = dmax[p * rps];
    const float* v = vol + p * vps;
    for (int k = 0; k < nq; ++k) {
        const float q = query[k * qds + y * qys + x * qxs];
        out[(long)k * h * w + p] = lookup1d(v, vds, Dp, q, lo, hi);
    }
}

// GetCost.forward (models/Effi_MVS_plus.py:257-303) behind scale_inv_depth (:138-148)
__global__ void getcost_kernel(const float* __restrict__ inv_depth, const float* __restrict__ disp_range,
                               int n_range, int input_is_depth, const float* __restrict__ interval,
                               const float* __restrict__ cur_vol, long cds, long cps, int Dcur,
                               const float* __restrict__ reg_vol, long rds, long rps_, int Dreg,
                               const float* __restrict__ dmin, const float* __restrict__ dmax, long range_ps,
                               int nq, int hw, float* __restrict__ cost) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    const float itv = interval[0];
    float depth = inv_depth[p];
    if (!input_is_depth) depth = effi_inv_to_depth(depth, disp_range[0], disp_range[n_range - 1]);
    const float dv = 1.0f / depth;                                     // :272
    const float half = (float)(nq / 2) * itv;                          // module.py:558-560
    const float smin = fmaxf(dv - half, 1e-4f);
    const float smax = fminf(fmaxf(dv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(nq - 1);
    const float rlo = dmin[p * range_ps], rhi = dmax[p * range_ps];
    for (int k = 0; k < nq; ++k) {
        const float s = fmaxf(smin + (float)k * step, 1e-5f);          // module.py:566-570
        const float qd = 1.0f / s;                                     // :285
        cost[(long)k * hw + p] = lookup1d(cur_vol + p * cps, cds, Dcur, qd, rlo, rhi);
        cost[(long)(nq + k) * hw + p] = lookup1d(reg_vol + p * rps_, rds, Dreg, qd, rlo, rhi);
    }
}

// GetCost followed by the encoder's 1x1 convolution + ReLU (convc1, models/update.py:73,86): the 2*NQ looked-up costs of
// a pixel stay in registers and go straight through the [2*NQ] -> [cout] matrix (weights through the scalar cache), so the
// cost map is neither written nor read back.
struct GetcostConvArgs {
    const float* inv_depth; const float* disp_range; int n_range; int input_is_depth; const float* interval;
    const float* cur_vol; long cds, cps; int Dcur;
    const float* reg_vol; long rds, rps; int Dreg;
    const float* dmin; const float* dmax; long range_ps;
    int hw; const float* weight; const float* bias; int cout; int relu; float* out;
    // split-resident outputs (effi_encoder_inputs_bf16x3_sr; nullptr: the fp32 maps above): see effi_sr_store4, common.hpp
    unsigned short* out_sr; unsigned short* out7_sr; int w, sr_hp, sr_wp;
};

template <int NQ>
__device__ __forceinline__ void getcost_conv1x1_block(const GetcostConvArgs& g, int bx) {
    const float* __restrict__ inv_depth = g.inv_depth; const float* __restrict__ disp_range = g.disp_range;
    const int n_range = g.n_range, input_is_depth = g.input_is_depth, Dcur = g.Dcur, Dreg = g.Dreg, hw = g.hw, cout = g.cout;
    const int relu = g.relu;
    const float* __restrict__ interval = g.interval; const float* __restrict__ cur_vol = g.cur_vol;
    const float* __restrict__ reg_vol = g.reg_vol; const float* __restrict__ dmin = g.dmin; const float* __restrict__ dmax = g.dmax;
    const float* __restrict__ weight = g.weight; const float* __restrict__ bias = g.bias; float* __restrict__ out = g.out;
    const long cds = g.cds, cps = g.cps, rds = g.rds, rps_ = g.rps, range_ps = g.range_ps;
    const int p = bx * TPB + threadIdx.x;
    if (p >= hw) return;
    float cost[2 * NQ];
    effi_getcost_pixel<NQ>(inv_depth[p], input_is_depth, disp_range, n_range, interval[0], cur_vol + p * cps, cds, Dcur, reg_vol + p * rps_, rds,
                           Dreg, dmin[p * range_ps], dmax[p * range_ps], cost);
    for (int c0 = 0; c0 < cout; c0 += 8) {                 // cout % 8 == 0 (checked by the caller)
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = bias[c0 + j];
#pragma unroll
        for (int k = 0; k < 2 * NQ; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(cost[k], weight[k * cout + c0 + j], acc[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = relu ? fmaxf(acc[j], 0.0f) : acc[j];
        if (g.out_sr) {                        // split-resident output (uniform branch): the octet c0 / 8 of this pixel, 16 bytes per part
            const int y = p / g.w;
            effi_sr_store8(g.out_sr, g.sr_hp, g.sr_wp, c0, y, p - y * g.w, acc);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) out[(long)(c0 + j) * hw + p] = acc[j];
        }
    }
}

template <int NQ>
__global__ __launch_bounds__(TPB) void getcost_conv1x1_kernel(const GetcostConvArgs g) { getcost_conv1x1_block<NQ>(g, blockIdx.x); }

// Both inputs of the update block's encoder in ONE launch (models/update.py:86,90): workgroups [0, n7) are tiles of
// relu(convd1(inv_depth)) (7x7, effi_c1k7_relu_tile), the rest are blocks of relu(convc1(GetCost(inv_depth))).  The two are
// independent and each alone underfills the chip at the coarse stages; as two kernels on two streams they overlap as well,
// but every fork / join of streams inside a captured graph costs ~5 / ~11 us of idle GPU (measured), nine times per view.
// X3: the 7x7 tiles on the matrix cores in split precision (effi_c1k7_relu_tile_x3: one workgroup per pixel tile, all channels).
template <int NQ, int COUT, bool X3 = false>
__global__ __launch_bounds__(TPB) void encoder_inputs_kernel(const GetcostConvArgs g, const float* __restrict__ w7,
                                                             const float* __restrict__ b7, int h, int w,
                                                             float* __restrict__ out7, int gx, int gy, int n7) {
    const int b = blockIdx.x;
    if (b < n7) {
        if (X3) {
            effi_c1k7_relu_tile_x3<COUT>(g.inv_depth, w7, b7, h, w, out7, b % gx, b / gx, g.out7_sr, g.sr_hp, g.sr_wp);
            return;
        }
        const int bz = b / (gx * gy), r = b - bz * gx * gy;
        effi_c1k7_relu_tile<COUT>(g.inv_depth, w7, b7, h, w, out7, r % gx, r / gx, bz);
    } else {
        getcost_conv1x1_block<NQ>(g, b - n7);
    }
}

// ------------------------------------------------------------------------------------------------
// K10: convex upsampling x2 (models/Effi_MVS_plus.py:167-178) + scale_inv_depth
// ------------------------------------------------------------------------------------------------
__global__ void convex_upsample2x_kernel(const float* __restrict__ inv, const float* __restrict__ mask,
                                         const float* __restrict__ disp_range, int n_range, int h, int w,
                                         float* __restrict__ out_inv, float* __restrict__ out_depth,
                                         float* __restrict__ out_dinv) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    const long hw = (long)h * w;
    float nb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        nb[k] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? inv[(long)yy * w + xx] : 0.0f;   // F.unfold zero pad
    }
    float lo = 0.0f, hi = 0.0f;
    if (out_depth) { lo = disp_range[0]; hi = disp_range[n_range - 1]; }
    const int W2 = 2 * w;
#pragma unroll
    for (int r = 0; r < 4; ++r) {            // r = ry*2 + rx; mask channel = k*4 + r
        float mv[9], m = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mv[k] = mask[(long)(k * 4 + r) * hw + p];
            m = fmaxf(m, mv[k]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            mv[k] = expf(mv[k] - m);
            sum = sum + mv[k];
        }
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc = acc + (mv[k] / sum) * nb[k];
        const long o = (long)(2 * y + (r >> 1)) * W2 + 2 * x + (r & 1);
        if (out_inv) out_inv[o] = acc;
        if (out_depth) {
            const float dep = effi_inv_to_depth(acc, lo, hi);
            out_depth[o] = dep;
            if (out_dinv) out_dinv[o] = effi_depth_to_inv(dep, lo, hi);     // what the next stage starts from (:538)
        }
    }
}

}  // namespace

// =================================================================================================
extern "C" int effi_planar_to_nhwc_f32(const float* const* srcs, float* const* dsts, int n, int C, int HW,
                                       effi_stream_t stream) {
    if (!srcs || !dsts || n < 1 || n > EFFI_MAX_VIEWS + 1 || C < 1 || C > 128 || HW < 1) return EFFI_ERR_BADARG;
    EffiPtrList s;
    EffiOutList d;
    for (int i = 0; i < n; ++i) {
        if (!srcs[i] || !dsts[i]) return EFFI_ERR_BADARG;
        s.p[i] = srcs[i];
        d.p[i] = dsts[i];
    }
    hipLaunchKernelGGL(planar_to_nhwc_kernel, dim3(effi_cdiv(HW, 64), n), dim3(TPB), (size_t)C * 65 * sizeof(float),
                       effi_s(stream), s, d, C, HW);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_split_tanh_relu_f32(const float* ctx, int hd, int cd, int hw, float* hidden, float* inp,
                                        effi_stream_t stream) {
    if (!ctx || !hidden || !inp || hd < 1 || cd < 1 || hw < 1) return EFFI_ERR_BADARG;
    const long n = (long)(hd + cd) * hw;
    if ((hw & 3) == 0 && ((reinterpret_cast<uintptr_t>(ctx) | reinterpret_cast<uintptr_t>(hidden) | reinterpret_cast<uintptr_t>(inp)) & 15) == 0)
        hipLaunchKernelGGL(split_tanh_relu4_kernel, dim3(effi_cdiv(n / 4, TPB)), dim3(TPB), 0, effi_s(stream),
                           reinterpret_cast<const float4*>(ctx), (long)hd * hw / 4, n / 4, reinterpret_cast<float4*>(hidden),
                           reinterpret_cast<float4*>(inp));
    else
        hipLaunchKernelGGL(split_tanh_relu_kernel, dim3(min(effi_cdiv(n, TPB), 4096)), dim3(TPB), 0, effi_s(stream),
                           ctx, hd, cd, hw, hidden, inp);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_split_tanh_relu_stages_f32(const float* const* ctx, const int* hd, const int* cd, const int* hw,
                                               float* const* hidden, float* const* inp, int n_stages, effi_stream_t stream) {
    if (!ctx || !hd || !cd || !hw || !hidden || !inp || n_stages < 1 || n_stages > 4) return EFFI_ERR_BADARG;
    SplitStages a;
    int blocks = 0;
    for (int k = 0; k < 4; ++k) {
        const int j = k < n_stages ? k : 0;
        if (!ctx[j] || !hidden[j] || !inp[j] || hd[j] < 1 || cd[j] < 1 || hw[j] < 1) return EFFI_ERR_BADARG;
        a.ctx[k] = ctx[j]; a.hidden[k] = hidden[j]; a.inp[k] = inp[j];
        a.nh[k] = (long)hd[j] * hw[j];
        a.n[k] = (long)(hd[j] + cd[j]) * hw[j];
        a.first[k] = blocks;
        if (k < n_stages) blocks += effi_cdiv(a.n[k], 4L * TPB);
    }
    for (int k = n_stages; k < 4; ++k) a.first[k] = blocks;      // unused maps own no blocks
    a.first[4] = blocks;
    a.vec4 = 1;
    for (int k = 0; k < n_stages; ++k)
        if ((a.nh[k] & 3) || (a.n[k] & 3) ||
            ((reinterpret_cast<uintptr_t>(ctx[k]) | reinterpret_cast<uintptr_t>(hidden[k]) | reinterpret_cast<uintptr_t>(inp[k])) & 15))
            a.vec4 = 0;
    hipLaunchKernelGGL(split_tanh_relu_stages_kernel, dim3(blocks), dim3(TPB), 0, effi_s(stream), a);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_depth_to_inv_f32(const float* depth, const float* disp_range, int n_range, int n, float* inv,
                                     effi_stream_t stream) {
    if (!depth || !disp_range || !inv || n_range < 2 || n < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(depth_to_inv_kernel, dim3(min(effi_cdiv(n, TPB), 4096)), dim3(TPB), 0, effi_s(stream),
                       depth, disp_range, n_range, n, inv);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_stage1_hypotheses_f32(const float* disp_range, int n_range, int D, float* depths,
                                          float* intervals, effi_stream_t stream) {
    if (!disp_range || !depths || !intervals || n_range < 2 || D < 2) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(stage1_hypotheses_kernel, dim3(1), dim3(128), 0, effi_s(stream), disp_range, n_range, D,
                       depths, intervals);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_upsample_nearest_f32(const float* in, int C, int h, int w, int f, float* out,
                                         effi_stream_t stream) {
    if (!in || !out || C < 1 || h < 1 || w < 1 || f < 1) return EFFI_ERR_BADARG;
    const long n = (long)C * h * f * w * f;
    hipLaunchKernelGGL(upsample_nearest_kernel, dim3(min(effi_cdiv(n, TPB), 8192)), dim3(TPB), 0, effi_s(stream),
                       in, C, h, w, f, out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_head_update_f32(const float* partial9, const float* bias2, const float* inv_depth, const float* disp_range,
                                    int n_range, int h, int w, float* out_inv, float* out_depth, effi_stream_t stream) {
    if (!partial9 || !bias2 || !inv_depth || !disp_range || n_range < 2 || !out_inv || !out_depth || h < 1 || w < 1)
        return EFFI_ERR_BADARG;
    if (w & 3) return EFFI_ERR_UNSUPPORTED;
    const long n = (long)h * (w >> 2);
    hipLaunchKernelGGL(head_update_kernel, dim3((unsigned)effi_cdiv(n, TPB)), dim3(TPB), 0, effi_s(stream), partial9, bias2, inv_depth,
                       disp_range, n_range, h, w, out_inv, out_depth);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_view_aggregate_f32(const float* sim_views, const float* weights, int S, int D, int hw,
                                       float* out, effi_stream_t stream) {
    if (!sim_views || !out || S < 1 || S > EFFI_MAX_VIEWS || D < 1 || hw < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(view_aggregate_kernel, dim3(effi_cdiv(hw, TPB), effi_cdiv(D, 8)), dim3(TPB), 0, effi_s(stream),
                       sim_views, weights, S, D, hw, out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_softmax_regress_conf_f32(const float* logits, const float* depth, long dds, long dps, int D,
                                             int hw, float* out_depth, float* out_conf, const float* disp_range, int n_range,
                                             float* out_depth_inv, effi_stream_t stream) {
    if (!logits || !depth || !out_depth || !out_conf || D < 1 || hw < 1) return EFFI_ERR_BADARG;
    if (out_depth_inv && (!disp_range || n_range < 2)) return EFFI_ERR_BADARG;
    const dim3 grid(effi_cdiv(hw, 64));
    hipStream_t st = effi_s(stream);
#define EFFI_SM(DT)                                                                                                      \
    hipLaunchKernelGGL(softmax_regress_conf_reg_kernel<DT>, grid, dim3(64), 0, st, logits, depth, dds, dps, hw, out_depth, \
                       out_conf, disp_range, n_range, out_depth_inv)
    switch (D) {
        case 8: EFFI_SM(8); break;
        case 16: EFFI_SM(16); break;
        case 32: EFFI_SM(32); break;
        case 48: EFFI_SM(48); break;
        case 64: EFFI_SM(64); break;
        case 96: EFFI_SM(96); break;
        default:
            hipLaunchKernelGGL(softmax_regress_conf_kernel, grid, dim3(64), 0, st, logits, depth, dds, dps, D, hw, out_depth, out_conf,
                               disp_range, n_range, out_depth_inv);
    }
#undef EFFI_SM
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_vol_lookup1d_f32(const float* vol, long vds, long vps, int Dp, const float* query, long qds,
                                     long qys, long qxs, int nq, const float* dmin, const float* dmax, long rps,
                                     int h, int w, float* out, effi_stream_t stream) {
    if (!vol || !query || !dmin || !dmax || !out || Dp < 2 || nq < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(vol_lookup1d_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, effi_s(stream), vol, vds,
                       vps, Dp, query, qds, qys, qxs, nq, dmin, dmax, rps, h, w, out, nullptr, nullptr);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_bilinear_sampler1d_f32(const float* img, int N, int C, int W, const float* coords, long nq, float* out,
                                           float* mask, effi_stream_t stream) {
    if (!img || !coords || !out || N < 1 || C < 1 || W < 1 || nq < 1) return EFFI_ERR_BADARG;
    const long total = (long)N * nq;
    hipLaunchKernelGGL(bilinear_sampler1d_kernel, dim3(effi_cdiv(total, TPB)), dim3(TPB), 0, effi_s(stream), img, C, W, coords, nq,
                       total, out, mask);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_vol_lookup1d_pair_f32(const float* vol_a, const float* vol_b, long vds, long vps, int Dp, const float* query,
                                          long qds, long qys, long qxs, int nq, const float* dmin, const float* dmax, long rps,
                                          int h, int w, float* out_a, float* out_b, effi_stream_t stream) {
    if (!vol_a || !vol_b || !query || !dmin || !dmax || !out_a || !out_b || Dp < 2 || nq < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(vol_lookup1d_kernel, dim3(effi_cdiv((long)h * w, TPB), 2), dim3(TPB), 0, effi_s(stream), vol_a, vds,
                       vps, Dp, query, qds, qys, qxs, nq, dmin, dmax, rps, h, w, out_a, vol_b, out_b);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_getcost_f32(const float* inv_depth, const float* disp_range, int n_range, int input_is_depth,
                                const float* interval,
                                const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds,
                                long rps, int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h,
                                int w, float* cost, effi_stream_t stream) {
    if (!inv_depth || !interval || !cur_vol || !reg_vol || !dmin || !dmax || !cost) return EFFI_ERR_BADARG;
    if (!input_is_depth && (!disp_range || n_range < 2)) return EFFI_ERR_BADARG;
    if ( Dcur < 2 || Dreg < 2 || nq < 2 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(getcost_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, effi_s(stream), inv_depth,
                       disp_range, n_range, input_is_depth, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps, Dreg, dmin, dmax,
                       range_ps, nq, h * w, cost);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_getcost_conv1x1_f32(const float* inv_depth, const float* disp_range, int n_range, int input_is_depth,
                                        const float* interval, const float* cur_vol, long cds, long cps, int Dcur,
                                        const float* reg_vol, long rds, long rps, int Dreg, const float* dmin,
                                        const float* dmax, long range_ps, int nq, int h, int w, const float* weight,
                                        const float* bias, int cout, int relu, float* out, effi_stream_t stream) {
    if (!inv_depth || !interval || !cur_vol || !reg_vol || !dmin || !dmax || !weight || !bias || !out) return EFFI_ERR_BADARG;
    if (!input_is_depth && (!disp_range || n_range < 2)) return EFFI_ERR_BADARG;
    if (Dcur < 2 || Dreg < 2 || h < 1 || w < 1 || cout < 8) return EFFI_ERR_BADARG;
    if (cout % 8) return EFFI_ERR_UNSUPPORTED;
    const dim3 grid(effi_cdiv((long)h * w, TPB));
    hipStream_t st = effi_s(stream);
    const GetcostConvArgs g{inv_depth, disp_range, n_range, input_is_depth, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps,
                            Dreg, dmin, dmax, range_ps, h * w, weight, bias, cout, relu, out};
#define EFFI_GC(NQ) hipLaunchKernelGGL(getcost_conv1x1_kernel<NQ>, grid, dim3(TPB), 0, st, g)
    switch (nq) {
        case 2: EFFI_GC(2); break;
        case 3: EFFI_GC(3); break;
        case 4: EFFI_GC(4); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
#undef EFFI_GC
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

static int encoder_inputs_launch(bool x3, const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                 const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds,
                                 long rps, int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h,
                                 int w, const float* weight_c1, const float* bias_c1, const float* weight_d1,
                                 const float* bias_d1, int cout, float* out_c1, float* out_d1, effi_stream_t stream,
                                 void* sr_c1 = nullptr, void* sr_d1 = nullptr, int hp = 0, int wp = 0) {
    const bool sr = sr_c1 != nullptr;
    if (!inv_depth || !interval || !cur_vol || !reg_vol || !dmin || !dmax || !weight_c1 || !bias_c1 || !weight_d1 || !bias_d1 ||
        (!sr && (!out_c1 || !out_d1)) || !disp_range || n_range < 2)
        return EFFI_ERR_BADARG;
    if (sr && (!sr_d1 || !x3 || hp < h + 2 || wp < w + 2 || ((reinterpret_cast<uintptr_t>(sr_c1) | reinterpret_cast<uintptr_t>(sr_d1)) & 15)))
        return EFFI_ERR_BADARG;
    if (Dcur < 2 || Dreg < 2 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (nq != 3 || (cout != 16 && cout != 32 && cout != 48)) return EFFI_ERR_UNSUPPORTED;
    const int gx = effi_cdiv(w, EFFI_C1K7_TX), gy = effi_cdiv(h, EFFI_C1K7_TY), n7 = gx * gy * (x3 ? 1 : cout / 16);
    const dim3 grid(n7 + effi_cdiv((long)h * w, TPB));
    hipStream_t st = effi_s(stream);
    const GetcostConvArgs g{inv_depth, disp_range, n_range, 0, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps,
                            Dreg, dmin, dmax, range_ps, h * w, weight_c1, bias_c1, cout, 1, out_c1,
                            reinterpret_cast<unsigned short*>(sr_c1), reinterpret_cast<unsigned short*>(sr_d1), w, hp, wp};
#define EFFI_EI(CO)                                                                                                                     \
    do {                                                                                                                                \
        if (x3) hipLaunchKernelGGL((encoder_inputs_kernel<3, CO, true>), grid, dim3(TPB), 0, st, g, weight_d1, bias_d1, h, w, out_d1, gx, gy, n7);  \
        else hipLaunchKernelGGL((encoder_inputs_kernel<3, CO, false>), grid, dim3(TPB), 0, st, g, weight_d1, bias_d1, h, w, out_d1, gx, gy, n7);    \
    } while (0)
    switch (cout) {
        case 16: EFFI_EI(16); break;
        case 32: EFFI_EI(32); break;
        default: EFFI_EI(48); break;
    }
#undef EFFI_EI
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_encoder_inputs_f32(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                       const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds,
                                       long rps, int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h,
                                       int w, const float* weight_c1, const float* bias_c1, const float* weight_d1,
                                       const float* bias_d1, int cout, float* out_c1, float* out_d1, effi_stream_t stream) {
    return encoder_inputs_launch(false, inv_depth, disp_range, n_range, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps, Dreg, dmin,
                                 dmax, range_ps, nq, h, w, weight_c1, bias_c1, weight_d1, bias_d1, cout, out_c1, out_d1, stream);
}

extern "C" int effi_encoder_inputs_bf16x3_f32(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                              const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds,
                                              long rps, int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h,
                                              int w, const float* weight_c1, const float* bias_c1, const float* weight_d1,
                                              const float* bias_d1, int cout, float* out_c1, float* out_d1, effi_stream_t stream) {
    return encoder_inputs_launch(true, inv_depth, disp_range, n_range, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps, Dreg, dmin,
                                 dmax, range_ps, nq, h, w, weight_c1, bias_c1, weight_d1, bias_d1, cout, out_c1, out_d1, stream);
}

// The same launch writing SPLIT-RESIDENT maps (csrc/common.hpp: effi_sr_store4) for the split-precision 3x3 convolutions that follow
// (convc2 / convd2, models/update.py:87,91): sr_c1 / sr_d1 are bf16 [cout/8][hi|lo][hp][wp][8] with a zero border.
extern "C" int effi_encoder_inputs_bf16x3_sr(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                             const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds,
                                             long rps, int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h,
                                             int w, const float* weight_c1, const float* bias_c1, const float* weight_d1,
                                             const float* bias_d1, int cout, void* sr_c1, void* sr_d1, int hp, int wp,
                                             effi_stream_t stream) {
    if (!sr_c1 || !sr_d1) return EFFI_ERR_BADARG;
    return encoder_inputs_launch(true, inv_depth, disp_range, n_range, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps, Dreg, dmin,
                                 dmax, range_ps, nq, h, w, weight_c1, bias_c1, weight_d1, bias_d1, cout, nullptr, nullptr, stream,
                                 sr_c1, sr_d1, hp, wp);
}

// ---- split-resident maps: geometry, border, conversion from an fp32 map (block boundary of the update block) -------------------
namespace {
struct SrMaps {                       // up to 4 groups of planes with one geometry each (the stages of the cascade)
    unsigned short* base[4];
    int planes[4], h[4], w[4], hp[4], wp[4];
    int first[5];                     // first block of each group
};
// zero what a producer never writes: row 0, rows h+1.., column 0, columns w+1.. of every plane (16-byte units)
__global__ __launch_bounds__(TPB) void sr_clear_border_kernel(SrMaps a) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) k += ((int)blockIdx.x >= a.first[j]) ? 1 : 0;
    unsigned short* base = a.base[0];
    int planes = a.planes[0], h = a.h[0], w = a.w[0], hp = a.hp[0], wp = a.wp[0], b0 = a.first[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (k == j) { base = a.base[j]; planes = a.planes[j]; h = a.h[j]; w = a.w[j]; hp = a.hp[j]; wp = a.wp[j]; b0 = a.first[j]; }
    // border units of one plane: (hp - h) full rows + h rows of (wp - w) units
    const int rowpart = (hp - h) * wp, per_plane = rowpart + h * (wp - w);
    const long i = (long)(blockIdx.x - b0) * TPB + threadIdx.x;
    if (i >= (long)planes * per_plane) return;
    const int pl = (int)(i / per_plane), r = (int)(i - (long)pl * per_plane);
    int y, x;
    if (r < rowpart) {
        const int ry = r / wp;
        x = r - ry * wp;
        y = ry == 0 ? 0 : h + ry;                                  // row 0, then rows h+1 .. hp-1
    } else {
        const int q = r - rowpart, ry = q / (wp - w), cx = q - ry * (wp - w);
        y = 1 + ry;
        x = cx == 0 ? 0 : w + cx;                                  // column 0, then columns w+1 .. wp-1
    }
    *reinterpret_cast<float4*>(base + (((long)pl * hp + y) * wp + x) * 8) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// fp32 planar [C][h][w] -> split-resident map: a thread owns (octet, pixel)
__global__ __launch_bounds__(TPB) void sr_from_planar_kernel(const float* __restrict__ in, int C, int h, int w, unsigned short* __restrict__ sr,
                                                             int hp, int wp) {
    const long hw = (long)h * w, i = (long)blockIdx.x * TPB + threadIdx.x;
    if (i >= (long)(C >> 3) * hw) return;
    const int o = (int)(i / hw);
    const long p = i - (long)o * hw;
    const int y = (int)(p / w), x = (int)(p - (long)y * w);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = in[(long)(8 * o + e) * hw + p];
    effi_sr_store8(sr, hp, wp, 8 * o, y, x, v);
}

// the hidden halves of up to 4 context maps as fp32 AND split-resident maps, the input halves as fp32 (split_tanh_relu_stages + SR)
struct SplitStagesSr {
    const float* ctx[4];
    float* hidden[4];
    float* inp[4];
    unsigned short* hidden_sr[4];
    int hd[4], cd[4], h[4], w[4], hp[4], wp[4], q4[4];
    int first[5];
};
__global__ __launch_bounds__(TPB) void split_tanh_relu_stages_sr_kernel(SplitStagesSr a) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j) k += ((int)blockIdx.x >= a.first[j]) ? 1 : 0;
    const float* __restrict__ ctx = a.ctx[0];
    float* __restrict__ hidden = a.hidden[0];
    float* __restrict__ inp = a.inp[0];
    unsigned short* __restrict__ hsr = a.hidden_sr[0];
    int hd = a.hd[0], cd = a.cd[0], h = a.h[0], w = a.w[0], hp = a.hp[0], wp = a.wp[0], b0 = a.first[0], q4 = a.q4[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (k == j) {
            ctx = a.ctx[j]; hidden = a.hidden[j]; inp = a.inp[j]; hsr = a.hidden_sr[j];
            hd = a.hd[j]; cd = a.cd[j]; h = a.h[j]; w = a.w[j]; hp = a.hp[j]; wp = a.wp[j]; b0 = a.first[j]; q4 = a.q4[j];
        }
    // work items: (octet of hidden channels, pixel) for the hidden half [hd % 8 == 0], then (channel quad, pixel) for the input half [cd % 4 == 0]
    const long hw = (long)h * w, n_h = (long)(hd >> 3) * hw, n_i = (long)(cd >> 2) * hw;
    const long i = (long)(blockIdx.x - b0) * TPB + threadIdx.x;
    if (i < n_h) {
        const int o = (int)(i / hw);
        const long p = i - (long)o * hw;
        const int y = (int)(p / w), x = (int)(p - (long)y * w);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tanhf(ctx[(long)(8 * o + e) * hw + p]);
        if (q4) {                                  // [hd/4][h][w][4]: the octet is two 16-byte pieces (EFFI_EPI_Q4)
            *reinterpret_cast<float4*>(hidden + ((long)(2 * o) * hw + p) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(hidden + ((long)(2 * o + 1) * hw + p) * 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) hidden[(long)(8 * o + e) * hw + p] = v[e];
        }
        effi_sr_store8(hsr, hp, wp, 8 * o, y, x, v);
    } else if (i < n_h + n_i) {
        const long q = i - n_h;
        const int c4 = (int)(q / hw);
        const long p = q - (long)c4 * hw;
#pragma unroll
        for (int e = 0; e < 4; ++e) inp[(long)(4 * c4 + e) * hw + p] = fmaxf(ctx[(long)(hd + 4 * c4 + e) * hw + p], 0.0f);
    }
}
}  // namespace

extern "C" int effi_sr_geometry(int h, int w, int* hp, int* wp) {
    if (h < 1 || w < 1 || !hp || !wp) return EFFI_ERR_BADARG;
    *hp = ((h + 15) & ~15) + 2;            // one zero row above, rows down to the bottom of the last 16-row tile + 1 below
    // one zero column left, columns up to the right edge of the last tile + 1: the 4 x 64 tiles are only used from 512 columns on
    // (conv2d_x3.hpp: launch_bf16x3), below that tiles are 16 columns wide
    *wp = (w >= 512 ? ((w + 63) & ~63) : ((w + 15) & ~15)) + 2;
    return EFFI_OK;
}

extern "C" int effi_sr_clear_border(void* const* maps, const int* planes, const int* h, const int* w, const int* hp, const int* wp,
                                    int n_groups, effi_stream_t stream) {
    if (!maps || !planes || !h || !w || !hp || !wp || n_groups < 1 || n_groups > 4) return EFFI_ERR_BADARG;
    SrMaps a;
    int blocks = 0;
    for (int k = 0; k < 4; ++k) {
        const int j = k < n_groups ? k : 0;
        if (!maps[j] || planes[j] < 1 || h[j] < 1 || w[j] < 1 || hp[j] < h[j] + 2 || wp[j] < w[j] + 2 || (reinterpret_cast<uintptr_t>(maps[j]) & 15))
            return EFFI_ERR_BADARG;
        a.base[k] = reinterpret_cast<unsigned short*>(maps[j]);
        a.planes[k] = planes[j]; a.h[k] = h[j]; a.w[k] = w[j]; a.hp[k] = hp[j]; a.wp[k] = wp[j];
        a.first[k] = blocks;
        if (k < n_groups) blocks += effi_cdiv((long)planes[j] * ((long)(hp[j] - h[j]) * wp[j] + (long)h[j] * (wp[j] - w[j])), TPB);
    }
    a.first[4] = blocks;
    hipLaunchKernelGGL(sr_clear_border_kernel, dim3(blocks), dim3(TPB), 0, effi_s(stream), a);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_sr_from_planar_f32(const float* in, int channels, int h, int w, void* sr, int hp, int wp, effi_stream_t stream) {
    if (!in || !sr || channels < 8 || (channels & 7) || h < 1 || w < 1 || hp < h + 2 || wp < w + 2 || (reinterpret_cast<uintptr_t>(sr) & 15))
        return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(sr_from_planar_kernel, dim3(effi_cdiv((long)(channels >> 3) * h * w, TPB)), dim3(TPB), 0, effi_s(stream), in,
                       channels, h, w, reinterpret_cast<unsigned short*>(sr), hp, wp);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_split_tanh_relu_stages_sr_f32(const float* const* ctx, const int* hd, const int* cd, const int* h, const int* w,
                                                  float* const* hidden, void* const* hidden_sr, const int* hp, const int* wp,
                                                  float* const* inp, const int* hidden_q4, int n_stages, effi_stream_t stream) {
    if (!ctx || !hd || !cd || !h || !w || !hidden || !hidden_sr || !hp || !wp || !inp || n_stages < 1 || n_stages > 4) return EFFI_ERR_BADARG;
    SplitStagesSr a;
    int blocks = 0;
    for (int k = 0; k < 4; ++k) {
        const int j = k < n_stages ? k : 0;
        if (!ctx[j] || !hidden[j] || !hidden_sr[j] || !inp[j] || hd[j] < 8 || cd[j] < 4 || h[j] < 1 || w[j] < 1) return EFFI_ERR_BADARG;
        if ((hd[j] & 7) || (cd[j] & 3)) return EFFI_ERR_UNSUPPORTED;
        if (hp[j] < h[j] + 2 || wp[j] < w[j] + 2 || (reinterpret_cast<uintptr_t>(hidden_sr[j]) & 15)) return EFFI_ERR_BADARG;
        a.ctx[k] = ctx[j]; a.hidden[k] = hidden[j]; a.inp[k] = inp[j]; a.hidden_sr[k] = reinterpret_cast<unsigned short*>(hidden_sr[j]);
        a.hd[k] = hd[j]; a.cd[k] = cd[j]; a.h[k] = h[j]; a.w[k] = w[j]; a.hp[k] = hp[j]; a.wp[k] = wp[j];
        a.q4[k] = hidden_q4 ? hidden_q4[j] : 0;
        if (a.q4[k] && (reinterpret_cast<uintptr_t>(hidden[j]) & 15)) return EFFI_ERR_BADARG;
        a.first[k] = blocks;
        if (k < n_stages) blocks += effi_cdiv((long)h[j] * w[j] * ((hd[j] >> 3) + (cd[j] >> 2)), TPB);
    }
    a.first[4] = blocks;
    hipLaunchKernelGGL(split_tanh_relu_stages_sr_kernel, dim3(blocks), dim3(TPB), 0, effi_s(stream), a);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

namespace {
template <int COUT>
__global__ __launch_bounds__(TPB) void conv2d_c1k7_relu_x3_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                                  const float* __restrict__ bias, int h, int w, float* __restrict__ out) {
    effi_c1k7_relu_tile_x3<COUT>(in, wgt, bias, h, w, out, blockIdx.x, blockIdx.y);
}
}  // namespace

extern "C" int effi_conv2d_c1k7_relu_bf16x3_f32(const float* in, const float* weight, const float* bias, int cout, int h, int w,
                                                float* out, effi_stream_t stream) {
    if (!in || !weight || !bias || !out || h < 1 || w < 1) return EFFI_ERR_BADARG;
    const dim3 grid(effi_cdiv(w, EFFI_C1K7_TX), effi_cdiv(h, EFFI_C1K7_TY));
    hipStream_t st = effi_s(stream);
    switch (cout) {
        case 16: hipLaunchKernelGGL(conv2d_c1k7_relu_x3_kernel<16>, grid, dim3(TPB), 0, st, in, weight, bias, h, w, out); break;
        case 32: hipLaunchKernelGGL(conv2d_c1k7_relu_x3_kernel<32>, grid, dim3(TPB), 0, st, in, weight, bias, h, w, out); break;
        case 48: hipLaunchKernelGGL(conv2d_c1k7_relu_x3_kernel<48>, grid, dim3(TPB), 0, st, in, weight, bias, h, w, out); break;
        default: return EFFI_ERR_UNSUPPORTED;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_convex_upsample2x_f32(const float* inv_depth, const float* mask, const float* disp_range,
                                          int n_range, int h, int w, float* out_inv, float* out_depth,
                                          float* out_depth_inv, effi_stream_t stream) {
    if (!inv_depth || !mask || (!out_inv && !out_depth) || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (out_depth && (!disp_range || n_range < 2)) return EFFI_ERR_BADARG;
    if (out_depth_inv && !out_depth) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(convex_upsample2x_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, effi_s(stream),
                       inv_depth, mask, disp_range, n_range, h, w, out_inv, out_depth, out_depth_inv);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
