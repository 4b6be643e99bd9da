// K4: view-weight net (models/Effi_MVS_plus.py:361-362): three 3x3 conv+BN+ReLU layers
// (1->16->16->8), a 1x1 conv (8->1) and a sigmoid, fused into one kernel.  A 256-thread block owns a
// 16x16 output tile; the 22x22 input patch and both intermediate activations live in LDS (48 KiB), so
// the only HBM traffic is the entropy map in and the weight map out.  Weights are read through the
// scalar cache (uniform addresses) and feed v_fma as SGPR operands.
//
// params (host-packed, BN folded): w0[9][16] b0[16] | w1[16][9][16] b1[16] | w2[16][9][8] b2[8] | w3[8] b3[1]
// i.e. every 3x3 layer is stored [cin][tap][cout].
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int T = 16;                 // output tile
constexpr int IN_W = T + 6;           // 22
constexpr int A1_W = T + 4;           // 20
constexpr int A2_W = T + 2;           // 18
constexpr int OFF_W0 = 0, OFF_B0 = 144, OFF_W1 = 160, OFF_B1 = OFF_W1 + 16 * 9 * 16, OFF_W2 = OFF_B1 + 16,
              OFF_B2 = OFF_W2 + 16 * 9 * 8, OFF_W3 = OFF_B2 + 8, OFF_B3 = OFF_W3 + 8;

__global__ __launch_bounds__(256) void pixelwise_net_kernel(const float* __restrict__ entropy,
                                                            const float* __restrict__ prm, int h, int w,
                                                            float* __restrict__ weight) {
    __shared__ float s_in[IN_W * IN_W];
    __shared__ float s_a1[16 * A1_W * A1_W];
    __shared__ float s_a2[16 * A2_W * A2_W];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T;
    const float* __restrict__ ent = entropy + (long)blockIdx.z * h * w;

    for (int e = tid; e < IN_W * IN_W; e += 256) {
        const int py = e / IN_W, px = e - py * IN_W;
        const int gy = y0 - 3 + py, gx = x0 - 3 + px;
        s_in[e] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? ent[(long)gy * w + gx] : 0.0f;
    }
    __syncthreads();

    // layer 1: 1 -> 16 on the 20x20 patch
    for (int e = tid; e < A1_W * A1_W; e += 256) {
        const int py = e / A1_W, px = e - py * A1_W;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        const bool inside = (gy >= 0 && gy < h && gx >= 0 && gx < w);
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = prm[OFF_B0 + c];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float v = s_in[(py + k / 3) * IN_W + px + k % 3];
#pragma unroll
            for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, prm[OFF_W0 + k * 16 + c], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) s_a1[c * (A1_W * A1_W) + e] = inside ? fmaxf(acc[c], 0.0f) : 0.0f;
    }
    __syncthreads();

    // layer 2: 16 -> 16 on the 18x18 patch
    for (int e = tid; e < A2_W * A2_W; e += 256) {
        const int py = e / A2_W, px = e - py * A2_W;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool inside = (gy >= 0 && gy < h && gx >= 0 && gx < w);
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = prm[OFF_B1 + c];
        for (int ci = 0; ci < 16; ++ci) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float v = s_a1[ci * (A1_W * A1_W) + (py + k / 3) * A1_W + px + k % 3];
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, prm[OFF_W1 + (ci * 9 + k) * 16 + c], acc[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) s_a2[c * (A2_W * A2_W) + e] = inside ? fmaxf(acc[c], 0.0f) : 0.0f;
    }
    __syncthreads();

    // layer 3: 16 -> 8, then 1x1 8 -> 1 and sigmoid, one output pixel per thread
    {
        const int py = tid / T, px = tid - py * T;
        const int gy = y0 + py, gx = x0 + px;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = prm[OFF_B2 + c];
        for (int ci = 0; ci < 16; ++ci) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float v = s_a2[ci * (A2_W * A2_W) + (py + k / 3) * A2_W + px + k % 3];
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, prm[OFF_W2 + (ci * 9 + k) * 8 + c], acc[c]);
            }
        }
        float o = prm[OFF_B3];
#pragma unroll
        for (int c = 0; c < 8; ++c) o = fmaf(fmaxf(acc[c], 0.0f), prm[OFF_W3 + c], o);
        if (gy < h && gx < w) weight[(long)blockIdx.z * h * w + (long)gy * w + gx] = effi_sigmoid(o);
    }
}

// The same network with layers 2 and 3 on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation in
// k order -- the fmaf chain of the kernel above).  The vector form spends 2,304 FMAs per pixel of the 18x18 patch on layer 2 and
// runs its second pass with 68 of 256 threads; here a wave owns 16-pixel tiles of the patch: D[cout][pixel] += W[cout][k] * act[k][pixel],
// k = (cin, tap) in 36 steps of 4.  The weight fragments of a layer (36 registers per lane) are loaded once; a lane's activation
// operand is one LDS read at (its pixel + its quarter's base) + a compile-time offset.  Layer 3 has 8 output channels (rows 8-15 of the tile are
// zero); its 1x1 + sigmoid tail runs as the original chain over the 8 channels: lane quarter 0 (channels 0-3) hands its partial
// sum to quarter 1 (channels 4-7) through LDS after a barrier -- nothing crosses lanes in registers.
typedef float pn_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void pixelwise_net_mfma_kernel(const float* __restrict__ entropy,
                                                                 const float* __restrict__ prm, int h, int w,
                                                                 float* __restrict__ weight) {
    __shared__ float s_in[IN_W * IN_W];
    __shared__ float s_a1[16 * A1_W * A1_W];
    __shared__ float s_a2[16 * A2_W * A2_W];
    __shared__ float s_part[T * T];
    __shared__ __attribute__((aligned(16))) float s_w0[160];              // layer 1: w0[9][16] | b0[16]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kq = lane >> 4;
    if (tid < 160) s_w0[tid] = prm[tid];                                  // OFF_W0 = 0, OFF_B0 = 144
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T;
    const float* __restrict__ ent = entropy + (long)blockIdx.z * h * w;

    for (int e = tid; e < IN_W * IN_W; e += 256) {
        const int py = e / IN_W, px = e - py * IN_W;
        const int gy = y0 - 3 + py, gx = x0 - 3 + px;
        s_in[e] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? ent[(long)gy * w + gx] : 0.0f;
    }
    // layer-2 operands of this lane: weights W1[cout = n][k = 4s + kq] and the LDS offset of k = (cin, tap) in the 20x20 planes
    // K order: lane quarter kq owns input channels 4 kq .. 4 kq + 3, step s_ = (channel within the quarter, tap) -- the LDS offset
    // of a step is then a compile-time constant on top of the quarter's base (no offset registers: 3 workgroups per CU instead of 2,
    // which is what decides the launch: 65 workgroups per XCD on 32 CUs).  The accumulation order therefore differs from the
    // vector kernel's single chain (same products, ~1e-7 relative).
    float wa[36];
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_) wa[s_] = prm[OFF_W1 + (kq * 36 + s_) * 16 + n];
    __syncthreads();

    // layer 1: 1 -> 16 on the 20x20 patch (144 FMAs per pixel: vector ALU, as above).  Its 160 parameters are read from LDS as
    // broadcast quads (s_w0, filled before the barrier above): as scalar operands they did not fit the scalar registers next to
    // the rest of the kernel and were spilled through v_writelane / v_readlane (397 of them in the ISA of round 4)
    for (int e = tid; e < A1_W * A1_W; e += 256) {
        const int py = e / A1_W, px = e - py * A1_W;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        const bool inside = (gy >= 0 && gy < h && gx >= 0 && gx < w);
        pn_f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = *reinterpret_cast<const pn_f32x4*>(&s_w0[144 + 4 * q]);
#pragma unroll 1
        for (int k = 0; k < 9; ++k) {
            const int ky = k / 3;
            const float v = s_in[(py + ky) * IN_W + px + k - 3 * ky];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const pn_f32x4 wq = *reinterpret_cast<const pn_f32x4*>(&s_w0[k * 16 + 4 * q]);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[q][r] = fmaf(v, wq[r], acc[q][r]);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) s_a1[c * (A1_W * A1_W) + e] = inside ? fmaxf(acc[c >> 2][c & 3], 0.0f) : 0.0f;
    }
    __syncthreads();

    // layer 2: 16 -> 16 on the 18x18 patch = 21 tiles of 16 pixels (the last one partial), tiles wv, wv + 4, ...
    {
        const pn_f32x4 b1 = {prm[OFF_B1 + 4 * kq], prm[OFF_B1 + 4 * kq + 1], prm[OFF_B1 + 4 * kq + 2], prm[OFF_B1 + 4 * kq + 3]};
        constexpr int NPX = A2_W * A2_W, NTILE = (NPX + 15) / 16;
        for (int t = wv; t < NTILE; t += 4) {
            const int e = 16 * t + n, ec = min(e, NPX - 1);
            const int py = ec / A2_W, px = ec - py * A2_W;
            const float* __restrict__ src = s_a1 + kq * 4 * (A1_W * A1_W) + py * A1_W + px;
            pn_f32x4 acc = b1;
#pragma unroll
            for (int s_ = 0; s_ < 36; ++s_)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_], src[(s_ / 9) * (A1_W * A1_W) + ((s_ % 9) / 3) * A1_W + s_ % 3], acc, 0, 0, 0);
            if (e < NPX) {
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool inside = (gy >= 0 && gy < h && gx >= 0 && gx < w);
#pragma unroll
                for (int r = 0; r < 4; ++r) s_a2[(4 * kq + r) * NPX + e] = inside ? fmaxf(acc[r], 0.0f) : 0.0f;
            }
        }
    }
    // layer-3 operands: W2[cout = n < 8][k], offsets in the 18x18 planes
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_) wa[s_] = (n < 8) ? prm[OFF_W2 + (kq * 36 + s_) * 8 + n] : 0.0f;
    __syncthreads();

    // layer 3: 16 -> 8 on the 16x16 tile = 16 row tiles, rows wv, wv + 4, wv + 8, wv + 12 for this wave
    pn_f32x4 a3[4];
    {
        pn_f32x4 b2 = {0.0f, 0.0f, 0.0f, 0.0f};
        if (kq < 2) b2 = pn_f32x4{prm[OFF_B2 + 4 * kq], prm[OFF_B2 + 4 * kq + 1], prm[OFF_B2 + 4 * kq + 2], prm[OFF_B2 + 4 * kq + 3]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int py = wv + 4 * i;
            const float* __restrict__ src = s_a2 + kq * 4 * (A2_W * A2_W) + py * A2_W + n;
            pn_f32x4 acc = b2;
#pragma unroll
            for (int s_ = 0; s_ < 36; ++s_)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_], src[(s_ / 9) * (A2_W * A2_W) + ((s_ % 9) / 3) * A2_W + s_ % 3], acc, 0, 0, 0);
            a3[i] = acc;
        }
    }
    // 1x1 (8 -> 1) + sigmoid as ONE chain over the channels: quarter 0 starts it (channels 0-3), quarter 1 finishes it (4-7)
    if (kq == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float o = prm[OFF_B3];
#pragma unroll
            for (int r = 0; r < 4; ++r) o = fmaf(fmaxf(a3[i][r], 0.0f), prm[OFF_W3 + r], o);
            s_part[(wv + 4 * i) * T + n] = o;
        }
    }
    __syncthreads();
    if (kq == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int py = wv + 4 * i;
            float o = s_part[py * T + n];
#pragma unroll
            for (int r = 0; r < 4; ++r) o = fmaf(fmaxf(a3[i][r], 0.0f), prm[OFF_W3 + 4 + r], o);
            const int gy = y0 + py, gx = x0 + n;
            if (gy < h && gx < w) weight[(long)blockIdx.z * h * w + (long)gy * w + gx] = effi_sigmoid(o);
        }
    }
}

}  // namespace

extern "C" int effi_pixelwise_net_f32(const float* entropy, const float* params, int n, int h, int w, float* weight,
                                      effi_stream_t stream) {
    if (!entropy || !params || !weight || n < 1 || n > 65535 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (effi_option(EFFI_OPT_PIXNET_MFMA) == 0)                // A/B switch (effi_set_option): the vector-ALU kernel
        hipLaunchKernelGGL(pixelwise_net_kernel, dim3(effi_cdiv(w, T), effi_cdiv(h, T), n), dim3(256), 0, effi_s(stream),
                           entropy, params, h, w, weight);
    else
        hipLaunchKernelGGL(pixelwise_net_mfma_kernel, dim3(effi_cdiv(w, T), effi_cdiv(h, T), n), dim3(256), 0, effi_s(stream),
                           entropy, params, h, w, weight);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
