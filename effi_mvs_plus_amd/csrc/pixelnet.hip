// K4: view-weight net (models/Effi_MVS_plus.py:361-362): three 3x3 conv+BN+ReLU layers
// (1->16->16->8), a 1x1 conv (8->1) and a sigmoid, fused into one kernel.  A 256-thread block owns a
// 16x16 output tile; the 22x22 input patch and both intermediate activations live in LDS (48 KiB), so
// the only HBM traffic is the entropy map in and the weight map out.  Weights are read through the
// scalar cache (uniform addresses) and feed v_fma as SGPR operands.
//
// params (host-packed, BN folded): w0[9][16] b0[16] | w1[16][9][16] b1[16] | w2[16][9][8] b2[8] | w3[8] b3[1]
// i.e. every 3x3 layer is stored [cin][tap][cout].
#include "common.hpp"

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

}  // namespace

extern "C" int effi_pixelwise_net_f32(const float* entropy, const float* params, int n, int h, int w, float* weight,
                                      effi_stream_t stream) {
    if (!entropy || !params || !weight || n < 1 || n > 65535 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(pixelwise_net_kernel, dim3(effi_cdiv(w, T), effi_cdiv(h, T), n), dim3(256), 0, effi_s(stream),
                       entropy, params, h, w, weight);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
