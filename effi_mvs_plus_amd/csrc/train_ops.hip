// Scope row n2: the kernels that make the path trainable -- what `loss.backward()` of the
// reference's train.py:229-263 needs below the modules of models/module.py, models/update.py, models/Effi_MVS_plus.py.
//
//   * weight gradients of every convolution on the path (2-D k1/k3/k7, 3-D k3 with stride (1|2, 1|2, 1|2), and the
//     transposed 3-D convolutions) through ONE kernel family: dW[a][b][tap] = sum_o A[a][o] * B[b][o*stride + tap - pad].
//     For a convolution A = grad_out, B = input (torch layout [cout][cin][k..]); for a transposed convolution A = input,
//     B = grad_out ([cin][cout][k..]).  Input gradients reuse the forward kernels with re-arranged weights (host side).
//   * BatchNorm in training mode (batch statistics) forward / backward, with the ReLU that always follows it fused.
//   * the element-wise epilogues of the GRU block and the depth head (activation derivatives, gating) forward / backward.
//   * backward of the 1-D volume lookups (pro_bilinear_sampler / GetCost), of the soft-argmin, of the view-weighted
//     aggregation, of the stage-2/3 warp + correlation and of the convex upsampling.
// Everything is fp32; sums over many elements are accumulated per thread, reduced per workgroup and added with fp32 atomics,
// so the last bits of a gradient vary between runs (as they do in PyTorch's own convolution backward).
#include "common.hpp"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Positions [i0, i1) of the flattened (sample b, position r < n) range of ONE channel, TPB threads striding: the sample index is
// advanced by comparison instead of a 64-bit division per element (the division was most of these kernels' instructions).
#define EFFI_FOR_BATCH_RANGE(i0, i1, n, b, r)                                                                      \
    for (long i_ = (i0) + threadIdx.x, b = i_ / (n), r = i_ - b * (n); i_ < (i1);                                 \
         i_ += TPB, r += TPB, (r >= (n)) ? (b += r / (n), r %= (n)) : 0L)

// ------------------------------------------------------------------------------------------------
// weight gradient, generic: A on the small grid (Da,ha,wa), B on the large one (Db,hb,wb), position o of A meets
// B at o*stride + tap - pad.  grid = (ceil(ca / CAB), cb, strips); every workgroup reduces its strip and adds atomically.
// ------------------------------------------------------------------------------------------------
template <int KD, int KS, int CAB>
__global__ __launch_bounds__(TPB) void wgrad_nd_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int ca,
                                                       int cb_total, int cb_off, int Da, int ha, int wa, int Db, int hb, int wb,
                                                       int sz, int sxy, float* __restrict__ dW) {
    constexpr int TAPS = KD * KS * KS, PZ = KD / 2, PXY = KS / 2;
    const int ca0 = blockIdx.x * CAB, cbi = blockIdx.y;
    const long na = (long)Da * ha * wa, nb = (long)Db * hb * wb;
    const long per = (na + gridDim.z - 1) / gridDim.z;
    const long o0 = blockIdx.z * per, o1 = min(na, o0 + per);
    float acc[CAB][TAPS];
#pragma unroll
    for (int i = 0; i < CAB; ++i)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[i][t] = 0.0f;
    const float* __restrict__ Bc = Bm + (long)cbi * nb;
    for (long o = o0 + threadIdx.x; o < o1; o += TPB) {
        const int x = (int)(o % wa);
        const long t_ = o / wa;
        const int y = (int)(t_ % ha), z = (int)(t_ / ha);
        float a[CAB];
#pragma unroll
        for (int i = 0; i < CAB; ++i) a[i] = (ca0 + i < ca) ? A[(long)(ca0 + i) * na + o] : 0.0f;
#pragma unroll
        for (int kz = 0; kz < KD; ++kz) {
            const int bz = z * sz + kz - PZ;
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int by = y * sxy + ky - PXY;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int bx = x * sxy + kx - PXY;
                    const bool in = (bz >= 0) & (bz < Db) & (by >= 0) & (by < hb) & (bx >= 0) & (bx < wb);
                    const float b = in ? Bc[((long)bz * hb + by) * wb + bx] : 0.0f;
                    const int tap = (kz * KS + ky) * KS + kx;
#pragma unroll
                    for (int i = 0; i < CAB; ++i) acc[i][tap] = fmaf(a[i], b, acc[i][tap]);
                }
            }
        }
    }
    // Reduction over the workgroup's 256 threads of CAB * TAPS values each.  A butterfly (wave_sum) per value costs 12 instructions
    // per value and lane -- as much as 16 positions of the loop above; instead each wave transposes its values through LDS, 32 at a
    // time ([value][lane], rows padded to 65 words), and lane v adds row v: 1 write + 2 reads + 2 adds per value.
    constexpr int NV = CAB * TAPS, CH = 32;
    __shared__ float red[TPB / 64][NV];
    __shared__ float tr[TPB / 64][CH][65];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c0 = 0; c0 < NV; c0 += CH) {
#pragma unroll
        for (int v = 0; v < CH; ++v)
            if (c0 + v < NV) tr[wv][v][lane] = acc[(c0 + v) / TAPS][(c0 + v) % TAPS];
        __syncthreads();
        {
            // lanes 0..31: first half of row (lane & 31), lanes 32..63: second half; the two halves meet through one shuffle
            const int row = lane & 31, j0 = (lane >> 5) * 32;
            float sum = 0.0f;
#pragma unroll
            for (int j = 0; j < 32; ++j) sum += tr[wv][row][j0 + j];
            sum += __shfl_xor(sum, 32);
            if (lane < 32 && c0 + row < NV) red[wv][c0 + row] = sum;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < CAB * TAPS; e += TPB) {
        const int i = e / TAPS, t = e - i * TAPS;
        if (ca0 + i >= ca) continue;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < TPB / 64; ++k) s += red[k][e];
        unsafeAtomicAdd(&dW[((long)(ca0 + i) * cb_total + cb_off + cbi) * TAPS + t], s);
    }
}

// per-channel sums of a [B][C][n] tensor (bias gradients): out[c] = sum_{b,i} g[b][c][i]
__global__ __launch_bounds__(TPB) void channel_sum_kernel(const float* __restrict__ g, int Bn, int C, long n, float* __restrict__ out,
                                                          float* __restrict__ partial) {
    const int c = blockIdx.x;
    const long total = (long)Bn * n;
    const long per = (total + gridDim.y - 1) / gridDim.y;
    const long i0 = blockIdx.y * per, i1 = min(total, i0 + per);
    // double accumulation: bias gradients are sums with heavy cancellation (PixelwiseNet's single output channel: the fp32 sum was
    // 1.9e-2 of the result away from the fp64 one, four times the reference's own fp32 error); the kernel is bound by its loads
    double s = 0.0;
    EFFI_FOR_BATCH_RANGE(i0, i1, n, b, r) s += (double)g[((long)b * C + c) * n + r];
    __shared__ double red[TPB / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (float)(red[0] + red[1] + red[2] + red[3]);
        if (partial) partial[(long)c * gridDim.y + blockIdx.y] = t;
        else out[c] = t;                                    // one workgroup per channel
    }
}

// Second stage of the split reductions: out_k[c] = sum_j partial[(c * nsplit + j) * K + k], j in ascending order -- a fixed order,
// so the result does not depend on which workgroup finished first (with atomics, WHICH activations sit on a ReLU kink differed
// between runs of the same training step).
template <int K>
__global__ void reduce_partials_kernel(const float* __restrict__ partial, int C, int nsplit, float* __restrict__ out0,
                                       float* __restrict__ out1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float s = 0.0f;
        for (int j = 0; j < nsplit; ++j) s += partial[((long)c * nsplit + j) * K + k];
        if (k == 0) out0[c] = s;
        else out1[c] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm, training mode (nn.BatchNorm2d / 3d on batch statistics; models/module.py:148-157,191-200,217-220)
// ------------------------------------------------------------------------------------------------
// out[c] = sum (x - shift[c])^P over batch and positions (P = 1 with shift = nullptr: the sum; P = 2 with shift = mean)
// shift_partial (with P = 2): the first pass' partial sums [C][ns_in] instead of a finished mean -- every workgroup adds them (in
// ascending order: all get the same bits) and divides by N itself, which saves the launch that used to do it; workgroup (c, 0) also
// publishes the mean to mean_out.
template <int P>
__global__ __launch_bounds__(TPB) void bn_moment_kernel(const float* __restrict__ x, int Bn, int C, long n, const float* __restrict__ shift,
                                                        float* __restrict__ out, float* __restrict__ partial,
                                                        const float* __restrict__ shift_partial = nullptr, int ns_in = 0,
                                                        float* __restrict__ mean_out = nullptr) {
    const int c = blockIdx.x;
    const long total = (long)Bn * n;
    const long per = (total + gridDim.y - 1) / gridDim.y;
    const long i0 = blockIdx.y * per, i1 = min(total, i0 + per);
    float sh = shift ? shift[c] : 0.0f;
    if (shift_partial) {
        float m = 0.0f;
        for (int j = 0; j < ns_in; ++j) m += shift_partial[(long)c * ns_in + j];
        sh = m / (float)total;
        if (blockIdx.y == 0 && threadIdx.x == 0) mean_out[c] = sh;
    }
    float s = 0.0f;
    EFFI_FOR_BATCH_RANGE(i0, i1, n, b, r) {
        const float v = x[((long)b * C + c) * n + r] - sh;
        s += (P == 1) ? v : v * v;
    }
    __shared__ float red[TPB / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = red[0] + red[1] + red[2] + red[3];
        if (partial) partial[(long)c * gridDim.y + blockIdx.y] = t;
        else out[c] = t;
    }
}

// Weights of a 2-D convolution [cout][cin][taps] -> the B-operand order of the fp32 matrix-core kernels [kg][tap][nt][k][j]
// (packing.pack_conv2d_mfma: lane k * 16 + j holds W[16 n + j][4 g + k]; rows / columns past the end are zero) + the bias padded to
// 16 nt.  dgrad != 0: the weights of the INPUT-GRADIENT convolution instead (in / out swapped, taps flipped):
// W'[co][ci][t] = W[ci][co][taps - 1 - t] with co < cin, ci < cout.  One launch instead of the seven torch launches of the host
// form, per layer and step (the weights change every step).
__global__ __launch_bounds__(TPB) void pack_conv2d_mfma_kernel(const float* __restrict__ w, const float* __restrict__ bias, int cout, int cin,
                                                               int taps, int dgrad, int nt, int kg, float* __restrict__ wp,
                                                               float* __restrict__ bp) {
    const long total = (long)kg * taps * nt * 64;
    const long e = (long)blockIdx.x * TPB + threadIdx.x;
    if (e < 16 * nt) {
        const int co_n = dgrad ? cin : cout;
        bp[e] = (bias && e < co_n) ? bias[e] : 0.0f;
    }
    if (e >= total) return;
    const int j = (int)(e & 15), k = (int)((e >> 4) & 3);
    long r = e >> 6;
    const int n = (int)(r % nt);
    r /= nt;
    const int t = (int)(r % taps), g = (int)(r / taps);
    const int co = 16 * n + j, ci = 4 * g + k;
    float v = 0.0f;
    if (!dgrad) {
        if (co < cout && ci < cin) v = w[((long)co * cin + ci) * taps + t];
    } else {
        if (co < cin && ci < cout) v = w[((long)ci * cin + co) * taps + (taps - 1 - t)];
    }
    wp[e] = v;
}

// Input gradient of a 5x5 / stride-2 / padding-2 convolution as ONE 3x3 stride-1 convolution over the output gradient g followed by a
// pixel shuffle: the input pixels of parity class (py, px) receive g through the taps ky = py (mod 2), kx = px (mod 2) only, and
//   dx[ci][2 yy + py][2 xx + px] = sum_co sum_{dy,dx in -1..1} g[co][yy + dy][xx + dx] K[(ci,py,px)][co][dy + 1][dx + 1],
//   K[(ci,py,px)][co][dy + 1][dx + 1] = W[co][ci][py + 2 - 2 dy][px + 2 - 2 dx]   (zero where that tap does not exist).
// This kernel writes K in the matrix-core operand order (pack_conv2d_mfma_kernel) for the output channels (ci, py, px) = 4 ci + 2 py + px,
// ci in [ci_off, ci_off + ci_n): the 3x3 kernel then does the work at matrix-core rate (the direct vector form took 226 us per call).
__global__ __launch_bounds__(TPB) void pack_k5s2_dgrad_kernel(const float* __restrict__ w, int cout, int cin, int ci_off, int ci_n, int nt, int kg,
                                                              float* __restrict__ wp, float* __restrict__ bp) {
    const long total = (long)kg * 9 * nt * 64;
    const long e = (long)blockIdx.x * TPB + threadIdx.x;
    if (e < 16 * nt) bp[e] = 0.0f;
    if (e >= total) return;
    const int j = (int)(e & 15), k = (int)((e >> 4) & 3);
    long r = e >> 6;
    const int n = (int)(r % nt);
    r /= nt;
    const int t = (int)(r % 9), g = (int)(r / 9);
    const int cq = 16 * n + j, co = 4 * g + k;                 // output channel (ci, py, px) of the 3x3 kernel; its input channel = co
    float v = 0.0f;
    if (cq < 4 * ci_n && co < cout) {
        const int ci = ci_off + (cq >> 2), py = (cq >> 1) & 1, px = cq & 1;
        const int ky = py + 2 - 2 * (t / 3 - 1), kx = px + 2 - 2 * (t % 3 - 1);
        if (ky >= 0 && ky < 5 && kx >= 0 && kx < 5) v = w[(((long)co * cin + ci) * 5 + ky) * 5 + kx];
    }
    wp[e] = v;
}

// y = (x - mean) * invstd * gamma + beta, then ReLU if asked
struct BnFinish {                // second half of the statistics, done by the apply kernel (see effi_bn_train_fwd_f32); var_partial == NULL: off
    const float* var_partial; int ns; float n_total, eps, momentum, unbias;
    float* invstd_out; float* running_mean; float* running_var; long long* n_tracked;
};
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const float* __restrict__ x, int Bn, int C, long n, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int relu, float* __restrict__ y, const BnFinish f) {
    // grid (chunks of a plane, planes = Bn * C): the channel is a workgroup constant (no division per element)
    const int c = (int)(blockIdx.y % C);
    float is;
    if (f.var_partial) {
        // variance from the second pass' partial sums (every workgroup of the channel, same order, same bits); the first workgroup
        // of the channel publishes invstd and updates the running statistics (nn.BatchNorm: unbiased variance, momentum)
        float sq = 0.0f;
        for (int j = 0; j < f.ns; ++j) sq += f.var_partial[(long)c * f.ns + j];
        const float var = sq / f.n_total;
        is = rsqrtf(var + f.eps);
        if (blockIdx.x == 0 && blockIdx.y == (unsigned)c && threadIdx.x == 0) {
            f.invstd_out[c] = is;
            if (f.running_mean) {
                const float keep = 1.0f - f.momentum;
                f.running_mean[c] = f.running_mean[c] * keep + f.momentum * mean[c];
                f.running_var[c] = f.running_var[c] * keep + f.momentum * (var * f.unbias);
            }
            if (c == 0 && f.n_tracked) f.n_tracked[0] += 1;
        }
    } else {
        is = invstd[c];
    }
    const float mu = mean[c], ga = gamma[c], be = beta[c];
    const long base = (long)blockIdx.y * n;
    for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < n; r += (long)gridDim.x * TPB) {
        float v = (x[base + r] - mu) * is * ga + be;
        if (relu) v = fmaxf(v, 0.0f);
        y[base + r] = v;
    }
}

// s1[c] = sum g', s2[c] = sum g' * xhat, with g' = gy masked by the ReLU (y > 0) and xhat = (x - mean) * invstd.
// Both sums are accumulated in DOUBLE (threads, waves, workgroup partials): gx = gamma * invstd * (g' - s1/N - xhat * s2/N) below is a
// projection that cancels most of g' -- on a 8 x 148 x 200 volume a relative 1e-6 on s1 / N is as large as the small entries of gx, and
// the weight gradients of the layers behind it inherit that (round 4; the products g' * xhat themselves stay fp32, as in torch).
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                            const float* __restrict__ x, int Bn, int C, long n,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd, int relu,
                                                            float* __restrict__ s1, float* __restrict__ s2, double* __restrict__ partial) {
    const int c = blockIdx.x;
    const long total = (long)Bn * n;
    const long per = (total + gridDim.y - 1) / gridDim.y;
    const long i0 = blockIdx.y * per, i1 = min(total, i0 + per);
    const float mu = mean[c], is = invstd[c];
    double a = 0.0, b2 = 0.0;
    EFFI_FOR_BATCH_RANGE(i0, i1, n, b, r) {
        const long e = ((long)b * C + c) * n + r;
        float g = gy[e];
        if (relu && !(y[e] > 0.0f)) g = 0.0f;
        a += (double)g;
        b2 += (double)(g * ((x[e] - mu) * is));
    }
    __shared__ double red[2][TPB / 64];
    a = wave_sum(a);
    b2 = wave_sum(b2);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t1 = red[0][0] + red[0][1] + red[0][2] + red[0][3], t2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        if (partial) {
            partial[((long)c * gridDim.y + blockIdx.y) * 2 + 0] = t1;
            partial[((long)c * gridDim.y + blockIdx.y) * 2 + 1] = t2;
        } else {
            s1[c] = (float)t1;
            s2[c] = (float)t2;
        }
    }
}

// gx = gamma * invstd * (g' - s1/N - xhat * s2/N)
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                           const float* __restrict__ x, int Bn, int C, long n,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, float* __restrict__ s1,
                                                           float* __restrict__ s2, int relu, float* __restrict__ gx,
                                                           const double* __restrict__ partial, int ns) {
    const double N = (double)((long)Bn * n);
    const int c = (int)(blockIdx.y % C);                       // grid (chunks of a plane, planes = Bn * C)
    double t1, t2;
    if (partial) {        // the reduce kernel's partial sums [C][ns][2] (double): added here (ascending order), published by the channel's first workgroup
        t1 = t2 = 0.0;
        for (int j = 0; j < ns; ++j) {
            t1 += partial[((long)c * ns + j) * 2 + 0];
            t2 += partial[((long)c * ns + j) * 2 + 1];
        }
        if (blockIdx.x == 0 && blockIdx.y == (unsigned)c && threadIdx.x == 0) { s1[c] = (float)t1; s2[c] = (float)t2; }
    } else {
        t1 = (double)s1[c];
        t2 = (double)s2[c];
    }
    const float mu = mean[c], is = invstd[c], ga = gamma[c], a1 = (float)(t1 / N), a2 = (float)(t2 / N);
    const long base = (long)blockIdx.y * n;
    for (long r = (long)blockIdx.x * TPB + threadIdx.x; r < n; r += (long)gridDim.x * TPB) {
        const long i = base + r;
        float g = gy[i];
        if (relu && !(y[i] > 0.0f)) g = 0.0f;
        const float xh = (x[i] - mu) * is;
        gx[i] = ga * is * (g - a1 - xh * a2);
    }
}

// ------------------------------------------------------------------------------------------------
// element-wise pieces of the GRU block / heads (models/update.py:20-27,40-49,86-99; Effi_MVS_plus.py:138-148)
// ------------------------------------------------------------------------------------------------
struct PwArgs {
    const float* a; const float* b; const float* c; const float* d;
    float* o0; float* o1; float* o2;
    float s0, s1;
    long n, inner;       // inner: elements per channel (EFFI_PW_SCALE_CH)
    int C;
};

__global__ __launch_bounds__(TPB) void pointwise_kernel(int op, PwArgs p) {
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < p.n; i += (long)gridDim.x * TPB) {
        switch (op) {
            case EFFI_PW_ACT_BWD_RELU: p.o0[i] = (p.b[i] > 0.0f) ? p.a[i] : 0.0f; break;                      // a = g, b = y
            case EFFI_PW_ACT_BWD_SIGMOID: { const float y = p.b[i]; p.o0[i] = p.a[i] * (y * (1.0f - y)); } break;
            case EFFI_PW_ACT_BWD_TANH: { const float y = p.b[i]; p.o0[i] = p.a[i] * (1.0f - y * y); } break;
            case EFFI_PW_TANH: p.o0[i] = tanhf(p.a[i]); break;
            case EFFI_PW_RELU: p.o0[i] = fmaxf(p.a[i], 0.0f); break;
            case EFFI_PW_SIGMOID: p.o0[i] = effi_sigmoid(p.a[i]); break;
            case EFFI_PW_MUL: p.o0[i] = p.a[i] * p.b[i]; break;
            case EFFI_PW_MUL_BWD: { const float g = p.a[i]; p.o0[i] = g * p.c[i]; p.o1[i] = g * p.b[i]; } break;   // b = x, c = y
            case EFFI_PW_GRU: { const float z = p.a[i]; p.o0[i] = (1.0f - z) * p.b[i] + z * p.c[i]; } break;       // a = z, b = h, c = q
            case EFFI_PW_GRU_BWD: {                                                                                // a = g, b = z, c = h, d = q
                const float g = p.a[i], z = p.b[i];
                p.o0[i] = g * (p.d[i] - p.c[i]);      // d/dz
                p.o1[i] = g * (1.0f - z);             // d/dh
                p.o2[i] = g * z;                      // d/dq
            } break;
            case EFFI_PW_INV_TO_DEPTH: p.o0[i] = effi_inv_to_depth(p.a[i], p.s0, p.s1); break;                      // s0 = lo, s1 = hi
            case EFFI_PW_INV_TO_DEPTH_BWD: {                                                                       // a = g, b = inv
                const float max_depth = 1.0f / p.s0, min_depth = 1.0f / p.s1;
                const float min_disp = 1.0f / max_depth, max_disp = 1.0f / min_depth;
                const float s = min_disp + (max_disp - min_disp) * p.b[i];
                p.o0[i] = (s > 1e-4f) ? -p.a[i] * (max_disp - min_disp) / (s * s) : 0.0f;
            } break;
            case EFFI_PW_SCALE_CH: p.o0[i] = p.a[i] * p.b[(i / p.inner) % p.C]; break;                              // b = per-channel factor
            default: break;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 1-D lookup backward (pro_bilinear_sampler, models/Effi_MVS_plus.py:102-134): the query coordinates carry no gradient on
// this path (they come from detached depths); the looked-up vector receives g * (w0, w1) at (x0, x0 + 1).
// One thread owns one pixel's D-vector, so the accumulation is a plain read-modify-write.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lookup1d_bwd(float* __restrict__ gvol, long dstride, int Dp, float q_depth, float dmin, float dmax,
                                             float g) {
    // the forward's position and weights (common.hpp: one arithmetic for both passes)
    int x0;
    float w0, w1;
    lookup1d_index(Dp, q_depth, dmin, dmax, x0, w0, w1);
    if (x0 >= 0 && x0 <= Dp - 1) gvol[x0 * dstride] += g * w0;
    if (x0 + 1 >= 0 && x0 + 1 <= Dp - 1) gvol[(x0 + 1) * dstride] += g * w1;
}

__global__ void vol_lookup1d_bwd_kernel(float* __restrict__ gvol, long vds, long vps, int Dp, const float* __restrict__ query,
                                        long qds, long qys, long qxs, int nq, const float* __restrict__ dmin,
                                        const float* __restrict__ dmax, long rps, int h, int w, const float* __restrict__ gout) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    const float lo = dmin[p * rps], hi = dmax[p * rps];
    for (int k = 0; k < nq; ++k)
        lookup1d_bwd(gvol + p * vps, vds, Dp, query[k * qds + y * qys + x * qxs], lo, hi, gout[(long)k * h * w + p]);
}

// GetCost backward (models/Effi_MVS_plus.py:257-303): gcost [2*nq][hw] -> gcur [Dcur][hw], greg [Dreg][hw] (accumulated)
__global__ void getcost_bwd_kernel(const float* __restrict__ inv_depth, const float* __restrict__ disp_range, int n_range,
                                   int input_is_depth, const float* __restrict__ interval, float* __restrict__ gcur, long cds,
                                   long cps, int Dcur, float* __restrict__ greg, long rds, long rps_, int Dreg,
                                   const float* __restrict__ dmin, const float* __restrict__ dmax, long range_ps, int nq, int hw,
                                   const float* __restrict__ gcost) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    const float itv = interval[0];
    float depth = inv_depth[p];
    if (!input_is_depth) depth = effi_inv_to_depth(depth, disp_range[0], disp_range[n_range - 1]);
    const float dv = effi_lk_rcp(depth);                      // as effi_getcost_pixel (common.hpp)
    const float half = (float)(nq / 2) * itv;
    const float smin = fmaxf(dv - half, 1e-4f);
    const float smax = fminf(fmaxf(dv + half, 1e-4f), 1e4f);
    const float step = (smax - smin) / (float)(nq - 1);
    const float rlo = dmin[p * range_ps], rhi = dmax[p * range_ps];
    for (int k = 0; k < nq; ++k) {
        const float s = fmaxf(smin + (float)k * step, 1e-5f);
        const float qd = effi_lk_rcp(s);
        lookup1d_bwd(gcur + p * cps, cds, Dcur, qd, rlo, rhi, gcost[(long)k * hw + p]);
        lookup1d_bwd(greg + p * rps_, rds, Dreg, qd, rlo, rhi, gcost[(long)(nq + k) * hw + p]);
    }
}

// soft-argmin backward (models/Effi_MVS_plus.py:79-81): depth = sum_d p_d * hyp_d, p = softmax(logits)
//   d depth / d logit_d = p_d * (hyp_d - depth)
__global__ void softargmin_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ hyp, long dds, long dps, int D,
                                      int hw, const float* __restrict__ gdepth, float* __restrict__ glogits) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    float m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, logits[(long)d * hw + p]);
    float sum = 0.0f;
    for (int d = 0; d < D; ++d) sum += expf(logits[(long)d * hw + p] - m);
    // glogits_d = g p_d (hyp_d - E[hyp]): hypotheses of 425..935 mm a few mm apart -- the difference cancels 2-3 digits, so the expectation
    // and the difference are formed in double (the probabilities themselves are the forward's fp32 values)
    double dep = 0.0;
    for (int d = 0; d < D; ++d) dep += (double)(expf(logits[(long)d * hw + p] - m) / sum) * (double)hyp[d * dds + p * dps];
    const float g = gdepth[p];
    for (int d = 0; d < D; ++d) {
        const float pr = expf(logits[(long)d * hw + p] - m) / sum;
        glogits[(long)d * hw + p] = (float)((double)g * (double)pr * ((double)hyp[d * dds + p * dps] - dep));
    }
}

// view-weighted aggregation backward (models/Effi_MVS_plus.py:48-53,67): out_d = sum_v s_vd w_v / (sum_v w_v + 1e-6)
//   g s_vd = g_d w_v / den;   g w_v = sum_d g_d (s_vd - out_d) / den
__global__ void view_aggregate_bwd_kernel(const float* __restrict__ sim_views, const float* __restrict__ weights, int S, int D, int hw,
                                          const float* __restrict__ gout, float* __restrict__ gsim, float* __restrict__ gw) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= hw) return;
    float wsum = 0.0f;
    for (int v = 0; v < S; ++v) wsum += weights[(long)v * hw + p];
    const float den = wsum + 1e-6f;
    // g w_v = sum_d g_d (s_vd - out_d) / den is a sum of differences that mostly cancel (every view sees nearly the same similarity);
    // the view-weight net's gradients are sums of it over all pixels.  Aggregate, difference and sum in double, one rounding at the end.
    double gwa[EFFI_MAX_VIEWS];                  // constant trip counts below: stays in registers
#pragma unroll
    for (int v = 0; v < EFFI_MAX_VIEWS; ++v) gwa[v] = 0.0;
    for (int d = 0; d < D; ++d) {
        double acc = 0.0;
        for (int v = 0; v < S; ++v) acc += (double)sim_views[((long)v * D + d) * hw + p] * (double)weights[(long)v * hw + p];
        const double o = acc / (double)den;
        const float g = gout[(long)d * hw + p];
#pragma unroll
        for (int v = 0; v < EFFI_MAX_VIEWS; ++v) {
            if (v < S) {
                const float s = sim_views[((long)v * D + d) * hw + p];
                gsim[((long)v * D + d) * hw + p] = g * weights[(long)v * hw + p] / den;
                gwa[v] += (double)g * ((double)s - o) / (double)den;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < EFFI_MAX_VIEWS; ++v)
        if (v < S) gw[(long)v * hw + p] = (float)gwa[v];
}

// convex upsampling backward (models/Effi_MVS_plus.py:167-178), ratio 2: up[2y+i][2x+j] = sum_k softmax_k(mask[k][i][j]) * nb_k,
// nb_k = inv at (y + k/3 - 1, x + k%3 - 1) (zero outside).  gmask is written; the contribution of pixel p to its neighbour k goes to
// contrib[k][p], and convex_upsample2x_gather_kernel sums, for every pixel, the nine contributions addressed to it in a fixed order.
__global__ void convex_upsample2x_bwd_kernel(const float* __restrict__ inv, const float* __restrict__ mask, int h, int w,
                                             const float* __restrict__ gup, float* __restrict__ gmask, float* __restrict__ contrib) {
    const int p = blockIdx.x * TPB + threadIdx.x;
    if (p >= h * w) return;
    const int y = p / w, x = p - y * w;
    const long hw = (long)h * w;
    float nb[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        nb[k] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? inv[(long)yy * w + xx] : 0.0f;
    }
    float gn[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) gn[k] = 0.0f;
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
        const int i = ij >> 1, j = ij & 1;
        float m[9], mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) {                       // mask channel = k*4 + i*2 + j  (view(N,1,9,r,r,H,W))
            m[k] = mask[(long)(k * 4 + ij) * hw + p];
            mx = fmaxf(mx, m[k]);
        }
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); sum += m[k]; }
        float up = 0.0f;
#pragma unroll
        for (int k = 0; k < 9; ++k) { m[k] = m[k] / sum; up += m[k] * nb[k]; }
        const float g = gup[(long)(2 * y + i) * (2 * w) + 2 * x + j];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            gmask[(long)(k * 4 + ij) * hw + p] = g * m[k] * (nb[k] - up);
            gn[k] += g * m[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) contrib[(long)k * hw + p] = gn[k];
}

__global__ void convex_upsample2x_gather_kernel(const float* __restrict__ contrib, int h, int w, float* __restrict__ ginv) {
    const int q = blockIdx.x * TPB + threadIdx.x;
    if (q >= h * w) return;
    const int y = q / w, x = q - y * w;
    const long hw = (long)h * w;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {                       // pixel (y - dy, x - dx) names this pixel as its neighbour k = (dy + 1) * 3 + dx + 1
        const int yy = y - (k / 3 - 1), xx = x - (k % 3 - 1);
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) acc += contrib[(long)k * hw + (long)yy * w + xx];
    }
    ginv[q] = acc;
}

}  // namespace

// ================================================================================================
extern "C" int effi_conv_wgrad_f32(const float* a, const float* b, int ca, int cb, int cb_total, int cb_off, int kd, int ks, int Da,
                                   int ha, int wa, int Db, int hb, int wb, int sz, int sxy, float* dw, effi_stream_t stream) {
    if (!a || !b || !dw || ca < 1 || cb < 1 || cb_off < 0 || cb_off + cb > cb_total) return EFFI_ERR_BADARG;
    if (Da < 1 || ha < 1 || wa < 1 || Db < 1 || hb < 1 || wb < 1 || sz < 1 || sz > 2 || sxy < 1 || sxy > 2) return EFFI_ERR_BADARG;
    hipStream_t s = effi_s(stream);
    const long na = (long)Da * ha * wa;
    // positions per thread: ~32 (the end-of-workgroup reduction is amortised over them), fewer when that leaves the chip short of
    // workgroups (small maps / few channel blocks)
    auto strips_for = [&](int cab) {
        const long blocks = (long)((ca + cab - 1) / cab) * cb;
        long st = max(1L, min(64L, na / 8192));
        while (blocks * st < 768 && na / st > 2048 && st < 64) st *= 2;
        return (int)st;
    };
#define EFFI_WG(KD_, KS_, CAB_)                                                                                            \
    hipLaunchKernelGGL((wgrad_nd_kernel<KD_, KS_, CAB_>), dim3((ca + CAB_ - 1) / CAB_, cb, strips_for(CAB_)), dim3(TPB), 0, s, a, b, ca, \
                       cb_total, cb_off, Da, ha, wa, Db, hb, wb, sz, sxy, dw)
    if (kd == 1 && ks == 1) EFFI_WG(1, 1, 16);
    else if (kd == 1 && ks == 3) EFFI_WG(1, 3, 8);
    else if (kd == 1 && ks == 5) EFFI_WG(1, 5, 4);
    else if (kd == 1 && ks == 7) EFFI_WG(1, 7, 2);
    else if (kd == 3 && ks == 3) EFFI_WG(3, 3, 4);
    else return EFFI_ERR_UNSUPPORTED;
#undef EFFI_WG
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

// Split reductions: nsplit workgroups per channel write partial sums to ``scratch`` ([C][nsplit][K] floats, caller-owned) and a second
// tiny launch adds them in a fixed order; scratch == NULL or nsplit <= 1: one workgroup per channel.  Either way the order of the
// additions is fixed: a training step's BatchNorm statistics are bitwise repeatable.
// grid of the per-plane element-wise kernels: (chunks of a plane, planes); ~4 elements per thread, at most ~16 K workgroups in all
static dim3 effi_plane_grid(long n, int planes) {
    long gx = (n + 4 * TPB - 1) / (4 * TPB);
    const long cap = max(1L, 16384L / planes);
    if (gx > cap) gx = cap;
    return dim3((unsigned)gx, (unsigned)planes);
}

static int effi_nsplit(const float* scratch, int nsplit) { return (scratch && nsplit > 1) ? (nsplit > 1024 ? 1024 : nsplit) : 1; }

extern "C" int effi_channel_sum_f32(const float* g, int B, int C, long n, float* out, float* scratch, int nsplit, effi_stream_t stream) {
    if (!g || !out || B < 1 || C < 1 || n < 1) return EFFI_ERR_BADARG;
    const int ns = effi_nsplit(scratch, nsplit);
    hipStream_t s = effi_s(stream);
    hipLaunchKernelGGL(channel_sum_kernel, dim3(C, ns), dim3(TPB), 0, s, g, B, C, n, out, ns > 1 ? scratch : nullptr);
    if (ns > 1) hipLaunchKernelGGL(reduce_partials_kernel<1>, dim3((C + 63) / 64), dim3(64), 0, s, scratch, C, ns, out, out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_bn_moment_f32(const float* x, int B, int C, long n, const float* shift, int power, float* out, float* scratch,
                                  int nsplit, effi_stream_t stream) {
    if (!x || !out || B < 1 || C < 1 || n < 1 || (power != 1 && power != 2)) return EFFI_ERR_BADARG;
    const int ns = effi_nsplit(scratch, nsplit);
    hipStream_t s = effi_s(stream);
    const dim3 grid(C, ns);
    float* part = ns > 1 ? scratch : nullptr;
    if (power == 1) hipLaunchKernelGGL(bn_moment_kernel<1>, grid, dim3(TPB), 0, s, x, B, C, n, shift, out, part);
    else hipLaunchKernelGGL(bn_moment_kernel<2>, grid, dim3(TPB), 0, s, x, B, C, n, shift, out, part);
    if (ns > 1) hipLaunchKernelGGL(reduce_partials_kernel<1>, dim3((C + 63) / 64), dim3(64), 0, s, scratch, C, ns, out, out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

// nn.BatchNorm in training mode, forward, as ONE entry (five launches: sum, mean, centred squares, variance + running statistics,
// apply): what ops.bn_moments + torch arithmetic + ops.bn_apply did in ~17 launches.  running_mean / running_var / n_tracked may be
// NULL (no tracking); scratch: 2 x [C][nsplit] floats (always used: nsplit >= 1).
extern "C" int effi_bn_train_fwd_f32(const float* x, int B, int C, long n, const float* gamma, const float* beta, float eps, float momentum,
                                     float* running_mean, float* running_var, long long* n_tracked, int relu, float* y, float* mean,
                                     float* invstd, float* scratch, int nsplit, effi_stream_t stream) {
    if (!x || !gamma || !beta || !y || !mean || !invstd || !scratch || B < 1 || C < 1 || n < 1 || nsplit < 1) return EFFI_ERR_BADARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EFFI_ERR_BADARG;
    if ((long)B * C > 65535) return EFFI_ERR_UNSUPPORTED;
    const int ns = nsplit > 1024 ? 1024 : nsplit;
    hipStream_t s = effi_s(stream);
    const dim3 grid(C, ns);
    const long N = (long)B * n;
    const float nf = (float)N, unbias = (float)((double)N / (double)(N > 1 ? N - 1 : 1));
    // three launches: sums -> centred squares (each workgroup finishes the mean from the partial sums itself) -> apply (each
    // workgroup finishes the variance itself; the first one of a channel writes invstd and the running statistics)
    float* part1 = scratch;
    float* part2 = scratch + (long)C * ns;
    hipLaunchKernelGGL(bn_moment_kernel<1>, grid, dim3(TPB), 0, s, x, B, C, n, (const float*)nullptr, mean, part1, (const float*)nullptr, 0,
                       (float*)nullptr);
    hipLaunchKernelGGL(bn_moment_kernel<2>, grid, dim3(TPB), 0, s, x, B, C, n, (const float*)nullptr, invstd, part2, (const float*)part1, ns,
                       mean);
    const BnFinish fin{part2, ns, nf, eps, momentum, unbias, invstd, running_mean, running_var, n_tracked};
    hipLaunchKernelGGL(bn_apply_kernel, effi_plane_grid(n, B * C), dim3(TPB), 0, s, x, B, C, n, mean, invstd, gamma, beta, relu, y, fin);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_pack_conv2d_mfma_f32(const float* weight, const float* bias, int cout, int cin, int ks, int dgrad, float* wpack,
                                         float* bias_pack, effi_stream_t stream) {
    if (!weight || !wpack || !bias_pack || cout < 1 || cin < 1 || ks < 1 || ks > 7) return EFFI_ERR_BADARG;
    const int co = dgrad ? cin : cout, ci = dgrad ? cout : cin;
    const int nt = (co + 15) / 16, kg = (ci + 3) / 4, taps = ks * ks;
    const long total = (long)kg * taps * nt * 64;
    hipLaunchKernelGGL(pack_conv2d_mfma_kernel, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, effi_s(stream), weight, bias, cout, cin,
                       taps, dgrad, nt, kg, wpack, bias_pack);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_pack_conv2d_k5s2_dgrad_f32(const float* weight, int cout, int cin, int ci_off, int ci_n, float* wpack, float* bias_pack,
                                               effi_stream_t stream) {
    if (!weight || !wpack || !bias_pack || cout < 1 || cin < 1 || ci_off < 0 || ci_n < 1 || ci_off + ci_n > cin) return EFFI_ERR_BADARG;
    const int nt = (4 * ci_n + 15) / 16, kg = (cout + 3) / 4;
    const long total = (long)kg * 9 * nt * 64;
    hipLaunchKernelGGL(pack_k5s2_dgrad_kernel, dim3((unsigned)((total + TPB - 1) / TPB)), dim3(TPB), 0, effi_s(stream), weight, cout, cin, ci_off,
                       ci_n, nt, kg, wpack, bias_pack);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_bn_apply_f32(const float* x, int B, int C, long n, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, int relu, float* y, effi_stream_t stream) {
    if (!x || !mean || !invstd || !gamma || !beta || !y || B < 1 || C < 1 || n < 1) return EFFI_ERR_BADARG;
    if ((long)B * C > 65535) return EFFI_ERR_UNSUPPORTED;
    const BnFinish off{nullptr, 0, 0.0f, 0.0f, 0.0f, 0.0f, nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(bn_apply_kernel, effi_plane_grid(n, B * C), dim3(TPB), 0, effi_s(stream), x, B, C, n, mean, invstd, gamma, beta, relu, y, off);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_bn_bwd_f32(const float* gy, const float* y, const float* x, int B, int C, long n, const float* mean,
                               const float* invstd, const float* gamma, int relu, float* s1, float* s2, float* gx,
                               float* scratch, int nsplit, effi_stream_t stream) {
    if (!gy || !y || !x || !mean || !invstd || !gamma || !s1 || !s2 || !gx || B < 1 || C < 1 || n < 1) return EFFI_ERR_BADARG;
    if ((long)B * C > 65535) return EFFI_ERR_UNSUPPORTED;
    hipStream_t s = effi_s(stream);
    const int ns = effi_nsplit(scratch, nsplit);
    // scratch: [C][ns][2] DOUBLES (8-byte aligned; the caller sizes it as 4 floats per (channel, split))
    if (ns > 1 && (reinterpret_cast<uintptr_t>(scratch) & 7)) return EFFI_ERR_BADARG;
    double* part = ns > 1 ? reinterpret_cast<double*>(scratch) : nullptr;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, ns), dim3(TPB), 0, s, gy, y, x, B, C, n, mean, invstd, relu, s1, s2, part);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, effi_plane_grid(n, B * C), dim3(TPB), 0, s, gy, y, x, B, C, n, mean, invstd, gamma, s1, s2, relu, gx,
                       (const double*)part, ns);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_pointwise_f32(int op, const float* a, const float* b, const float* c, const float* d, float s0, float s1, long n,
                                  long inner, int C, float* o0, float* o1, float* o2, effi_stream_t stream) {
    if (!a || !o0 || n < 1 || op < 0 || op >= EFFI_PW_COUNT) return EFFI_ERR_BADARG;
    PwArgs p{a, b, c, d, o0, o1, o2, s0, s1, n, inner > 0 ? inner : 1, C > 0 ? C : 1};
    hipLaunchKernelGGL(pointwise_kernel, dim3((unsigned)min((n + TPB - 1) / TPB, 16384L)), dim3(TPB), 0, effi_s(stream), op, p);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_vol_lookup1d_bwd_f32(float* gvol, long vds, long vps, int Dp, const float* query, long qds, long qys, long qxs,
                                         int nq, const float* dmin, const float* dmax, long rps, int h, int w, const float* gout,
                                         effi_stream_t stream) {
    if (!gvol || !query || !dmin || !dmax || !gout || Dp < 2 || nq < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(vol_lookup1d_bwd_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, effi_s(stream), gvol, vds, vps, Dp, query,
                       qds, qys, qxs, nq, dmin, dmax, rps, h, w, gout);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_getcost_bwd_f32(const float* inv_depth, const float* disp_range, int n_range, int input_is_depth,
                                    const float* interval, float* gcur, long cds, long cps, int Dcur, float* greg, long rds, long rps,
                                    int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h, int w,
                                    const float* gcost, effi_stream_t stream) {
    if (!inv_depth || !interval || !gcur || !greg || !dmin || !dmax || !gcost || nq < 2 || h < 1 || w < 1 || Dcur < 2 || Dreg < 2)
        return EFFI_ERR_BADARG;
    if (!input_is_depth && (!disp_range || n_range < 2)) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(getcost_bwd_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, effi_s(stream), inv_depth, disp_range, n_range,
                       input_is_depth, interval, gcur, cds, cps, Dcur, greg, rds, rps, Dreg, dmin, dmax, range_ps, nq, h * w, gcost);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_softargmin_bwd_f32(const float* logits, const float* hyp, long dds, long dps, int D, int hw, const float* gdepth,
                                       float* glogits, effi_stream_t stream) {
    if (!logits || !hyp || !gdepth || !glogits || D < 1 || hw < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(softargmin_bwd_kernel, dim3(effi_cdiv(hw, TPB)), dim3(TPB), 0, effi_s(stream), logits, hyp, dds, dps, D, hw, gdepth,
                       glogits);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_view_aggregate_bwd_f32(const float* sim_views, const float* weights, int S, int D, int hw, const float* gout,
                                           float* gsim, float* gw, effi_stream_t stream) {
    if (!sim_views || !weights || !gout || !gsim || !gw || S < 1 || S > EFFI_MAX_VIEWS || D < 1 || hw < 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(view_aggregate_bwd_kernel, dim3(effi_cdiv(hw, TPB)), dim3(TPB), 0, effi_s(stream), sim_views, weights, S, D, hw,
                       gout, gsim, gw);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_convex_upsample2x_bwd_f32(const float* inv_depth, const float* mask, int h, int w, const float* gup, float* gmask,
                                              float* ginv, float* scratch9, effi_stream_t stream) {
    if (!inv_depth || !mask || !gup || !gmask || !ginv || !scratch9 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    hipLaunchKernelGGL(convex_upsample2x_bwd_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, st, inv_depth, mask, h,
                       w, gup, gmask, scratch9);
    hipLaunchKernelGGL(convex_upsample2x_gather_kernel, dim3(effi_cdiv((long)h * w, TPB)), dim3(TPB), 0, st, scratch9, h, w, ginv);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

// ------------------------------------------------------------------------------------------------
// Input gradient of the feature pyramid's 5x5 / stride-2 / padding-2 convolutions (models/module.py:32-75,376-388):
//   dx[ci][y][x] = sum_co sum_{ky,kx} g[co][(y + 2 - ky) / 2][(x + 2 - kx) / 2] * W[co][ci][ky][kx]   over the taps whose
//   (y + 2 - ky), (x + 2 - kx) are even and land inside g.
// One thread per input pixel and block of 8 input channels; taps of the wrong parity are skipped by stepping ky, kx by 2.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void conv2d_k5s2_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ wgt, int cin,
                                                                int cout, int hin, int win, int ho, int wo, float* __restrict__ dx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hin * win) return;
    const int y = p / win, x = p - y * win;
    const int c0 = blockIdx.y * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    for (int co = 0; co < cout; ++co) {
        const float* __restrict__ gc = g + (long)co * ho * wo;
        const float* __restrict__ wc = wgt + ((long)co * cin + c0) * 25;
        for (int ky = y & 1; ky < 5; ky += 2) {
            const int oy = (y + 2 - ky) >> 1;
            if ((y + 2 - ky) < 0 || oy >= ho) continue;
            for (int kx = x & 1; kx < 5; kx += 2) {
                const int ox = (x + 2 - kx) >> 1;
                if ((x + 2 - kx) < 0 || ox >= wo) continue;
                const float gv = gc[(long)oy * wo + ox];
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c0 + i < cin) acc[i] = fmaf(gv, wc[i * 25 + ky * 5 + kx], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (c0 + i < cin) dx[(long)(c0 + i) * hin * win + p] = acc[i];
}
}  // namespace

extern "C" int effi_conv2d_k5s2_dgrad_f32(const float* grad_out, const float* weight, int cin, int cout, int hin, int win, float* grad_in,
                                          effi_stream_t stream) {
    if (!grad_out || !weight || !grad_in || cin < 1 || cout < 1 || hin < 1 || win < 1) return EFFI_ERR_BADARG;
    const int ho = (hin - 1) / 2 + 1, wo = (win - 1) / 2 + 1;
    hipLaunchKernelGGL(conv2d_k5s2_dgrad_kernel, dim3(effi_cdiv((long)hin * win, 256), (cin + 7) / 8), dim3(256), 0, effi_s(stream), grad_out,
                       weight, cin, cout, hin, win, ho, wo, grad_in);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
