// K11: projection algebra on the device (no host round trip, graph-capture safe).
// Reference: models/Effi_MVS_plus.py:34-37,217-220 (K.[R|t]) and models/module.py:314-316
// (src_proj . inverse(ref_proj)).  The reference does this in fp32 with LAPACK's LU; here the
// 4x4 algebra runs in fp64 and is rounded to fp32 once, which is at least as close to the exact
// value as the reference's own result ("numerics").
#include "common.hpp"

namespace {

__device__ void compose_k_rt(const float* pair, double P[4][4]) {
    // pair: [2][4][4]; P = extrinsic with rows 0..2 replaced by K[:3,:3] . E[:3,:4]
    const float* E = pair;
    const float* K = pair + 16;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += (double)K[r * 4 + k] * (double)E[k * 4 + c];
            P[r][c] = s;
        }
    for (int c = 0; c < 4; ++c) P[3][c] = (double)E[12 + c];
}

__device__ void load4x4(const float* m, double P[4][4]) {
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) P[r][c] = (double)m[r * 4 + c];
}

// Gauss-Jordan with partial pivoting; a singular matrix yields inf/nan like torch.inverse would raise.
__device__ void invert4x4(const double A[4][4], double inv[4][4]) {
    double M[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            M[r][c] = A[r][c];
            M[r][c + 4] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(M[col][col]);
        for (int r = col + 1; r < 4; ++r) {
            double v = fabs(M[r][col]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != col)
            for (int c = 0; c < 8; ++c) { double t = M[col][c]; M[col][c] = M[piv][c]; M[piv][c] = t; }
        const double d = 1.0 / M[col][col];
        for (int c = 0; c < 8; ++c) M[col][c] *= d;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = M[r][col];
            for (int c = 0; c < 8; ++c) M[r][c] -= f * M[col][c];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) inv[r][c] = M[r][c + 4];
}

__device__ void write_rt(const double S[4][4], const double Rinv[4][4], float* rt) {
    double M[3][4];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += S[r][k] * Rinv[k][c];
            M[r][c] = s;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) rt[r * 3 + c] = (float)M[r][c];
    for (int r = 0; r < 3; ++r) rt[9 + r] = (float)M[r][3];
}

__global__ void compose_rel_proj_kernel(const float* __restrict__ pairs, int n_views, float* __restrict__ rt) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;   // source index 0..n_views-2
    if (v >= n_views - 1) return;
    double R[4][4], Rinv[4][4], S[4][4];
    compose_k_rt(pairs, R);
    invert4x4(R, Rinv);
    compose_k_rt(pairs + (long)(v + 1) * 32, S);
    write_rt(S, Rinv, rt + v * 12);
}

// all stages of the cascade in one launch: blockIdx.x = stage
struct PairList { const float* p[4]; };
__global__ void compose_rel_proj_stages_kernel(PairList pl, int n_views, float* __restrict__ rt_all) {
    const int v = threadIdx.x;
    if (v >= n_views - 1) return;
    const float* pairs = pl.p[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if ((int)blockIdx.x == j) pairs = pl.p[j];
    double R[4][4], Rinv[4][4], S[4][4];
    compose_k_rt(pairs, R);
    invert4x4(R, Rinv);
    compose_k_rt(pairs + (long)(v + 1) * 32, S);
    write_rt(S, Rinv, rt_all + ((long)blockIdx.x * (n_views - 1) + v) * 12);
}

__global__ void rel_proj_kernel(const float* __restrict__ src, const float* __restrict__ ref, float* __restrict__ rt) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double R[4][4], Rinv[4][4], S[4][4];
    load4x4(ref, R);
    invert4x4(R, Rinv);
    load4x4(src, S);
    write_rt(S, Rinv, rt);
}

}  // namespace

extern "C" int effi_compose_rel_proj_f32(const float* pairs, int n_views, float* rt_out, effi_stream_t stream) {
    if (!pairs || !rt_out || n_views < 2 || n_views > EFFI_MAX_VIEWS + 1) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(compose_rel_proj_kernel, dim3(1), dim3(64), 0, effi_s(stream), pairs, n_views, rt_out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_compose_rel_proj_stages_f32(const float* const* pairs, int n_stages, int n_views, float* rt_out,
                                                effi_stream_t stream) {
    if (!pairs || !rt_out || n_stages < 1 || n_stages > 4 || n_views < 2 || n_views > EFFI_MAX_VIEWS + 1) return EFFI_ERR_BADARG;
    PairList pl;
    for (int k = 0; k < 4; ++k) {
        pl.p[k] = pairs[k < n_stages ? k : 0];
        if (!pl.p[k]) return EFFI_ERR_BADARG;
    }
    hipLaunchKernelGGL(compose_rel_proj_stages_kernel, dim3(n_stages), dim3(64), 0, effi_s(stream), pl, n_views, rt_out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_rel_proj_f32(const float* src_proj, const float* ref_proj, float* rt_out, effi_stream_t stream) {
    if (!src_proj || !ref_proj || !rt_out) return EFFI_ERR_BADARG;
    hipLaunchKernelGGL(rel_proj_kernel, dim3(1), dim3(64), 0, effi_s(stream), src_proj, ref_proj, rt_out);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
