// Sustained rate of the fp32-input MFMA shapes on gfx950 (operands in registers, independent accumulators).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate mfma_f32_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
double run(F launch, double flop_per_launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return flop_per_launch * 5 / (ms * 1e-3) / 1e12;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 4096 * 4);
    const int iters = 20000;
    for (int wpc : {1, 2, 4}) {   // workgroups (4 waves each) per CU -> waves per SIMD
        int grid = 256 * wpc;
        double f16 = 2.0 * 16 * 16 * 4 * 4 /*acc*/ * (double)iters * 4 /*waves*/ * grid;
        printf("16x16x4 f32, 4 acc, %d waves/SIMD: %.1f TFLOP/s\n", wpc, run([&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, f16));
        double f16_1 = 2.0 * 16 * 16 * 4 * 1 * (double)iters * 4 * grid;
        printf("16x16x4 f32, 1 acc, %d waves/SIMD: %.1f TFLOP/s\n", wpc, run([&] { hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, f16_1));
        double f32 = 2.0 * 32 * 32 * 2 * 2 * (double)iters * 4 * grid;
        printf("32x32x2 f32, 2 acc, %d waves/SIMD: %.1f TFLOP/s\n", wpc, run([&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); }, f32));
    }
    return 0;
}
