// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of this repository's kernels.
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B; a 128-B request is tallied at 64 B, so a wide coalesced
// stream (16 B per lane) reads exactly HALF; "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern".  Every kernel below reads each byte of a 1 GiB buffer (far larger than the 256 MiB Infinity Cache) exactly ONCE
// in one of those patterns, so true bytes / FETCH_SIZE is the factor to apply to a kernel with that pattern:
//   stream16   16 B per lane, 1 KiB per wave-instruction            (reference point: expect 2.0)
//   stream4     4 B per lane, 256 B per wave-instruction            (encoder_inputs / lookups / conv3d tile fill)
//   seg<N>     16 B per lane in row segments of N bytes, consecutive segments one image row apart (N = 96: the 16-pixel
//              tiles of the split-precision conv / transposed-conv staging incl. halo; 160; 288: the 4 x 64 "wide" tiles)
//   seg32      16 B per lane in 32-byte pieces at scattered pixels  (channel-last taps of the C = 8 warp kernel)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip && rocprofv3 --pmc FETCH_SIZE ... -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void stream16_kernel(const float4* __restrict__ p, long n4, float* __restrict__ sink) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5f) sink[0] = acc;
}
__global__ void stream4_kernel(const float* __restrict__ p, long n, float* __restrict__ sink) {
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 1234.5f) sink[0] = acc;
}
// the buffer is an "image" of rows of ROWB bytes; a tile is (rows R) x (segment SEG bytes); every tile is read once, tiles tile the
// image without overlap (SEG divides ROWB).  A wave's lanes cover consecutive 16-byte pieces of a segment, then the next row.
template <int SEG>
__global__ void seg_kernel(const char* __restrict__ p, long rows, int rowb, float* __restrict__ sink) {
    constexpr int Q = SEG / 16;                       // 16-byte pieces per segment
    const int tiles_x = rowb / SEG;
    const long row_groups = rows / 8;                 // tiles are 8 rows tall
    const long ntiles = row_groups * tiles_x;
    float acc = 0.f;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const long ty = t / tiles_x, tx = t - ty * tiles_x;
        for (int e = threadIdx.x; e < 8 * Q; e += blockDim.x) {
            const int r = e / Q, q = e - r * Q;
            const float4 v = *reinterpret_cast<const float4*>(p + ((ty * 8 + r) * (long)rowb + tx * SEG + q * 16));
            acc += v.x + v.y + v.z + v.w;
        }
    }
    if (acc == 1234.5f) sink[0] = acc;
}
// 32-byte pixels visited in a scattered (multiplicative-hash) order, two lanes per pixel
__global__ void seg32_kernel(const char* __restrict__ p, long npix, float* __restrict__ sink) {
    float acc = 0.f;
    const long nl = npix * 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nl; i += (long)gridDim.x * blockDim.x) {
        const long pix = ((i >> 1) * 2654435761L) % npix;        // odd multiplier, npix a power of two: a permutation
        const float4 v = *reinterpret_cast<const float4*>(p + pix * 32 + (i & 1) * 16);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5f) sink[0] = acc;
}

int main() {
    const long bytes = 1L << 30;
    char* buf; float* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, bytes));
    const int rowb = 1440 * 16;                       // 23040 B: divisible by 96, 160 and 288
    const long rows = (bytes / rowb) / 8 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16_kernel, dim3(4096), dim3(256), 0, 0, (const float4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(stream4_kernel, dim3(4096), dim3(256), 0, 0, (const float*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(seg_kernel<96>, dim3(8192), dim3(64), 0, 0, buf, rows, rowb, sink);
        hipLaunchKernelGGL(seg_kernel<160>, dim3(8192), dim3(128), 0, 0, buf, rows, rowb, sink);
        hipLaunchKernelGGL(seg_kernel<288>, dim3(8192), dim3(192), 0, 0, buf, rows, rowb, sink);
        hipLaunchKernelGGL(seg32_kernel, dim3(4096), dim3(256), 0, 0, buf, bytes / 32, sink);
    }
    CK(hipDeviceSynchronize());
    printf("true_bytes stream16 %ld stream4 %ld seg96 %ld seg160 %ld seg288 %ld seg32 %ld\n", bytes, bytes, rows * rowb, rows * rowb, rows * rowb, bytes);
    return 0;
}
