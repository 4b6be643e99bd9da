// What does a partially needed gather cost on the L1 / texture path of gfx950?  The stage-2/3 warp + correlation kernel
// (csrc/warpcorr.hip, warpcorr_dyn_kernel) issues 4 taps x 16 B per lane and hypothesis although consecutive hypotheses of a pixel
// mostly hit the same 2 x 2 block of source pixels (0.3-0.7 pixels apart on the DTU-shaped rig).  Three ways of NOT fetching a tap
// again are timed here on the kernel's own access shape (channel-last map, 32-byte pixels, two lanes per pixel, 592 x 800):
//   all        every lane loads every tap                                        (what the kernel does today)
//   masked     a lane loads only when it needs the tap (divergent branch, exec mask), keep-probability KEEP of not needing it
//   redirect   every lane loads, but lanes that do not need the tap read one shared hot address
//   rowmask    like masked, but whole 16-lane rows need / do not need the tap
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ta_gather ta_gather.hip && /tmp/ta_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int H = 592, W = 800, C = 8, NV = 4, NH = 8;

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
    unsigned x = a * 2654435761u ^ (b + 0x9e3779b9u) * 2246822519u ^ (c + 0x85ebca6bu) * 3266489917u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    return x;
}

// MODE 0 all, 1 masked, 2 redirect, 3 rowmask.  keep256 = probability x 256 that a (pixel, view, hypothesis) re-uses its block.
// All modes share one structure per view: (A) the flags and addresses of the NH hypotheses, (B) every load of the view issued
// back to back into its own registers (under the lane's mask in modes 1 / 3), (C) re-used blocks copied from the hypothesis before,
// then the arithmetic -- so that the modes differ in what the texture path has to do, not in how far the compiler overlaps loads.
template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ maps, int keep256, float* __restrict__ out) {
    const int g = blockIdx.x * 128 + (threadIdx.x >> 1), sub = threadIdx.x & 1;
    if (g >= H * W) return;
    const int y = g / W, x = g - y * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int v = 0; v < NV; ++v) {
        const float* __restrict__ m = maps + (long)v * H * W * C;
        // a rough depth map: the block of a pixel is displaced by a per-pixel pseudo-random amount (+-6 pixels), then walks to the right
        const int jx = (int)(hash3(g, v, 7u) % 13u) - 6, jy = (int)(hash3(g, v, 11u) % 5u) - 2;
        int bx = min(max(x + jx, 0), W - 10);
        const int by = min(max(y + jy, 0), H - 2);
        bool keep[NH];
        long o00[NH];
        float wgt[NH];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const unsigned r = hash3(g, v, 100u + hh) & 255u;
            keep[hh] = hh > 0 && (int)r < keep256;
            if (MODE == 3) keep[hh] = hh > 0 && (int)(hash3(blockIdx.x * 16 + (threadIdx.x >> 4), v, 100u + hh) & 255u) < keep256;
            if (!keep[hh] && hh > 0) bx += 1;
            o00[hh] = ((long)by * W + bx) * C + sub * 4;
            wgt[hh] = (float)(r + 1) * (1.0f / 256.0f);
        }
        float4 t[NH][4];
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
            const long z = sub * 4;
            const long a = (MODE == 2 && keep[hh]) ? z : o00[hh];
            const long dx = (MODE == 2 && keep[hh]) ? 0 : C, dy = (MODE == 2 && keep[hh]) ? 0 : (long)W * C;
            if (MODE == 0 || MODE == 2 || !keep[hh]) {
                t[hh][0] = *reinterpret_cast<const float4*>(m + a);
                t[hh][1] = *reinterpret_cast<const float4*>(m + a + dx);
                t[hh][2] = *reinterpret_cast<const float4*>(m + a + dy);
                t[hh][3] = *reinterpret_cast<const float4*>(m + a + dy + dx);
            }
        }
#pragma unroll
        for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (hh > 0 && MODE != 0 && keep[hh]) t[hh][k] = t[hh - 1][k];
                acc.x = fmaf(wgt[hh], t[hh][k].x, acc.x); acc.y = fmaf(wgt[hh], t[hh][k].y, acc.y);
                acc.z = fmaf(wgt[hh], t[hh][k].z, acc.z); acc.w = fmaf(wgt[hh], t[hh][k].w, acc.w);
            }
        }
    }
    out[(long)g * 2 + sub] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE>
static float run(const float* maps, int keep256, float* out, const char* name) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = (H * W + 127) / 128;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gather_kernel<MODE>, dim3(grid), dim3(256), 0, 0, maps, keep256, out);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather_kernel<MODE>, dim3(grid), dim3(256), 0, 0, maps, keep256, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s keep %3d/256  %8.1f us per launch\n", name, keep256, ms * 1000.0f / reps);
    return ms;
}

int main() {
    float *maps, *out;
    const size_t n = (size_t)NV * H * W * C;
    CK(hipMalloc(&maps, n * sizeof(float)));
    CK(hipMalloc(&out, (size_t)H * W * 2 * sizeof(float)));
    CK(hipMemset(maps, 0, n * sizeof(float)));
    printf("gather of %d views x %d hypotheses x 4 taps x 32 B per pixel, %d x %d pixels, two lanes per pixel\n", NV, NH, H, W);
    for (int keep : {0, 128, 171, 213}) {
        run<0>(maps, keep, out, "all");
        run<1>(maps, keep, out, "masked");
        run<2>(maps, keep, out, "redirect");
        run<3>(maps, keep, out, "rowmask");
    }
    CK(hipFree(maps)); CK(hipFree(out));
    return 0;
}
