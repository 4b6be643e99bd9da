// Scope row n4 (input pipeline): decoded 8-bit image -> the network's input plane set.
//   datasets/general_eval.py:83-88   read_img: np.array(img, float32) / 255.
//   datasets/general_eval.py:94-117  scale_mvs_input: cv2.resize(img, (new_w, new_h))  (INTER_LINEAR)
//   datasets/general_eval.py:160-166 second resize to the scene's standard size
//   datasets/tank.py:101-107         read_img: / 255., cv2.resize(..., INTER_LINEAR)
//   datasets/general_eval.py:189     np.stack(imgs).transpose([0, 3, 1, 2])
// One kernel does the /255, the bilinear resize and the HWC -> CHW transposition, so a view costs one H2D copy of the decoded
// bytes (3 B / pixel) instead of a float image (12 B / pixel) and no host arithmetic.  HBM-bound: 3 B read + 12 B written per
// output pixel.
#include "common.hpp"

namespace {

// cv2.resize, INTER_LINEAR, float image (the published algorithm of OpenCV's resize: pixel centres at half integers, source
// index clamped to the image, horizontal pass first, then vertical, all in fp32; the coordinate itself is computed in double
// and rounded to float).  Taps and weights of one axis:
__device__ __forceinline__ void linear_taps(int d, int dst_n, int src_n, int& s0, int& s1, float& a0, float& a1) {
    const double scale = 1.0 / ((double)dst_n / (double)src_n);
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.0f; }
    if (s >= src_n - 1) { s = src_n - 1; f = 0.0f; }
    s0 = s;
    s1 = min(s + 1, src_n - 1);
    a0 = 1.0f - f;
    a1 = f;
}

// img: [src_h][src_w][C] uint8; out: [C][dst_h][dst_w] fp32.  One thread = one output pixel, all channels.
template <int C>
__global__ __launch_bounds__(256) void image_prepare_kernel(const unsigned char* __restrict__ img, int src_h, int src_w,
                                                            int dst_h, int dst_w, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dst_w || y >= dst_h) return;
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    linear_taps(x, dst_w, src_w, x0, x1, a0, a1);
    linear_taps(y, dst_h, src_h, y0, y1, b0, b1);
    const unsigned char* r0 = img + (long)y0 * src_w * C;
    const unsigned char* r1 = img + (long)y1 * src_w * C;
    const long plane = (long)dst_h * dst_w, o = (long)y * dst_w + x;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float p00 = (float)r0[x0 * C + c] / 255.0f, p01 = (float)r0[x1 * C + c] / 255.0f;
        const float p10 = (float)r1[x0 * C + c] / 255.0f, p11 = (float)r1[x1 * C + c] / 255.0f;
        const float h0 = p00 * a0 + p01 * a1;            // horizontal pass of the two source rows
        const float h1 = p10 * a0 + p11 * a1;
        out[c * plane + o] = h0 * b0 + h1 * b1;          // vertical pass
    }
}

}  // namespace

extern "C" int effi_image_prepare_u8_f32(const unsigned char* img_hwc, int src_h, int src_w, int channels, int dst_h, int dst_w,
                                         float* out_chw, effi_stream_t stream) {
    if (!img_hwc || !out_chw || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1) return EFFI_ERR_BADARG;
    if (channels != 1 && channels != 3) return EFFI_ERR_UNSUPPORTED;
    const dim3 grid(effi_cdiv(dst_w, 64), effi_cdiv(dst_h, 4));
    hipStream_t st = effi_s(stream);
    if (channels == 3) hipLaunchKernelGGL(image_prepare_kernel<3>, grid, dim3(256), 0, st, img_hwc, src_h, src_w, dst_h, dst_w, out_chw);
    else hipLaunchKernelGGL(image_prepare_kernel<1>, grid, dim3(256), 0, st, img_hwc, src_h, src_w, dst_h, dst_w, out_chw);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

namespace {
// Same resize for an image that is already fp32 planar (the second resize of general_eval.py:160-166).
__global__ __launch_bounds__(256) void resize_planar_kernel(const float* __restrict__ in, int C, int src_h, int src_w, int dst_h,
                                                            int dst_w, float* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dst_w || y >= dst_h) return;
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    linear_taps(x, dst_w, src_w, x0, x1, a0, a1);
    linear_taps(y, dst_h, src_h, y0, y1, b0, b1);
    for (int c = 0; c < C; ++c) {
        const float* p = in + (long)c * src_h * src_w;
        const float h0 = p[(long)y0 * src_w + x0] * a0 + p[(long)y0 * src_w + x1] * a1;
        const float h1 = p[(long)y1 * src_w + x0] * a0 + p[(long)y1 * src_w + x1] * a1;
        out[((long)c * dst_h + y) * dst_w + x] = h0 * b0 + h1 * b1;
    }
}
}  // namespace

extern "C" int effi_resize_linear_f32(const float* in_chw, int channels, int src_h, int src_w, int dst_h, int dst_w, float* out_chw,
                                      effi_stream_t stream) {
    if (!in_chw || !out_chw || channels < 1 || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1) return EFFI_ERR_BADARG;
    const dim3 grid(effi_cdiv(dst_w, 64), effi_cdiv(dst_h, 4));
    hipLaunchKernelGGL(resize_planar_kernel, grid, dim3(256), 0, effi_s(stream), in_chw, channels, src_h, src_w, dst_h, dst_w, out_chw);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
