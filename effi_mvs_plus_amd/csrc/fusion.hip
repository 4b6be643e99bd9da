// Scope row n3: the dynamic geometric-consistency filter and depth averaging that consume the
// path's depth maps (reference: misc/fusion.py:8-46,117-181 and the tensor part of test_tank.py:466-512).
// One thread per reference pixel walks all source views: project into the source view, bilinear-sample its depth
// (F.grid_sample, align_corners=True, zero padding), project the sampled point back, compare position and depth against
// the V+1-thres_view threshold pairs, count, and finish with the averaged depth, the dynamic view-count rule, the
// photometric mask and the world-space point -- nothing of the reference's [n,v,3,h,w] intermediates is materialised
// unless the caller asks for reproj_xyd.  Arithmetic is fp32 in the reference's operation order; the 3x3 / 4x4 inverses
// are formed once per view on the device in fp64 and rounded (torch.inverse in fp32 differs from that by ~1e-7 relative).
#include "common.hpp"

namespace {

constexpr int FUS_MAX_VIEWS = 16;
constexpr int MAT_STRIDE = 52;        // per view: K[9] Kinv[9] E[16] Einv[16] (+2 pad)

__device__ void fus_invert(const double* A, int n, double* inv) {           // Gauss-Jordan, partial pivoting, n <= 4
    double M[4][8];
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            M[r][c] = A[r * n + c];
            M[r][c + n] = (r == c) ? 1.0 : 0.0;
        }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        double best = fabs(M[col][col]);
        for (int r = col + 1; r < n; ++r)
            if (fabs(M[r][col]) > best) { best = fabs(M[r][col]); piv = r; }
        if (piv != col)
            for (int c = 0; c < 2 * n; ++c) { const double t = M[col][c]; M[col][c] = M[piv][c]; M[piv][c] = t; }
        const double d = 1.0 / M[col][col];
        for (int c = 0; c < 2 * n; ++c) M[col][c] *= d;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            const double f = M[r][col];
            for (int c = 0; c < 2 * n; ++c) M[r][c] -= f * M[col][c];
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) inv[r * n + c] = M[r][c + n];
}

// cams: view 0 = reference ([2][4][4]: extrinsic, intrinsic in the top-left 3x3), views 1..V = sources
__global__ void fusion_prepare_kernel(const float* __restrict__ ref_cam, const float* __restrict__ src_cams, int V,
                                      float* __restrict__ mats) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > V) return;
    const float* cam = (v == 0) ? ref_cam : src_cams + (long)(v - 1) * 32;
    double K[9], Ki[9], E[16], Ei[16];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K[r * 3 + c] = (double)cam[16 + r * 4 + c];
    for (int i = 0; i < 16; ++i) E[i] = (double)cam[i];
    fus_invert(K, 3, Ki);
    fus_invert(E, 4, Ei);
    float* m = mats + (long)v * MAT_STRIDE;
    for (int i = 0; i < 9; ++i) { m[i] = (float)K[i]; m[9 + i] = (float)Ki[i]; }
    for (int i = 0; i < 16; ++i) { m[18 + i] = (float)E[i]; m[34 + i] = (float)Ei[i]; }
}

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

__device__ __forceinline__ V3 mul3(const float* M, V3 p) {
    return {M[0] * p.x + M[1] * p.y + M[2] * p.z, M[3] * p.x + M[4] * p.y + M[5] * p.z, M[6] * p.x + M[7] * p.y + M[8] * p.z};
}
__device__ __forceinline__ V4 mul4(const float* M, V4 p) {
    return {M[0] * p.x + M[1] * p.y + M[2] * p.z + M[3] * p.w, M[4] * p.x + M[5] * p.y + M[6] * p.z + M[7] * p.w,
            M[8] * p.x + M[9] * p.y + M[10] * p.z + M[11] * p.w, M[12] * p.x + M[13] * p.y + M[14] * p.z + M[15] * p.w};
}
// The reference divides every component by the same denominator.  On the way INTO the source view (up to the bilinear sample,
// whose floor() is discontinuous) the quotients are formed exactly as the reference does; on the way BACK the denominator is
// inverted once (IEEE division) and the components are multiplied (<= 1 ulp per component, continuous outputs only), which
// removes 12 of the 27 division sequences per pixel and source view -- the kernel is bound by them.
// idx_img2cam (misc/fusion.py:23-28): K^-1 [u v 1], normalised by its z (+1e-9), times depth; homogeneous w = 1
template <bool EXACT>
__device__ __forceinline__ V4 img2cam(const float* Kinv, float u, float v, float depth) {
    V3 c = mul3(Kinv, {u, v, 1.0f});
    const float d = c.z + 1e-9f;
    if (EXACT) return {c.x / d * depth, c.y / d * depth, c.z / d * depth, 1.0f};
    const float r = 1.0f / d;
    return {c.x * r * depth, c.y * r * depth, c.z * r * depth, 1.0f};
}
// idx_cam2world / idx_world2cam (:31-40): 4x4 times the point, normalised by w (+1e-9)
template <bool EXACT>
__device__ __forceinline__ V4 xform(const float* M, V4 p) {
    V4 q = mul4(M, p);
    const float d = q.w + 1e-9f;
    if (EXACT) return {q.x / d, q.y / d, q.z / d, q.w / d};
    const float r = 1.0f / d;
    return {q.x * r, q.y * r, q.z * r, q.w * r};
}
// idx_cam2img (:43-47)
template <bool EXACT>
__device__ __forceinline__ V3 cam2img(const float* K, V4 c) {
    const float d = c.w + 1e-9f;
    if (EXACT) {
        V3 i = mul3(K, {c.x / d, c.y / d, c.z / d});
        const float e = i.z + 1e-9f;
        return {i.x / e, i.y / e, i.z / e};
    }
    const float r = 1.0f / d;
    V3 i = mul3(K, {c.x * r, c.y * r, c.z * r});
    const float e = 1.0f / (i.z + 1e-9f);
    return {i.x * e, i.y * e, i.z * e};
}

__device__ __forceinline__ float sample_bilinear_zero(const float* __restrict__ img, int h, int w, float u, float v) {
    // F.grid_sample(mode=bilinear, padding_mode=zeros, align_corners=True) on coordinates normalised as the reference does
    const float gx = u / ((float)(w - 1) / 2.0f) - 1.0f, gy = v / ((float)(h - 1) / 2.0f) - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(w - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(h - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[(long)yy * w + xx] : 0.0f; };
    const float nw = (1.0f - tx) * (1.0f - ty), ne = tx * (1.0f - ty), sw = (1.0f - tx) * ty, se = tx * ty;
    return at(y0, x0) * nw + at(y0, x0 + 1) * ne + at(y0 + 1, x0) * sw + at(y0 + 1, x0 + 1) * se;
}

__global__ __launch_bounds__(256) void fusion_dynamic_filter_kernel(
    const float* __restrict__ ref_depth, const float* __restrict__ src_depths, int V, int h, int w,
    const float* __restrict__ mats, const float* __restrict__ conf, int ch, int cw, float prob_thr, int dh, float dist_base,
    float rel_base, int relative, float* __restrict__ out_depth, unsigned char* __restrict__ out_geo,
    unsigned char* __restrict__ out_prob, unsigned char* __restrict__ out_mask, float* __restrict__ out_points,
    float* __restrict__ out_xyd) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)h * w;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const float u = (float)x + 0.5f, vv = (float)y + 0.5f;
    const float* Mr = mats;
    const float dref = ref_depth[p];
    const V4 ref_cam_pt = img2cam<true>(Mr + 9, u, vv, dref);
    const V4 world = xform<true>(Mr + 34, ref_cam_pt);
    int counts[FUS_MAX_VIEWS + 1];
#pragma unroll
    for (int i = 0; i <= FUS_MAX_VIEWS; ++i) counts[i] = 0;
    const int nthr = V + 1 - dh;                                   // thresholds i = dh .. V
    float dsum = 0.0f;
    int nvis = 0;
    for (int s = 0; s < V; ++s) {
        const float* Ms = mats + (long)(s + 1) * MAT_STRIDE;
        const V4 c = xform<true>(Ms + 18, world);
        const V3 im = cam2img<true>(Ms, c);
        const float ds = sample_bilinear_zero(src_depths + (long)s * hw, h, w, im.x, im.y);
        const V4 sc = img2cam<false>(Ms + 9, im.x, im.y, ds);
        const V4 sw = xform<false>(Ms + 34, sc);
        const V4 rc = xform<false>(Mr + 18, sw);
        const float reproj_depth = rc.z;
        const V3 ri = cam2img<false>(Mr, rc);
        if (out_xyd) {
            out_xyd[((long)s * 3 + 0) * hw + p] = ri.x;
            out_xyd[((long)s * 3 + 1) * hw + p] = ri.y;
            out_xyd[((long)s * 3 + 2) * hw + p] = reproj_depth;
        }
        const float dx = ri.x - u, dy = ri.y - vv;
        const float cdiff = sqrtf(dx * dx + dy * dy);
        float ddiff = fabsf(dref - reproj_depth);
        if (relative) ddiff = ddiff / dref;
#pragma unroll
        for (int k = 0; k <= FUS_MAX_VIEWS; ++k) {
            if (k < nthr) {
                const float step = (float)(dh + k);
                const bool m = (cdiff < step / dist_base) & (ddiff < step / rel_base);
                counts[k] += m ? 1 : 0;
                if (k == nthr - 1 && m) {                           // vis_mask = loosest threshold (misc/fusion.py:179)
                    dsum += reproj_depth;
                    nvis += 1;
                }
            }
        }
    }
    const float davg = (dsum + dref) / (float)(nvis + 1);           // test_tank.py:498-499
    bool geo = nvis >= V + 1;                                        // :502 (never true; kept for fidelity)
#pragma unroll
    for (int k = 0; k <= FUS_MAX_VIEWS; ++k)
        if (k < nthr && k < V + 1 - dh) geo = geo | (counts[k] >= dh + k);   // :503-504
    bool pm = true;
    if (conf) {                                                     // F.interpolate(nearest) of the confidence, :471-473
        const int sy = min((int)floorf((float)y * ((float)ch / (float)h)), ch - 1);
        const int sx = min((int)floorf((float)x * ((float)cw / (float)w)), cw - 1);
        pm = conf[(long)sy * cw + sx] > prob_thr;
    }
    out_depth[p] = davg;
    if (out_geo) out_geo[p] = geo ? 1 : 0;
    if (out_prob) out_prob[p] = pm ? 1 : 0;
    if (out_mask) out_mask[p] = (geo & pm) ? 1 : 0;
    if (out_points) {                                               // :507-509
        const V4 pc = img2cam<true>(Mr + 9, u, vv, davg);
        const V4 pw = xform<true>(Mr + 34, pc);
        out_points[p] = pw.x;
        out_points[hw + p] = pw.y;
        out_points[2 * hw + p] = pw.z;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// DTU branch of row n3: reproject_with_depth + check_geometric_consistency + the array part of filter_depth of the reference's
// test_dtu_dypcd.py:164-333 (the numpy / cv2 filter its DTU driver runs per scan on the host, in a multiprocessing pool), one thread
// per reference pixel walking the source views.  Differences from the Tanks-and-Temples kernel above that matter for parity: pixel
// coordinates are INTEGERS (no half-pixel offset), the projection chain is evaluated in DOUBLE (numpy promotes int64 grid x
// float32 depth to float64) and rounded to float32 exactly where the reference calls .astype(np.float32), the source depth is
// sampled with cv2.remap(INTER_LINEAR)'s arithmetic (coordinates rounded to 1/32 pixel, table weights, zero outside), and ten
// (distance, depth-difference) threshold pairs i = s .. e-1 are counted.  PARITY UNPINNED: cv2 is not installed in the build image
// and the reference holds no fixtures for this code; the checker is oracle/effi_dtu_filter_oracle.py (the numpy lines restated with
// their dtypes, OpenCV's published remap algorithm restated).  Matrix inverses / products are formed once per view in double and
// rounded to float32 (the reference: single-precision LAPACK / float32 matmul; ~1e-7 relative apart).
// per view: ref: K[9] Kinv[9] E[16] Einv[16]; source s: K[9] Kinv[9] T_ref->src[16] = E_s . E_ref^-1, T_src->ref[16] = E_ref . E_s^-1
__global__ void fusion_dtu_prepare_kernel(const float* __restrict__ ref_cam, const float* __restrict__ src_cams, int V,
                                          float* __restrict__ mats) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v > V) return;
    auto load = [](const float* cam, double* K, double* E) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) K[r * 3 + c] = (double)cam[16 + r * 4 + c];
        for (int i = 0; i < 16; ++i) E[i] = (double)cam[i];
    };
    double K[9], Ki[9], E[16], Ei[16], Kr[9], Er[16], Eri[16];
    load(ref_cam, Kr, Er);
    fus_invert(Er, 4, Eri);
    float* m = mats + (long)v * MAT_STRIDE;
    if (v == 0) {
        fus_invert(Kr, 3, Ki);
        for (int i = 0; i < 9; ++i) { m[i] = (float)Kr[i]; m[9 + i] = (float)Ki[i]; }
        for (int i = 0; i < 16; ++i) { m[18 + i] = (float)Er[i]; m[34 + i] = (float)Eri[i]; }
        return;
    }
    load(src_cams + (long)(v - 1) * 32, K, E);
    fus_invert(K, 3, Ki);
    fus_invert(E, 4, Ei);
    for (int i = 0; i < 9; ++i) { m[i] = (float)K[i]; m[9 + i] = (float)Ki[i]; }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double a = 0.0, b = 0.0;
            for (int k = 0; k < 4; ++k) {
                a += (double)(float)E[r * 4 + k] * (double)(float)Eri[k * 4 + c];       // operands as float32, as the reference holds them
                b += (double)(float)Er[r * 4 + k] * (double)(float)Ei[k * 4 + c];
            }
            m[18 + r * 4 + c] = (float)a;
            m[34 + r * 4 + c] = (float)b;
        }
}

struct D3 { double x, y, z; };
__device__ __forceinline__ D3 dmul3(const float* M, double x, double y, double z) {
    return {(double)M[0] * x + (double)M[1] * y + (double)M[2] * z, (double)M[3] * x + (double)M[4] * y + (double)M[5] * z,
            (double)M[6] * x + (double)M[7] * y + (double)M[8] * z};
}
__device__ __forceinline__ D3 dmul4_xyz(const float* M, D3 p) {          // rows 0..2 of M . [p; 1]
    return {(double)M[0] * p.x + (double)M[1] * p.y + (double)M[2] * p.z + (double)M[3],
            (double)M[4] * p.x + (double)M[5] * p.y + (double)M[6] * p.z + (double)M[7],
            (double)M[8] * p.x + (double)M[9] * p.y + (double)M[10] * p.z + (double)M[11]};
}

// cv2.remap(src, map_x, map_y, INTER_LINEAR), BORDER_CONSTANT 0, one sample (OpenCV imgwarp.cpp: INTER_BITS = 5)
__device__ __forceinline__ float cv_remap_linear(const float* __restrict__ img, int h, int w, float mx, float my) {
    auto fix = [](float v) -> long {
        double f = (double)v * 32.0;
        if (!(f == f)) return -2147483648L;                    // NaN: cvRound gives INT_MIN
        f = fmin(fmax(f, -2147483648.0), 2147483647.0);
        return (long)rint(f);                                   // cvRound: round half to even
    };
    const long sx = fix(mx), sy = fix(my);
    const int ax = (int)(sx & 31), ay = (int)(sy & 31);
    const long x0 = min(max(sx >> 5, -32768L), 32767L), y0 = min(max(sy >> 5, -32768L), 32767L);      // saturate_cast<short>
    const float tx = (float)ax * (1.0f / 32.0f), ty = (float)ay * (1.0f / 32.0f);
    const float wx0 = 1.0f - tx, wy0 = 1.0f - ty;
    auto at = [&](long yy, long xx) { return (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[yy * w + xx] : 0.0f; };
    return at(y0, x0) * (wy0 * wx0) + at(y0, x0 + 1) * (wy0 * tx) + at(y0 + 1, x0) * (ty * wx0) + at(y0 + 1, x0 + 1) * (ty * tx);
}

constexpr int DTU_MAX_THR = 16;

// reproject_with_depth for ONE reference pixel and one source view (test_dtu_dypcd.py:164-205), shared by the fused filter kernel and
// the per-function kernel below: one arithmetic.
struct DtuReproj {
    float x_src, y_src, depth_rep, x_rep, y_rep;
};
__device__ __forceinline__ DtuReproj dtu_reproject_pixel(const float* __restrict__ Mr, const float* __restrict__ Ms,
                                                         const float* __restrict__ src_depth, int h, int w, double xd, double yd,
                                                         float dref) {
    // reference 3-D point: K_ref^-1 . ([x y 1] * depth)                                                     (:172-173)
    const D3 xyz_ref = dmul3(Mr + 9, xd * (double)dref, yd * (double)dref, (double)dref);
    const D3 xs = dmul4_xyz(Ms + 18, xyz_ref);                                                             // (:175-176)
    const D3 kx = dmul3(Ms, xs.x, xs.y, xs.z);                                                             // (:178-179)
    const double u = kx.x / kx.z, v = kx.y / kx.z;
    DtuReproj r;
    r.x_src = (float)u;                                                                                    // (:182-183)
    r.y_src = (float)v;
    const float samp = cv_remap_linear(src_depth, h, w, r.x_src, r.y_src);                                 // (:184)
    const D3 s3 = dmul3(Ms + 9, u * (double)samp, v * (double)samp, (double)samp);                         // (:190-191)
    const D3 rp = dmul4_xyz(Ms + 34, s3);                                                                  // (:193-194)
    r.depth_rep = (float)rp.z;                                                                             // (:196)
    D3 kr = dmul3(Mr, rp.x, rp.y, rp.z);                                                                   // (:197)
    if (kr.z == 0.0) kr.z += 0.00001;                                                                      // (:198)
    r.x_rep = (float)(kr.x / kr.z);                                                                        // (:199-201)
    r.y_rep = (float)(kr.y / kr.z);
    return r;
}

__global__ __launch_bounds__(256) void fusion_dtu_filter_kernel(
    const float* __restrict__ ref_depth, const float* __restrict__ src_depths, int V, int h, int w, const float* __restrict__ mats,
    const float* __restrict__ conf, float conf_thr, float conf_keep, int s_lo, int e_hi, float dist_base, float diff_base,
    float* __restrict__ out_depth, unsigned char* __restrict__ out_photo, unsigned char* __restrict__ out_geo,
    unsigned char* __restrict__ out_final, float* __restrict__ out_points) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)h * w;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const double xd = (double)x, yd = (double)y;
    const float* Mr = mats;
    const float dref = ref_depth[p];
    const int nthr = e_hi - s_lo;
    float thr_diff[DTU_MAX_THR];
    double thr_dist[DTU_MAX_THR];
    int counts[DTU_MAX_THR];
#pragma unroll
    for (int k = 0; k < DTU_MAX_THR; ++k) {
        const int i = s_lo + k;
        counts[k] = 0;
        thr_dist[k] = (double)i * (double)dist_base;                                                          // i * dist_base (:227)
        thr_diff[k] = (float)(log(fmax((double)i, 1.05)) / log(10.0) * (double)diff_base);    // math.log(max(i, 1.05), 10) * diff_base; float32 comparison
    }
    float num = 0.0f;                          // sum(all_srcview_depth_ests): float32, in view order (:299)
    int nlast = 0;
    for (int sv = 0; sv < V; ++sv) {
        const float* Ms = mats + (long)(sv + 1) * MAT_STRIDE;
        const DtuReproj rr = dtu_reproject_pixel(Mr, Ms, src_depths + (long)sv * hw, h, w, xd, yd, dref);     // (:164-205)
        const float depth_rep = rr.depth_rep, xr = rr.x_rep, yr = rr.y_rep;
        const double dxr = (double)xr - xd, dyr = (double)yr - yd;
        const double dist = sqrt(dxr * dxr + dyr * dyr);                                                       // (:218) float32 - int64 -> float64
        const float ddiff = fabsf(depth_rep - dref);                                                           // (:225)
        bool last = false;
#pragma unroll
        for (int k = 0; k < DTU_MAX_THR; ++k) {
            if (k < nthr) {
                const bool m = (dist < thr_dist[k]) & (ddiff < thr_diff[k]);
                counts[k] += m ? 1 : 0;
                if (k == nthr - 1) last = m;
            }
        }
        if (last) {                            // depth_reprojected[~mask] = 0 with mask = the LAST (loosest) pair (:229)
            num = num + depth_rep;
            nlast += 1;
        }
    }
    double davg = (double)(num + dref) / (double)(nlast + 1);                                                  // (:299) float32 / int -> float64
    const float c = conf ? conf[p] : 1.0f;
    if (c > conf_keep) davg = (double)dref;                                                                    // (:300)
    bool geo = nlast >= e_hi;                                                                                  // (:303) dy_range = e
#pragma unroll
    for (int k = 0; k < DTU_MAX_THR; ++k)
        if (k < nthr) geo = geo | (counts[k] >= s_lo + k);                                                     // (:304-305)
    const bool photo = c > conf_thr;                                                                           // (:261)
    out_depth[p] = (float)davg;
    if (out_photo) out_photo[p] = photo ? 1 : 0;
    if (out_geo) out_geo[p] = geo ? 1 : 0;
    if (out_final) out_final[p] = (photo & geo) ? 1 : 0;
    if (out_points) {                                                                                          // (:327-330)
        const D3 pc = dmul3(Mr + 9, xd * davg, yd * davg, davg);
        const D3 pw = dmul4_xyz(Mr + 34, pc);
        out_points[p] = (float)pw.x;
        out_points[hw + p] = (float)pw.y;
        out_points[2 * hw + p] = (float)pw.z;
    }
}

// reproject_with_depth (test_dtu_dypcd.py:164-205) and, with ``masks``, check_geometric_consistency (:208-233) for one
// (reference, source) pair as functions of their own: out5 = depth_reprojected, x_reprojected, y_reprojected, x_src, y_src.
// With masks: masks[k] = dist < (s+k) * dist_base & depth_diff < log10(max(s+k, 1.05)) * diff_base, and the first three outputs are
// zeroed where the LAST mask is false (:229-231).  PARITY UNPINNED like the fused DTU kernel (cv2.remap restated).
__global__ __launch_bounds__(256) void fusion_dtu_reproject_kernel(const float* __restrict__ ref_depth, const float* __restrict__ src_depth,
                                                                   int h, int w, const float* __restrict__ mats, int s_lo, int e_hi,
                                                                   float dist_base, float diff_base, float* __restrict__ out5,
                                                                   unsigned char* __restrict__ masks) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)h * w;
    if (p >= hw) return;
    const int y = p / w, x = p - y * w;
    const double xd = (double)x, yd = (double)y;
    const float dref = ref_depth[p];
    DtuReproj r = dtu_reproject_pixel(mats, mats + MAT_STRIDE, src_depth, h, w, xd, yd, dref);
    if (masks) {
        const double dxr = (double)r.x_rep - xd, dyr = (double)r.y_rep - yd;
        const double dist = sqrt(dxr * dxr + dyr * dyr);
        const float ddiff = fabsf(r.depth_rep - dref);
        bool last = false;
        for (int k = 0; k < e_hi - s_lo; ++k) {
            const int i = s_lo + k;
            const float thr_diff = (float)(log(fmax((double)i, 1.05)) / log(10.0) * (double)diff_base);
            last = (dist < (double)i * (double)dist_base) & (ddiff < thr_diff);
            masks[(long)k * hw + p] = last ? 1 : 0;
        }
        if (!last) r.depth_rep = r.x_rep = r.y_rep = 0.0f;
    }
    out5[p] = r.depth_rep;
    out5[hw + p] = r.x_rep;
    out5[2 * hw + p] = r.y_rep;
    out5[3 * hw + p] = r.x_src;
    out5[4 * hw + p] = r.y_src;
}

// ------------------------------------------------------------------------------------------------------------------------
// The reference's misc/fusion.py functions the T&T driver calls ONE BY ONE (test_tank.py:486-509) -- vis_filter_dynamic and the
// idx_* point transforms -- as kernels of their own, so that the driver's lines run unchanged on this package.
// The fused kernel above remains the fast path (dynamic_filter); these reproduce the individual functions' outputs.

// vis_filter_dynamic (misc/fusion.py:157-181): masks[n][v][k][p] = |reproj_xy - pixel centre| < (thres+k)/dist_base  AND
// |ref_depth - reproj_depth| (/ ref_depth) < (thres+k)/rel_diff_base, k = 0 .. v - thres
__global__ __launch_bounds__(256) void fusion_vis_filter_kernel(const float* __restrict__ ref_depth, const float* __restrict__ xyd,
                                                                int V, int h, int w, float dist_base, float rel_base, int thres,
                                                                int relative, unsigned char* __restrict__ masks) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)h * w;
    if (p >= hw) return;
    const int b = blockIdx.y / V, s = blockIdx.y - b * V;
    const int y = p / w, x = p - y * w;
    const float u = (float)x + 0.5f, vv = (float)y + 0.5f;
    const float dref = ref_depth[(long)b * hw + p];
    const float* q = xyd + ((long)(b * V + s) * 3) * hw + p;
    const float dx = q[0] - u, dy = q[hw] - vv;
    const float cdiff = sqrtf(dx * dx + dy * dy);
    float ddiff = fabsf(dref - q[2 * hw]);
    if (relative) ddiff = ddiff / dref;
    const int nthr = V + 1 - thres;
    unsigned char* m = masks + ((long)(b * V + s) * nthr) * hw + p;
    for (int k = 0; k < nthr; ++k) {
        const float step = (float)(thres + k);
        m[(long)k * hw] = ((cdiff < step / dist_base) & (ddiff < step / rel_base)) ? 1 : 0;
    }
}

// one camera per batch element: K[9] Kinv[9] E[16] Einv[16]
__global__ void fusion_prepare_batch_kernel(const float* __restrict__ cams, int n, float* __restrict__ mats) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const float* cam = cams + (long)v * 32;
    double K[9], Ki[9], E[16], Ei[16];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) K[r * 3 + c] = (double)cam[16 + r * 4 + c];
    for (int i = 0; i < 16; ++i) E[i] = (double)cam[i];
    fus_invert(K, 3, Ki);
    fus_invert(E, 4, Ei);
    float* m = mats + (long)v * MAT_STRIDE;
    for (int i = 0; i < 9; ++i) { m[i] = (float)K[i]; m[9 + i] = (float)Ki[i]; }
    for (int i = 0; i < 16; ++i) { m[18 + i] = (float)E[i]; m[34 + i] = (float)Ei[i]; }
}

// MODE 0: idx_img2cam (:23-28)  in [..,3] (+ depth) -> out [..,4];  1: idx_cam2world (:31-34)  4 -> 4;
//      2: idx_world2cam (:37-40) 4 -> 4;                             3: idx_cam2img (:43-47)    4 -> 3
template <int MODE>
__global__ __launch_bounds__(256) void fusion_points_kernel(const float* __restrict__ in, long in_bstride, const float* __restrict__ depth,
                                                            const float* __restrict__ mats, int hw, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const int b = blockIdx.y;
    const float* M = mats + (long)b * MAT_STRIDE;
    constexpr int NI = (MODE == 0) ? 3 : 4, NO = (MODE == 3) ? 3 : 4;
    const float* q = in + (long)b * in_bstride + (long)p * NI;
    float* o = out + ((long)b * hw + p) * NO;
    if (MODE == 0) {
        const V3 c = mul3(M + 9, {q[0], q[1], q[2]});
        const float d = c.z + 1e-9f, dep = depth[(long)b * hw + p];
        o[0] = c.x / d * dep; o[1] = c.y / d * dep; o[2] = c.z / d * dep; o[3] = 1.0f;
    } else if (MODE == 1 || MODE == 2) {
        const V4 r = xform<true>(M + (MODE == 1 ? 34 : 18), {q[0], q[1], q[2], q[3]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    } else {
        const V3 r = cam2img<true>(M, {q[0], q[1], q[2], q[3]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    }
}

}  // namespace

extern "C" int effi_fusion_vis_filter_f32(const float* ref_depth, const float* reproj_xyd, int n, int n_src, int h, int w,
                                          float dist_base, float rel_diff_base, int thres_view, int relative, unsigned char* masks,
                                          effi_stream_t stream) {
    if (!ref_depth || !reproj_xyd || !masks || n < 1 || n_src < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    if (thres_view < 0 || thres_view > n_src || dist_base <= 0.0f || rel_diff_base <= 0.0f || (long)n * n_src > 65535) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    hipLaunchKernelGGL(fusion_vis_filter_kernel, dim3(effi_cdiv((long)h * w, 256), n * n_src), dim3(256), 0, st, ref_depth, reproj_xyd,
                       n_src, h, w, dist_base, rel_diff_base, thres_view, relative, masks);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_fusion_points_f32(int mode, const float* in, long in_batch_stride, const float* depth, const float* cams, int n,
                                      int h, int w, float* mats_scratch, float* out, effi_stream_t stream) {
    if (!in || !cams || !mats_scratch || !out || n < 1 || n > 65535 || h < 1 || w < 1 || mode < 0 || mode > 3) return EFFI_ERR_BADARG;
    if (mode == 0 && !depth) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    const int hw = h * w;
    hipLaunchKernelGGL(fusion_prepare_batch_kernel, dim3(effi_cdiv(n, 64)), dim3(64), 0, st, cams, n, mats_scratch);
    EFFI_LAUNCH_CHECK();
    const dim3 grid(effi_cdiv(hw, 256), n), blk(256);
    switch (mode) {
        case 0: hipLaunchKernelGGL(fusion_points_kernel<0>, grid, blk, 0, st, in, in_batch_stride, depth, mats_scratch, hw, out); break;
        case 1: hipLaunchKernelGGL(fusion_points_kernel<1>, grid, blk, 0, st, in, in_batch_stride, depth, mats_scratch, hw, out); break;
        case 2: hipLaunchKernelGGL(fusion_points_kernel<2>, grid, blk, 0, st, in, in_batch_stride, depth, mats_scratch, hw, out); break;
        default: hipLaunchKernelGGL(fusion_points_kernel<3>, grid, blk, 0, st, in, in_batch_stride, depth, mats_scratch, hw, out); break;
    }
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_fusion_dtu_reproject_f32(const float* ref_depth, const float* src_depth, const float* ref_cam, const float* src_cam,
                                             int h, int w, int s, int e, float dist_base, float diff_base, float* mats_scratch,
                                             float* out5, unsigned char* masks, effi_stream_t stream) {
    if (!ref_depth || !src_depth || !ref_cam || !src_cam || !mats_scratch || !out5 || h < 2 || w < 2) return EFFI_ERR_BADARG;
    if (masks && (s < 1 || e <= s || dist_base <= 0.0f || diff_base <= 0.0f)) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    hipLaunchKernelGGL(fusion_dtu_prepare_kernel, dim3(1), dim3(64), 0, st, ref_cam, src_cam, 1, mats_scratch);
    EFFI_LAUNCH_CHECK();
    hipLaunchKernelGGL(fusion_dtu_reproject_kernel, dim3(effi_cdiv((long)h * w, 256)), dim3(256), 0, st, ref_depth, src_depth, h, w,
                       mats_scratch, s, e, dist_base, diff_base, out5, masks);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_fusion_dtu_filter_f32(const float* ref_depth, const float* src_depths, const float* ref_cam, const float* src_cams,
                                          int n_src, int h, int w, const float* confidence, float conf_threshold, float conf_keep,
                                          int s, int e, float dist_base, float diff_base, float* mats_scratch, float* out_depth,
                                          unsigned char* out_photo_mask, unsigned char* out_geo_mask, unsigned char* out_final_mask,
                                          float* out_points, effi_stream_t stream) {
    if (!ref_depth || !src_depths || !ref_cam || !src_cams || !mats_scratch || !out_depth) return EFFI_ERR_BADARG;
    if (n_src < 1 || h < 2 || w < 2 || s < 1 || e <= s || dist_base <= 0.0f || diff_base <= 0.0f) return EFFI_ERR_BADARG;
    if (n_src > FUS_MAX_VIEWS || e - s > DTU_MAX_THR) return EFFI_ERR_UNSUPPORTED;
    hipStream_t st = effi_s(stream);
    hipLaunchKernelGGL(fusion_dtu_prepare_kernel, dim3(1), dim3(64), 0, st, ref_cam, src_cams, n_src, mats_scratch);
    EFFI_LAUNCH_CHECK();
    hipLaunchKernelGGL(fusion_dtu_filter_kernel, dim3(effi_cdiv((long)h * w, 256)), dim3(256), 0, st, ref_depth, src_depths, n_src, h, w,
                       mats_scratch, confidence, conf_threshold, conf_keep, s, e, dist_base, diff_base, out_depth, out_photo_mask,
                       out_geo_mask, out_final_mask, out_points);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}

extern "C" int effi_fusion_dynamic_filter_f32(const float* ref_depth, const float* src_depths, const float* ref_cam,
                                              const float* src_cams, int n_src, int h, int w, const float* ref_conf, int conf_h,
                                              int conf_w, float prob_threshold, int dh_view_num, float dist_base,
                                              float rel_diff_base, int relative, float* mats_scratch, float* out_depth,
                                              unsigned char* out_geo_mask, unsigned char* out_prob_mask,
                                              unsigned char* out_mask, float* out_points, float* out_reproj_xyd,
                                              effi_stream_t stream) {
    if (!ref_depth || !src_depths || !ref_cam || !src_cams || !mats_scratch || !out_depth) return EFFI_ERR_BADARG;
    if (n_src < 1 || h < 2 || w < 2 || dh_view_num < 1 || dh_view_num > n_src || dist_base <= 0.0f || rel_diff_base <= 0.0f)
        return EFFI_ERR_BADARG;
    if (n_src > FUS_MAX_VIEWS) return EFFI_ERR_UNSUPPORTED;
    if (ref_conf && (conf_h < 1 || conf_w < 1)) return EFFI_ERR_BADARG;
    hipStream_t st = effi_s(stream);
    hipLaunchKernelGGL(fusion_prepare_kernel, dim3(1), dim3(64), 0, st, ref_cam, src_cams, n_src, mats_scratch);
    EFFI_LAUNCH_CHECK();
    hipLaunchKernelGGL(fusion_dynamic_filter_kernel, dim3(effi_cdiv((long)h * w, 256)), dim3(256), 0, st, ref_depth, src_depths,
                       n_src, h, w, mats_scratch, ref_conf, conf_h, conf_w, prob_threshold, dh_view_num, dist_base, rel_diff_base,
                       relative, out_depth, out_geo_mask, out_prob_mask, out_mask, out_points, out_reproj_xyd);
    EFFI_LAUNCH_CHECK();
    return EFFI_OK;
}
