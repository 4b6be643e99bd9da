/*
 * effi_mvs_hip.h -- C ABI of the MI355X (gfx950) kernels for the Effi-MVS+ cost-volume hot path.
 *
 * The reference (bdwsq1996/Effi-MVS-plus) is pure Python/PyTorch: it has no FFI of its own.  What a
 * binding for this path would bind is the set of operator call sites listed in SURVEY.md section 2.1
 * (K1..K11); every entry point below names the reference lines it replaces.  Paths are relative to
 * the reference repository root.
 *
 * Conventions (all entry points):
 *   - plain C, no torch / HIP types in the signatures; `stream` is a hipStream_t passed as void*.
 *   - every pointer is a DEVICE pointer owned by the caller unless the comment says "host".
 *   - returns 0 (EFFI_OK) or a negative error code; never throws, allocates, frees or synchronises.
 *   - re-entrant; work is enqueued on `stream` of the current device; graph-capture safe.
 *   - tensors are fp32, batch size 1 per call (the host loops over the batch); layouts are spelled
 *     out per argument.  "planar" = [C][H][W] (torch NCHW with N=1), "nhwc" = [H][W][C].
 */
#ifndef EFFI_MVS_HIP_H
#define EFFI_MVS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EFFI_OK               0
#define EFFI_ERR_BADARG      -1   /* null pointer, non-positive size, unsupported channel count ... */
#define EFFI_ERR_UNSUPPORTED -2   /* shape outside what the kernels are instantiated for */
#define EFFI_ERR_LAUNCH      -3   /* hipPeekAtLastError() != hipSuccess after the launch (the error is left in place for the host framework) */
#define EFFI_ERR_WORKSPACE   -4   /* the current device has no workspace registered (effi_set_workspace) */

#define EFFI_MAX_VIEWS 12         /* source views per call (T&T script uses num_view=11 -> 10 sources) */
#define EFFI_MAX_SRC    3         /* concatenated input tensors of one 2-D / 3-D convolution */
#define EFFI_VIEW_TABLE_ROW (EFFI_MAX_VIEWS + 2)   /* pointers per row of a view table: reference, sources, nulls */
#define EFFI_VIEW_TABLE_MAX 64    /* pointers one effi_view_table_set call writes (4 rows) */

typedef void* effi_stream_t;

/* activation / epilogue selectors for effi_conv2d_f32 */
#define EFFI_ACT_NONE    0
#define EFFI_ACT_RELU    1
#define EFFI_ACT_SIGMOID 2
#define EFFI_ACT_TANH    3
#define EFFI_EPI_PLAIN   0   /* out0 = act(conv + bias) */
#define EFFI_EPI_GRU_ZR  1   /* channels [0,hd): out0 = sigmoid(.) (= z);  [hd,2hd): out1 = sigmoid(.) * aux0 (= r*h) */
#define EFFI_EPI_GRU_Q   2   /* q = tanh(.);  out0 = (1 - aux1) * aux0 + aux1 * q   (aux0 = h, aux1 = z) */
#define EFFI_EPI_Q4      0x100 /* flag, split-resident GRU epilogues only (effi_conv2d_k3_bf16x3_sr: EFFI_EPI_GRU_ZR | EFFI_EPI_Q4,
                               * EFFI_EPI_GRU_Q | EFFI_EPI_Q4): the fp32 maps of the call (aux0 = h, aux1 = z, out0) are laid out
                               * [channels/4][h][w][4] ("Q4") instead of planar [channels][h][w] -- a lane's 4 channels of a pixel are
                               * ONE 16-byte access instead of four 4-byte ones a plane apart.  Same values. */
#define EFFI_EPI_HEAD    3   /* 1 channel: out0 = aux0 + tanh(.) (inverse depth);  out1 = 1/clamp(lo+(hi-lo)*out0, 1e-4) */
#define EFFI_EPI_NHWC    5   /* out0 = act(conv + bias) written channel-last [h][w][cout] (what the warp kernels read) */
#define EFFI_EPI_ADD_SHUF2 9        /* out0 = conv + bias + pixel_shuffle(aux0): aux0 planar [4*cout][h/2][w/2], channel
                                      ((y&1)*2 + (x&1))*cout + co at (y>>1, x>>1); split-precision 3x3 entry only, cout <= 32 */
#define EFFI_EPI_NHWC_ADD_SHUF2 10  /* the same, written channel-last [h][w][cout] */
#define EFFI_EPI_ADD_UP2 4   /* out0 = act(conv + bias) + nearest_upsample_x2(aux0), aux0 planar [cout][h/2][w/2]
                              * (feature pyramid top-down path, models/module.py:403,407) */

int effi_version(void);
const char* effi_error_string(int code);

/* ---- caller-owned per-device workspace ----------------------------------------------------------
 * The library allocates nothing.  The split-precision convolution kernels read their zero padding from a small
 * block of device memory instead of masking every load; the caller provides that block once per device:
 *   effi_workspace_bytes()            bytes a device's workspace must hold (a constant of this build);
 *   effi_set_workspace(dev, p, n)     registers `p` (device memory of device `dev`, >= effi_workspace_bytes() bytes,
 *                                     ZERO-FILLED by the caller and never written by the library) for ordinal `dev`;
 *                                     p = NULL unregisters.  The caller keeps ownership and must keep it alive while
 *                                     kernels of that device may run.  Thread-safe against concurrent launches on
 *                                     other devices (one slot per device ordinal, up to EFFI_MAX_DEVICES).
 *   effi_get_workspace(dev)           the registered pointer or NULL.
 * Entry points that need it look up the slot of the CURRENT device (hipGetDevice) and return EFFI_ERR_WORKSPACE
 * when it is empty: nothing is allocated lazily, so a first call inside stream capture or from a second device of
 * the process (nn.DataParallel, test_dtu_dypcd.py:418) is safe. */
#define EFFI_MAX_DEVICES 64
long effi_workspace_bytes(void);
int effi_set_workspace(int device, void* workspace, long bytes);
const void* effi_get_workspace(int device);

/* ---- K11: projection algebra ------------------------------------------------------------------
 * models/Effi_MVS_plus.py:34-37,217-220 (K.[R|t]) and models/module.py:314-316 (src . ref^-1).
 * pairs: [n_views][2][4][4] ([v][0] = extrinsic, [v][1][:3][:3] = intrinsic); view 0 = reference.
 * rt_out: [n_views-1][12] = 9 row-major rotation entries then 3 translation entries of
 * P_src . P_ref^-1.  Computed in fp64 on the device, rounded once to fp32 (numerics). */
int effi_compose_rel_proj_f32(const float* pairs, int n_views, float* rt_out, effi_stream_t stream);
/* Same for matrices that are already composed (the argument form of homo_warping_new,
 * models/module.py:303-316).  src_proj, ref_proj: [4][4]; rt_out: [12]. */
/* effi_compose_rel_proj_f32 for all stages of the cascade in one launch: pairs = HOST array of n_stages (<= 4) device
 * pointers, rt_out [n_stages][n_views-1][12]. */
int effi_compose_rel_proj_stages_f32(const float* const* pairs, int n_stages, int n_views, float* rt_out,
                                     effi_stream_t stream);
int effi_rel_proj_f32(const float* src_proj, const float* ref_proj, float* rt_out, effi_stream_t stream);

/* ---- layout: planar [C][HW] -> nhwc [HW][C] for n tensors (feature maps arrive NCHW from the FPN,
 * models/module.py:400-409).  srcs/dsts: HOST arrays of n device pointers, n <= EFFI_MAX_VIEWS+1. */
int effi_planar_to_nhwc_f32(const float* const* srcs, float* const* dsts, int n, int C, int HW,
                            effi_stream_t stream);

/* ---- K1/K2: homography warp, full warped volume (API parity for homo_warping_new,
 * models/module.py:303-344; the fused kernels below never materialise this volume).
 * src_nhwc [h][w][C]; rt [12]; depth: hypothesis d of pixel p at depth[d*depth_dstride + p*depth_pstride]
 * (pstride 0 = same hypotheses for every pixel); out planar [C][D][h][w].  C in {8,16,32}. */
int effi_homo_warp_f32(const float* src_nhwc, const float* rt, const float* depth, long depth_dstride,
                       long depth_pstride, int C, int h, int w, int D, float* out, effi_stream_t stream);

/* ---- K1+K2+K3 (+ entropy of K4), stage 1: per-view correlation volume.
 * models/Effi_MVS_plus.py:38-40 (warp, mean_c(warped*ref)), :43-44 (softmax over D, entropy).
 * ref_nhwc [h][w][C]; src_nhwc: HOST array of S device pointers; rt [S][12]; depth as above;
 * sim_views [S][D][h][w]; entropy [S][h][w]. */
int effi_warpcorr_views_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                            const float* depth, long depth_dstride, long depth_pstride,
                            int C, int h, int w, int D, float* sim_views, float* entropy,
                            effi_stream_t stream);

/* View-table form of effi_warpcorr_views_f32 (round 4; the evaluation-set runner, datasets/general_eval.py:26-51 +
 * test_dtu_dypcd.py:424-439: every reference view is a forward over ITS (reference, sources) subset of a scan's images).
 * view_table_dev: DEVICE array of EFFI_VIEW_TABLE_ROW pointers -- [0] the reference map, [1..S] the sources, the rest null -- read by
 * the kernel when it runs, not when it is enqueued: a captured hipGraph of the hot path is pointed at another item's cached feature
 * maps by rewriting the table (effi_view_table_set) instead of copying ~150 MB of maps into static inputs.  Same kernels, same
 * arithmetic, bitwise the same result as the pointer form. */
int effi_warpcorr_views_tbl_f32(const float* const* view_table_dev, int S, const float* rt, const float* depth,
                                long depth_dstride, long depth_pstride, int C, int h, int w, int D, float* sim_views,
                                float* entropy, effi_stream_t stream);
/* n (<= EFFI_VIEW_TABLE_MAX) device pointers, given in a HOST array, written to table_dev[0..n) by a one-workgroup kernel on
 * `stream` (stream-ordered like every other entry; no host memory is read after the call returns). */
int effi_view_table_set(const void** table_dev, const void* const* ptrs, int n, effi_stream_t stream);

/* ---- K4: view-weight net, fused 3x(3x3 conv+BN+ReLU) -> 1x1 conv -> sigmoid.
 * models/Effi_MVS_plus.py:361-362, models/module.py:213-220.  BN is folded by the host.
 * entropy [n][h][w] -> weight [n][h][w].  params: packed fp32 block, every 3x3 layer stored
 * [cin][tap][cout]:  w0[9][16] b0[16] | w1[16][9][16] b1[16] | w2[16][9][8] b2[8] | w3[8] b3[1]. */
int effi_pixelwise_net_f32(const float* entropy, const float* params, int n, int h, int w, float* weight,
                           effi_stream_t stream);

/* ---- K3: view-weighted aggregation  sim = sum_v sim_v*w_v / (sum_v w_v + 1e-6).
 * models/Effi_MVS_plus.py:48-53,67.  sim_views [S][D][hw]; weights [S][hw]; out [D][hw].
 * weights == NULL: the unweighted branch (pixel_wise_net = None, :55-58,70): sim = sum_v sim_v / S. */
int effi_view_aggregate_f32(const float* sim_views, const float* weights, int S, int D, int hw, float* out,
                            effi_stream_t stream);

/* ---- K1+K2+K3 + hypothesis generation, stages 2/3: dynamic correlation volume.
 * models/Effi_MVS_plus.py:184-251 (GetCost_initvolume.forward, Inverse=True) and
 * models/module.py:554-570 (get_cur_depth_range_samples, in inverse depth).
 * cur_depth [h][w]; interval: device pointer to ONE float (inverse-depth step);
 * view_w [S][h>>vw_shift][w>>vw_shift] (nearest-upsampled on the fly: F.interpolate nearest,
 * models/Effi_MVS_plus.py:497); sim [D][h][w]; samples [D][h][w] (depth hypotheses). */
int effi_warpcorr_dyn_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                          const float* cur_depth, const float* interval, const float* view_w, int vw_shift,
                          int C, int h, int w, int D, float* sim, float* samples, effi_stream_t stream);

/* View-table form of effi_warpcorr_dyn_f32 (see effi_warpcorr_views_tbl_f32). */
int effi_warpcorr_dyn_tbl_f32(const float* const* view_table_dev, int S, const float* rt, const float* cur_depth,
                              const float* interval, const float* view_w, int vw_shift, int C, int h, int w, int D,
                              float* sim, float* samples, effi_stream_t stream);

/* ---- K5/K6: 3-D convolutions, kernel 3, padding 1, BN folded into (weight, bias) by the host.
 * models/module.py:124-160 (Conv3d), :168-203 (Deconv3d), :439-452, :505-508.
 * Input = channel concatenation of n_src planar tensors [Ci][D][h][w] (torch.cat, module.py:513).
 * weight: [cin][kd][ky][kx][cout] (host-packed from [cout][cin][3][3][3]); bias [cout] or NULL.
 * stride (sz,sxy,sxy) with sz in {1,2}, sxy in {1,2}; out planar [cout][Do][ho][wo],
 * Do = (D-1)/sz+1 etc.  out = relu?(conv + bias) (+ skip, added AFTER the ReLU: module.py:460-461). */
int effi_conv3d_k3_f32(const float* const* srcs, const int* src_channels, int n_src,
                       const float* weight, const float* bias, int cout,
                       int D, int h, int w, int sz, int sxy, int relu, const float* skip,
                       float* out, effi_stream_t stream);
/* Same operator for the low-resolution levels of the regulariser (cout in {16,32}, stride 1): every output plane z is
 * a 2-D convolution over the channel-concatenated planes z-1, z, z+1 and runs on the fp32 matrix cores
 * (models/module.py:443,446).  in planar [cin][D][h][w]; wpack = MFMA packing (see effi_conv2d_f32) of the weight viewed
 * as [cout][3*cin][3][3] with input channel index kd*cin + ci; bias [cout]; out planar [cout][D][h][w]. */
int effi_conv3d_k3s1_mfma_f32(const float* in, int cin, const float* wpack, const float* bias, int cout,
                              int D, int h, int w, int relu, float* out, effi_stream_t stream);
/* Stride-(2,2,2) form of the same operator (models/module.py:442,445), cout in {16,32}: in planar [cin][D][h][w],
 * out planar [cout][(D-1)/2+1][(h-1)/2+1][(w-1)/2+1]; same weight packing. */
int effi_conv3d_k3s2_mfma_f32(const float* in, int cin, const float* wpack, const float* bias, int cout,
                              int D, int h, int w, int relu, float* out, effi_stream_t stream);
/* 3x3 convolution (split precision, no activation) followed IN THE SAME KERNEL by the 1x1 convolution that consumes it
 * together with c_extra further channels -- convd then convc of the update block's encoder, models/update.py:78-80,93-96:
 * and the mask head's 3x3 + ReLU + 1x1, models/update.py:112-114:
 * out [cout2][h][w] = relu?(bias2 + W2 . cat(relu1?(conv3x3(srcs) + bias) [cout1], extra [c_extra][h][w])).
 * wpack_bf16 / bias as effi_conv2d_k3_bf16x3_f32 (cout1 <= 96); w2pack_bf16 = packing.pack_conv1x1_after
 * ([ceil(cout2/16)][ceil(cout1/16)+1][hi|lo][64][4] bf16: A fragments of v_mfma_f32_16x16x16_bf16); bias2 padded to
 * 16*ceil(cout2/16); cout2 <= 96, c_extra <= 16, w % 4 == 0; the intermediate never reaches HBM. */
int effi_conv2d_k3_k1_bf16x3_f32(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                 const float* bias, int cout1, int relu1, const float* extra, int c_extra,
                                 const void* w2pack_bf16, const float* bias2, int cout2, int relu, int h, int w, float* out,
                                 effi_stream_t stream);
/* Two independent 3x3 convolutions of the same shape (cout <= 64, h, w, activation; EPI_PLAIN) in ONE launch, each with
 * its own sources / weights / bias / output as in effi_conv2d_k3_bf16x3_f32: convc2 and convd2 of the encoder
 * (models/update.py:87,91), which do not depend on each other. */
int effi_conv2d_k3_bf16x3_pair_f32(const float* const* srcs_a, const int* src_channels_a, int n_src_a, const void* wpack_a,
                                   const float* bias_a, float* out_a, const float* const* srcs_b,
                                   const int* src_channels_b, int n_src_b, const void* wpack_b, const float* bias_b,
                                   float* out_b, int cout, int h, int w, int act, effi_stream_t stream);
/* 5x5 / stride 2 / padding 2 convolution (+bias, activation) in split precision: the pyramid's down-sampling layers
 * (models/module.py:359,365,370), same contract as effi_conv2d_k5s2_f32 with win % 4 == 0.  wpack_bf16 =
 * [ceil(cin/8)][G][7][NT][hi|lo][64][8] bf16 with G = groups of NT output tiles (NT = 1 for cout <= 16, else 2; G = ceil(cout/32)),
 * K item = tap ky*5 + kx of an 8-channel chunk, lane = q*16 + j holds W[16(g NT + n) + j][8 chunk + e][tap = 4 s + q]
 * (packing.pack_conv2d_k5s2_bf16x3); bias padded to 16 * NT * G. */
int effi_conv2d_k5s2_bf16x3_f32(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int hin, int win,
                                int act, float* out, effi_stream_t stream);
/* Stride-(2,2,2) 3-D convolution k3 p1 (+folded BN, ReLU) in split precision: the U-Net's down-sampling levels
 * (models/module.py:442,445), contract of effi_conv3d_k3s2_mfma_f32 with w % 4 == 0 and cout <= 64.  wpack_bf16 =
 * [3 * ceil(cin/8)][G][3][NT][hi|lo][64][8] bf16: chunk = (kd, octet), K item = tap ky*3 + kx, groups of NT output tiles as in
 * effi_conv2d_k5s2_bf16x3_f32 (packing.pack_conv3d_s2_bf16x3). */
int effi_conv3d_k3s2_bf16x3_f32(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int D, int h, int w,
                                int relu, float* out, effi_stream_t stream);
/* Two chained 3x3 convolutions with ReLU after each, at most 8 channels into each (the pyramid's full-resolution block conv0 =
 * Conv2d(3, 8) -> Conv2d(8, 8) with BatchNorm folded, models/module.py:353-356), in one kernel: the 8-channel intermediate map
 * stays in LDS (first layer computed on each 16 x 16 tile grown by one pixel).  in [cin][h][w], cin <= 8; both layers have at most 8
 * outputs, out [cout][h][w]; w1 / w2 = ROW-PAIR operands [3][hi|lo][64][8] bf16 (packing.pack_conv2d_bf16x3_oct: MFMA rows 0-7 = the
 * channels of an image row, rows 8-15 = the same channels of the row below, K = the 12 taps of the 4 x 3 window both rows see),
 * biases padded to 16; w % 4 == 0. */
int effi_conv2d_k3_twice_bf16x3_f32(const float* in, int cin, const void* w1_bf16, const float* bias1, const void* w2_bf16,
                                    const float* bias2, int cout, int h, int w, float* out, effi_stream_t stream);
/* Tail of the update block's encoder in one kernel (models/update.py:87-96): cor = relu(convc2(cor1)), dfm = relu(convd2(dfm1)),
 * out = relu(convc(cat(convd(cat(cor, dfm)), extra))) with extra = the context map -- the launches
 * effi_conv2d_k3_bf16x3_pair_f32 + effi_conv2d_k3_k1_bf16x3_f32 with the two intermediate maps kept in LDS (12 x 16 output tiles,
 * first layers computed on the tile grown by one pixel).  Same arithmetic, bitwise the result of the two launches.
 * wc2 / wd2 [hd -> hd] and wd [2 hd -> cmix] as packing.pack_conv2d_bf16x3, biases padded to multiples of 16; w2pack / bias2 as
 * packing.pack_conv1x1_after(convc, cmix, c_extra).  hd == 16, cmix <= 16, c_extra <= 16, w % 4 == 0, else EFFI_ERR_UNSUPPORTED. */
int effi_encoder_tail_bf16x3_f32(const float* cor1, const float* dfm1, int hd, const void* wc2_bf16, const float* bias_c2,
                                 const void* wd2_bf16, const float* bias_d2, const void* wd_bf16, const float* bias_d, int cmix,
                                 const float* extra, int c_extra, const void* w2pack_bf16, const float* bias2, int cout2, int h,
                                 int w, float* out, effi_stream_t stream);
/* Mask head + convex upsampling of the last GRU iteration in one kernel (models/update.py:109-112,136-138: 3x3 conv, ReLU, 1x1
 * conv to 36 channels, x0.25 folded into w2pack / bias2; upsample_depth, models/Effi_MVS_plus.py:167-178; scale_inv_depth
 * :138-148): the 36 mask values of a pixel stay in registers.  inv_depth [h][w]; out_depth [2h][2w]; out_depth_inv [2h][2w] or
 * NULL = depth_to_disp(out_depth) (what the next stage starts from); cout1 in {32, 64, 96}; w % 4 == 0.  w2pack / bias2 hold 48
 * rows: row 16 t + 4 q + r = mask channel 4 (4 t + r) + q, i.e. (tap 4 t + r, sub-pixel q), zero for taps 9..11
 * (packing.pack_mask_taps_per_lane): one lane owns the nine taps of one sub-pixel, softmax and weighted sum in tap order. */
int effi_conv2d_k3_k1_up2x_bf16x3_f32(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                      const float* bias, int cout1, const void* w2pack_bf16, const float* bias2,
                                      const float* inv_depth, const float* disp_range, int n_range, int h, int w,
                                      float* out_depth, float* out_depth_inv, effi_stream_t stream);
/* Same operator (stride 1, cout <= 32, any w: row ends that cut a pixel quad are staged element by element) in split precision: products as hi*hi + hi*lo + lo*hi on the bf16
 * matrix cores with fp32 accumulation (see effi_conv2d_k3_bf16x3_f32).  Input = channel concatenation of n_src planar
 * tensors [Ci][D][h][w] (models/module.py:513); wpack_bf16 = split-bf16 packing of the weight viewed as
 * [cout][3*cin][3][3] with input channel index kd*cin + ci; bias [16*ceil(cout/16)]; out planar [cout][D][h][w]. */
int effi_conv3d_k3s1_bf16x3_f32(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                const float* bias, int cout, int D, int h, int w, int relu, float* out,
                                effi_stream_t stream);
/* Same operator for cin in {8, 16} (one or two sources, the first with a multiple of 8 channels), cout <= 32, w % 4 == 0,
 * with a rolling window of input planes: a workgroup walks a run of output planes and fetches each input plane once
 * instead of three times.  wpack_bf16 = [ceil(27*cin/32)][ceil(cout/16)][hi|lo][64][8] bf16 with K index
 * ((kd*9 + ky*3 + kx) * cin/8 + ci/8) * 8 + ci%8 (packing.pack_conv3d_roll_bf16x3); bias [16*ceil(cout/16)].
 * cout <= 8 takes the ROW-PAIR operand instead (the same packing function emits it): [9*cin/8][1][hi|lo][64][8] with K index
 * (((kd*4 + dy)*3 + kx) * cin/8 + ci/8) * 8 + ci%8 over a 4 x 3 window of taps, MFMA rows 0-7 = W[co][ci][kd][dy][kx] (zero at dy = 3)
 * for image row y, rows 8-15 = W[co][ci][kd][dy-1][kx] (zero at dy = 0) for row y + 1: two output rows per tile, 36 % fewer MFMAs. */
int effi_conv3d_k3s1_roll_bf16x3_f32(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                     const float* bias, int cout, int D, int h, int w, int relu, float* out,
                                     effi_stream_t stream);
/* Transposed 3-D convolution, kernel 3, padding 1, stride (sz,2,2), output_padding (sz-1,1,1):
 * out dims (sz*D, 2h, 2w).  models/module.py:448-450 (sz=2), :508 (sz=1).
 * in planar [cin][D][h][w]; weight [cin][kd][ky][kx][cout] (host-packed from torch's
 * [cin][cout][3][3][3]); out = relu?(deconv + bias) (+ skip). */
int effi_deconv3d_k3_f32(const float* in, int cin, const float* weight, const float* bias, int cout,
                         int D, int h, int w, int sz, int relu, const float* skip,
                         float* out, effi_stream_t stream);
/* The stride-(2,2,2) case with cin % 16 == 0, cout <= 16 on the bf16 matrix cores in split precision (one GEMM per input
 * position over output parity x input neighbour, models/module.py:448-450).  wpack_bf16 =
 * packing.pack_deconv3d_s2_bf16x3: [cin/16][18][hi|lo][64][8] bf16; bias [16]; out = relu?(deconv + bias) (+ skip). */
int effi_deconv3d_k3s2_bf16x3_f32(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout,
                                  int D, int h, int w, int relu, const float* skip, float* out, effi_stream_t stream);

/* ---- K7: softmax over D, soft-argmin depth, 4-window confidence.
 * models/Effi_MVS_plus.py:79-88, models/module.py:518-524.
 * logits [D][hw]; depth hypotheses as in effi_homo_warp_f32; out_depth [hw]; out_conf [hw].
 * out_depth_inv (or NULL; then disp_range may be NULL): depth_to_disp(out_depth) on the global hypothesis range
 * disp_range[0] .. disp_range[n_range-1] -- the normalised inverse depth the stage's update block starts from (:538). */
int effi_softmax_regress_conf_f32(const float* logits, const float* depth, long depth_dstride,
                                  long depth_pstride, int D, int hw, float* out_depth, float* out_conf,
                                  const float* disp_range, int n_range, float* out_depth_inv, effi_stream_t stream);

/* ---- K8: 1-D volume lookup (pro_bilinear_sampler), models/Effi_MVS_plus.py:102-134,151-164.
 * vol: value k of pixel p at vol[k*vol_dstride + p*vol_pstride], Dp entries per pixel, (h,w) pixels.
 * query: depth q of pixel (y,x) at query[q*q_dstride + y*q_ystride + x*q_xstride]  (strides let the
 * caller read a nearest-downsampled view of a finer map, models/Effi_MVS_plus.py:514).
 * dmin/dmax: depth range of the volume, element of pixel p at dmin[p*range_pstride] (0 = global).
 * out [nq][h][w]. */
int effi_vol_lookup1d_f32(const float* vol, long vol_dstride, long vol_pstride, int Dp,
                          const float* query, long q_dstride, long q_ystride, long q_xstride, int nq,
                          const float* dmin, const float* dmax, long range_pstride,
                          int h, int w, float* out, effi_stream_t stream);

/* bilinear_sampler, models/Effi_MVS_plus.py:102-117 (the public wrapper pro_bilinear_sampler calls): img [N][C][1][W] rows sampled
 * at pixel coordinates coords [N][nq][2] = (x, y) (linear in x, zeros outside, align_corners=True; H == 1, so y only enters the
 * optional mask) -> out [N][C][nq]; mask (or NULL) [N][nq] = 1 where -1 < 2x/(W-1)-1 < 1 and -1 < y < 1 (:113). */
int effi_bilinear_sampler1d_f32(const float* img, int N, int C, int W, const float* coords, long nq, float* out, float* mask,
                                effi_stream_t stream);

/* ---- K8 inside the GRU loop: GetCost.forward, models/Effi_MVS_plus.py:257-303 with
 * models/Effi_MVS_plus.py:138-148 (scale_inv_depth) applied first.
 * inv_depth [h][w]: normalised inverse depth (input_is_depth = 0; scale_inv_depth is applied in the
 * kernel) or the depth map itself (input_is_depth = 1, the argument form of GetCost.forward);
 * disp_range: device pointer to depth_values[0..n-1] (ascending inverse depths, uses [0] and
 * [n_range-1]; unused when input_is_depth); interval: device pointer to one float;
 * cur_vol / reg_vol with strides as above; dmin/dmax as above; cost out [2*nq][h][w]
 * = cat(lookup(cur_vol), lookup(reg_vol)) at nq hypotheses (inv +/- (nq/2)*interval). */
int effi_getcost_f32(const float* inv_depth, const float* disp_range, int n_range, int input_is_depth,
                     const float* interval,
                     const float* cur_vol, long cur_dstride, long cur_pstride, int Dcur,
                     const float* reg_vol, long reg_dstride, long reg_pstride, int Dreg,
                     const float* dmin, const float* dmax, long range_pstride, int nq,
                     int h, int w, float* cost, effi_stream_t stream);

/* Same lookup fused with the encoder's 1x1 convolution that consumes it (convc1 + ReLU, models/update.py:73,86):
 * out [cout][h][w] = relu?(bias + W . cost), weight [2*nq][cout] (transposed from torch's [cout][2*nq][1][1]),
 * nq in {2,3,4}, cout % 8 == 0.  The cost map itself is not materialised. */
int effi_getcost_conv1x1_f32(const float* inv_depth, const float* disp_range, int n_range, int input_is_depth,
                             const float* interval,
                             const float* cur_vol, long cur_dstride, long cur_pstride, int Dcur,
                             const float* reg_vol, long reg_dstride, long reg_pstride, int Dreg,
                             const float* dmin, const float* dmax, long range_pstride, int nq,
                             int h, int w, const float* weight, const float* bias, int cout, int relu,
                             float* out, effi_stream_t stream);

/* Both inputs of the update block's encoder for one GRU iteration in ONE launch (models/update.py:86,90): out_c1 =
 * relu(convc1(GetCost(inv_depth))) exactly as effi_getcost_conv1x1_f32 (normalised inverse depth in, nq == 3), and out_d1 =
 * relu(convd1(inv_depth)) exactly as effi_conv2d_c1k7_relu_f32 (weight_d1 [49][cout], cout in {16,32,48}); the two share the
 * grid, so neither a second stream nor its fork / join is needed to overlap them. */
int effi_encoder_inputs_f32(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                            const float* cur_vol, long cur_dstride, long cur_pstride, int Dcur,
                            const float* reg_vol, long reg_dstride, long reg_pstride, int Dreg,
                            const float* dmin, const float* dmax, long range_pstride, int nq, int h, int w,
                            const float* weight_c1, const float* bias_c1, const float* weight_d1, const float* bias_d1,
                            int cout, float* out_c1, float* out_d1, effi_stream_t stream);
/* The same with the 7x7 convolution on the matrix cores in split precision (hi*hi + lo*hi + hi*lo bf16 MFMAs, fp32 accumulation; one
 * workgroup per 32 x 8 pixel tile computes all channels): what the default ("split") precision uses. */
int effi_encoder_inputs_bf16x3_f32(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                            const float* cur_vol, long cur_dstride, long cur_pstride, int Dcur,
                            const float* reg_vol, long reg_dstride, long reg_pstride, int Dreg,
                            const float* dmin, const float* dmax, long range_pstride, int nq, int h, int w,
                            const float* weight_c1, const float* bias_c1, const float* weight_d1, const float* bias_d1,
                            int cout, float* out_c1, float* out_d1, effi_stream_t stream);

/* ---- K9: 2-D convolutions of the update block on the fp32 MFMA path (v_mfma_f32_16x16x4_f32).
 * models/update.py:14-15,36-38,73-81,109-112.  ks in {1,3}, padding ks/2, stride 1.
 * Input = concatenation of n_src planar tensors.  wpack: host-packed
 * [ceil(cin/4)][ks*ks][ceil(cout/16)][64] (lane order of the MFMA B operand); bias [16*ceil(cout/16)].
 * cout == 1 && ks == 3 (depth head) runs on the vector ALUs and reads plain [cin][9] weights appended
 * to wpack at float offset ceil(cin/4)*9*64.
 * epilogue: EFFI_EPI_*; act used by EFFI_EPI_PLAIN / _ADD_UP2 / _NHWC; aux0/aux1/out1 per the EFFI_EPI_* comments;
 * disp_range (EPI_HEAD only): device pointer to depth_values, n_range entries. */
int effi_conv2d_f32(const float* const* srcs, const int* src_channels, int n_src,
                    const float* wpack, const float* bias, int cout, int ks, int h, int w,
                    int epilogue, int act, const float* aux0, const float* aux1,
                    const float* disp_range, int n_range,
                    float* out0, float* out1, effi_stream_t stream);
/* Split-precision variant of the 3x3 convolution (ks = 3, w % 4 == 0): operands are split x = hi + lo into two bf16
 * values and each product is evaluated as hi*hi + hi*lo + lo*hi on the bf16 matrix cores with fp32 accumulation
 * (~1e-5 relative error instead of 6e-8; ~3x the throughput of effi_conv2d_f32 on these layers).  Same arguments as
 * effi_conv2d_f32 except wpack_bf16: host-packed [ceil(cin/16)][5][ceil(cout/16)][hi|lo][64][8] bf16 (packing.py).
 * Epilogues: PLAIN, NHWC, GRU_ZR, GRU_Q. */
int effi_conv2d_k3_bf16x3_f32(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                              const float* bias, int cout, int h, int w, int epilogue, int act,
                              const float* aux0, const float* aux1, const float* disp_range, int n_range,
                              float* out0, float* out1, effi_stream_t stream);
/* 5x5, stride 2, padding 2 convolution (+ folded BN + activation) of the feature pyramid's down-sampling layers,
 * models/module.py:359,365,370.  in planar [cin][hin][win]; wpack: MFMA packing [ceil(cin/4)][25][ceil(cout/16)][64];
 * bias [16*ceil(cout/16)]; out planar [cout][(hin-1)/2+1][(win-1)/2+1] = act(conv + bias).  cout <= 64. */
int effi_conv2d_k5s2_f32(const float* in, int cin, const float* wpack, const float* bias, int cout,
                         int hin, int win, int act, float* out, effi_stream_t stream);
/* 7x7, one input channel (convd1, models/update.py:76,90): in [h][w]; weight [49][cout]
 * (host-packed), bias [cout]; out planar [cout][h][w] = relu(conv + bias).  cout in {16,32,48}. */
int effi_conv2d_c1k7_relu_f32(const float* in, const float* weight, const float* bias, int cout,
                              int h, int w, float* out, effi_stream_t stream);
/* The same on the matrix cores in split precision (see effi_encoder_inputs_bf16x3_f32). */
int effi_conv2d_c1k7_relu_bf16x3_f32(const float* in, const float* weight, const float* bias, int cout,
                              int h, int w, float* out, effi_stream_t stream);

/* ---- K10: convex upsampling x2 (upsample_depth, models/Effi_MVS_plus.py:167-178) fused with
 * scale_inv_depth (:138-148).  inv_depth [h][w]; mask [36][h][w] (already scaled by 0.25);
 * out_inv [2h][2w] (or NULL); out_depth [2h][2w] (or NULL, then disp_range may be NULL too);
 * out_depth_inv [2h][2w] (or NULL; needs out_depth): depth_to_disp(out_depth), what the NEXT stage starts from (:538). */
int effi_convex_upsample2x_f32(const float* inv_depth, const float* mask, const float* disp_range,
                               int n_range, int h, int w, float* out_inv, float* out_depth,
                               float* out_depth_inv, effi_stream_t stream);

/* ---- small fused element-wise steps of the stage loop --------------------------------------------
 * models/Effi_MVS_plus.py:445-450: hidden = tanh(ctx[:hd]), inp = relu(ctx[hd:hd+cd]); ctx [hd+cd][hw]. */
int effi_split_tanh_relu_f32(const float* ctx, int hd, int cd, int hw, float* hidden, float* inp,
                             effi_stream_t stream);
/* models/Effi_MVS_plus.py:538 (depth_to_disp with the global range): inv = (1/d - lo)/((hi-lo)+1e-10). */
/* effi_split_tanh_relu_f32 for all stages of the cascade in one launch: HOST arrays of n_stages (<= 4) device pointers / sizes. */
int effi_split_tanh_relu_stages_f32(const float* const* ctx, const int* hd, const int* cd, const int* hw,
                                    float* const* hidden, float* const* inp, int n_stages, effi_stream_t stream);
int effi_depth_to_inv_f32(const float* depth, const float* disp_range, int n_range, int n, float* inv,
                          effi_stream_t stream);
/* models/Effi_MVS_plus.py:464-474: D uniform hypotheses in inverse depth between disp_range[0] and
 * disp_range[n_range-1], returned as depths [D]; also writes intervals[0..4]: the per-stage
 * inverse-depth intervals ratio[s]*(hi-lo)/n_range for s<3 (:424,466,502), then depth_min_ = 1/hi
 * and depth_max_ = 1/lo (:413-414). */
int effi_stage1_hypotheses_f32(const float* disp_range, int n_range, int D, float* depths,
                               float* intervals, effi_stream_t stream);
/* nearest-neighbour integer upsampling of a planar map (F.interpolate nearest, :479,497):
 * in [C][h][w] -> out [C][h*f][w*f]. */
int effi_upsample_nearest_f32(const float* in, int C, int h, int w, int f, float* out, effi_stream_t stream);
/* Tail of DepthHead + the update of BasicUpdateBlock.forward (models/update.py:15,21,125-127) when conv2's nine 1x1 tap
 * projections of the hidden map were applied by the producer (effi_conv2d_k3_k1_bf16x3_f32 with cout2 = 9, relu1 = 1):
 * partial9 [9][h][w] (plane ky*3+kx = sum_c conv2.weight[0][c][ky][kx] * hid[c]) ->
 * out_inv = inv_depth + tanh(sum_tap partial9[tap][p + tap - 1] + bias2[0]) (zero outside the map = conv2's padding),
 * out_depth = 1 / clamp(lo + (hi - lo) * out_inv, 1e-4) as EFFI_EPI_HEAD.  w % 4 == 0. */
int effi_head_update_f32(const float* partial9, const float* bias2, const float* inv_depth, const float* disp_range,
                         int n_range, int h, int w, float* out_inv, float* out_depth, effi_stream_t stream);

/* ---- n3: dynamic geometric-consistency filter + depth averaging of the Tanks-and-Temples driver,
 * misc/fusion.py:117-181 (get_reproj_dynamic, vis_filter_dynamic) and the tensor part of test_tank.py:466-512, one reference
 * view per call.  ref_depth [h][w]; src_depths [n_src][h][w]; ref_cam [2][4][4] (extrinsic, intrinsic), src_cams
 * [n_src][2][4][4]; ref_conf [conf_h][conf_w] or NULL (then the photometric mask is all ones); thresholds i/dist_base and
 * i/rel_diff_base for i = dh_view_num .. n_src; relative != 0 divides the depth difference by ref_depth.
 * mats_scratch: >= 52*(n_src+1) floats of device memory (per-view K, K^-1, E, E^-1; filled by the call).
 * Outputs: out_depth [h][w] averaged depth; out_geo_mask / out_prob_mask / out_mask [h][w] bytes (0/1; any may be NULL);
 * out_points [3][h][w] world coordinates of the averaged depth or NULL; out_reproj_xyd [n_src][3][h][w] or NULL.
 * n_src <= 16. */
int effi_fusion_dynamic_filter_f32(const float* ref_depth, const float* src_depths, const float* ref_cam,
                                   const float* src_cams, int n_src, int h, int w, const float* ref_conf, int conf_h,
                                   int conf_w, float prob_threshold, int dh_view_num, float dist_base, float rel_diff_base,
                                   int relative, float* mats_scratch, float* out_depth, unsigned char* out_geo_mask,
                                   unsigned char* out_prob_mask, unsigned char* out_mask, float* out_points,
                                   float* out_reproj_xyd, effi_stream_t stream);

/* The same row's individual functions, for callers that use them one by one as the T&T driver does (test_tank.py:486-509):
 * vis_filter_dynamic (misc/fusion.py:157-181): ref_depth [n][1][h][w], reproj_xyd [n][n_src][3][h][w] ->
 * masks [n][n_src][n_src+1-thres_view][h][w] bytes (0/1); the reference's second return value is the last threshold's plane. */
int effi_fusion_vis_filter_f32(const float* ref_depth, const float* reproj_xyd, int n, int n_src, int h, int w, float dist_base,
                               float rel_diff_base, int thres_view, int relative, unsigned char* masks, effi_stream_t stream);
/* Point transforms of misc/fusion.py with one camera [2][4][4] per batch element (cams [n][2][4][4]); mode 0 = idx_img2cam (:23-28:
 * in [.][h][w][3] homogeneous pixels, batch stride in_batch_stride floats (0 = one grid for all), depth [n][h][w] -> out [n][h][w][4]),
 * 1 = idx_cam2world (:31-34), 2 = idx_world2cam (:37-40): [n][h][w][4] -> [n][h][w][4], 3 = idx_cam2img (:43-47): -> [n][h][w][3].
 * Inverses in fp64 on the device, rounded once.  mats_scratch: >= 52*n floats. */
int effi_fusion_points_f32(int mode, const float* in, long in_batch_stride, const float* depth, const float* cams, int n, int h,
                           int w, float* mats_scratch, float* out, effi_stream_t stream);

/* DTU branch, the two functions of test_dtu_dypcd.py:164-233 for ONE (reference, source) pair (PARITY UNPINNED like
 * effi_fusion_dtu_filter_f32, whose arithmetic they share): out5 [5][h][w] = depth_reprojected, x_reprojected, y_reprojected, x_src,
 * y_src (reproject_with_depth); masks != NULL -> check_geometric_consistency: masks [e-s][h][w] bytes and the first three planes
 * zeroed where the last mask is false.  ref_cam / src_cam [2][4][4]; mats_scratch >= 104 floats. */
int effi_fusion_dtu_reproject_f32(const float* ref_depth, const float* src_depth, const float* ref_cam, const float* src_cam, int h,
                                  int w, int s, int e, float dist_base, float diff_base, float* mats_scratch, float* out5,
                                  unsigned char* masks, effi_stream_t stream);

/* Split-precision form of effi_warpcorr_views_f32 / effi_warpcorr_views_tbl_f32 (round 4; the same reference lines,
 * models/Effi_MVS_plus.py:38-44 + models/module.py:303-344): correlation is linear in the warped feature, so the kernel evaluates
 * G[reference pixel][source pixel] = sum_c ref * src for a row segment's tap pixels on the matrix cores -- every fp32 product as three
 * bf16 partial products (hi*hi + lo*hi + hi*lo) with fp32 accumulation, the arithmetic of the path's "split" convolutions; hi_only != 0:
 * bf16 operands (precision "bf16") -- and interpolates the correlations with the bilinear weights (4 values per hypothesis instead of
 * 4 x C).  C = 32 with hypotheses shared by all pixels (depth_pstride = 0: the cascade's stage 1); any other shape runs the exact
 * kernels of effi_warpcorr_views_f32.  Exact-fp32 precision and training use the exact entries. */
int effi_warpcorr_views_x3_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                               const float* depth, long depth_dstride, long depth_pstride,
                               int C, int h, int w, int D, float* sim_views, float* entropy, int hi_only,
                               effi_stream_t stream);
int effi_warpcorr_views_x3_tbl_f32(const float* const* view_table_dev, int S, const float* rt, const float* depth,
                                   long depth_dstride, long depth_pstride, int C, int h, int w, int D, float* sim_views,
                                   float* entropy, int hi_only, effi_stream_t stream);
/* ---- scope row n2, first piece: backward of effi_warpcorr_views_f32's similarity output ------------------------------------
 * sim[v][d][p] = mean_c ref[p][c] * bilinear(src_v)[c] at the warped position (models/module.py:303-344,
 * models/Effi_MVS_plus.py:38-40); the sampling grid carries no gradient (module.py:313).  Inputs as the forward entry;
 * grad_sim [S][D][h*w]; grad_ref_nhwc [h][w][C] and grad_src_nhwc[v] [h][w][C] must be ZERO on entry (fp32 atomic adds: summation order,
 * hence the last bits, vary between runs).  C = 32 with shared hypotheses: the scatter into grad_src is privatised in an LDS window per
 * (tile, chunk of hypotheses) and flushed once (warpcorr_views_bwd_win_kernel). */
int effi_warpcorr_views_bwd_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt,
                                const float* depth, long depth_dstride, long depth_pstride, int C, int h, int w, int D,
                                const float* grad_sim, float* grad_ref_nhwc, float* const* grad_src_nhwc,
                                effi_stream_t stream);
/* Backward of effi_homo_warp_f32 w.r.t. the source features (the "differentiable homography warping" of homo_warping_new,
 * models/module.py:303-344): grad_out [C][D][h*w]; grad_src_nhwc [h][w][C] must be ZERO on entry (fp32 atomic scatter-add). */
int effi_homo_warp_bwd_f32(const float* rt, const float* depth, long depth_dstride, long depth_pstride, int C, int h, int w,
                           int D, const float* grad_out, float* grad_src_nhwc, effi_stream_t stream);

/* ---- pair launches for the two cross-scale blocks of a stage (CSP_R[s] / CSP_C[s], models/Effi_MVS_plus.py:520-531): the two
 * blocks are independent chains of the same five layers on volumes of the same shape.  Each entry below runs the SAME layer of
 * both blocks in one launch (a grid dimension picks the block), with the arithmetic of the single-block entry it is named after;
 * results are bitwise those of two single launches.  (On two streams the chains overlap as well, but each fork / join of
 * streams inside a captured graph costs ~5 / ~11 us of idle GPU.) */
/* effi_vol_lookup1d_f32 into two volumes of the same shape with the same queries and ranges. */
int effi_vol_lookup1d_pair_f32(const float* vol_a, const float* vol_b, long vol_dstride, long vol_pstride, int Dp,
                               const float* query, long q_dstride, long q_ystride, long q_xstride, int nq, const float* dmin,
                               const float* dmax, long range_pstride, int h, int w, float* out_a, float* out_b,
                               effi_stream_t stream);
/* effi_conv3d_k3_f32 twice: one source tensor of cin channels each, cout == 8, stride (1, sxy, sxy), sxy in {1, 2}, no skip. */
int effi_conv3d_k3_pair_f32(const float* in_a, const float* weight_a, const float* bias_a, float* out_a, const float* in_b,
                            const float* weight_b, const float* bias_b, float* out_b, int cin, int cout, int D, int h, int w,
                            int sxy, int relu, effi_stream_t stream);
/* effi_conv3d_k3s1_roll_bf16x3_f32 twice (same source channel split, cout <= 16). */
int effi_conv3d_k3s1_roll_bf16x3_pair_f32(const float* const* srcs_a, const void* wpack_a, const float* bias_a, float* out_a,
                                          const float* const* srcs_b, const void* wpack_b, const float* bias_b, float* out_b,
                                          const int* src_channels, int n_src, int cout, int D, int h, int w, int relu,
                                          effi_stream_t stream);
/* Cross-scale block, conv0 | conv_cost -> conv1 of TWO blocks over one fine volume in one launch (reference models/module.py:501-516,
 * caller models/Effi_MVS_plus.py:520-531): x [D][H][W]; per block prior [D][h][w] (h = (H-1)/2+1, w = (W-1)/2+1), w0 / wc [27][8] and
 * b0 / bc [8] (conv0 stride (1,2,2) and conv_cost, BatchNorm folded, ReLU), w1 / b1 = conv1 as effi_conv3d_k3s1_roll_bf16x3_f32 takes
 * it (row-pair operand, cin 16, cout 8, ReLU) -> out [8][D][h][w].  The 16 channels between the layers are generated inside the
 * rolling-window kernel and never reach memory; bitwise the result of effi_conv3d_k3_pair_f32 (sxy 2, sxy 1) +
 * effi_conv3d_k3s1_roll_bf16x3_pair_f32.  D <= 14. */
int effi_csp_gen_roll_bf16x3_pair_f32(const float* x, int D, int H, int W, const float* prior_a, const float* w0_a, const float* b0_a,
                                      const float* wc_a, const float* bc_a, const void* w1_a, const float* b1_a, float* out_a,
                                      const float* prior_b, const float* w0_b, const float* b0_b, const float* wc_b, const float* bc_b,
                                      const void* w1_b, const float* b1_b, float* out_b, effi_stream_t stream);
/* Diagnostics (tools/stress_c8_corun.py, tests): every CU's LDS is overwritten with `pattern` by a throw-away kernel, so that a later
 * kernel reading LDS it has not written sees that pattern instead of a previous workgroup's data.  sink4: 4 writable bytes. */
int effi_debug_poison_lds(unsigned pattern, void* sink4, effi_stream_t stream);
/* Diagnostics (tools/probe_pk_war.py): repeats `acc += (x, y) * w` as one v_pk_fma_f32 whose low source register is overwritten by the
 * very next instruction (and restored) -- variant 0: back to back, 1: an s_nop between, 3: two v_fma_f32 instead -- and returns the
 * accumulators of every lane in out[2 * 256 * blocks]; expected (16 reps x, 16 reps y) with x = 1 + (tid & 15) / 16, y = 2. */
int effi_debug_pk_war_probe(float* out, int blocks, int reps, int variant, effi_stream_t stream);
/* effi_deconv3d_k3_f32 twice: stride (1,2,2), cout == 1, no skip. */
int effi_deconv3d_k3_pair_f32(const float* in_a, const float* weight_a, const float* bias_a, float* out_a, const float* in_b,
                              const float* weight_b, const float* bias_b, float* out_b, int cin, int cout, int D, int h, int w,
                              int sz, int relu, effi_stream_t stream);

/* ---- scope row n4 (input pipeline): decoded 8-bit image -> network input planes -------------------------------------
 * datasets/general_eval.py:83-88 (read_img: / 255.), :94-117 and :160-166 (cv2.resize, INTER_LINEAR), :189 (HWC -> CHW);
 * datasets/tank.py:101-107.  img_hwc: [src_h][src_w][channels] uint8 (channels 1 or 3); out_chw: [channels][dst_h][dst_w]
 * fp32 = resize_linear(img / 255) with OpenCV's pixel-centre convention (horizontal pass, then vertical, fp32). */
int effi_image_prepare_u8_f32(const unsigned char* img_hwc, int src_h, int src_w, int channels, int dst_h, int dst_w,
                              float* out_chw, effi_stream_t stream);
/* The same resize for an image that is already fp32 planar [channels][src_h][src_w] (second resize to the scene's standard
 * size, datasets/general_eval.py:160-166). */
int effi_resize_linear_f32(const float* in_chw, int channels, int src_h, int src_w, int dst_h, int dst_w, float* out_chw,
                           effi_stream_t stream);

/* ==== scope row n2: what makes the path trainable (train.py:229-263: model.train() -> forward -> loss.backward()) =============
 * The reference gets these from torch autograd through nn.Conv2d / nn.Conv3d / nn.ConvTranspose3d / nn.BatchNorm / F.grid_sample;
 * here every one is an explicit kernel.  Input gradients of the convolutions reuse the forward entries above with re-arranged
 * weights (effi_mvs_plus_amd/autograd.py); the entries below are the rest. */

/* Weight gradient of every convolution on the path:  dw[a][cb_off + b][tap] += sum_o A[a][o] * B[b][o*stride + tap - pad],
 * A planar [ca][Da][ha][wa] (small grid), B planar [cb][Db][hb][wb] (large grid), tap = (kz, ky, kx) of a kd x ks x ks kernel,
 * pad = k/2, stride (sz, sxy, sxy) in {1, 2}.  dw planar [ca][cb_total][kd*ks*ks], ACCUMULATED (atomics; zero it first).
 *   nn.Conv2d / nn.Conv3d (models/module.py:124-166, models/update.py:14-15,36-38,73-81,109-112): A = grad_out, B = input
 *     -> torch's [cout][cin][k..]; a concatenated input is handled per part (cb_off = its first channel).
 *   nn.ConvTranspose3d (models/module.py:168-209): A = input, B = grad_out -> torch's [cin][cout][k..].
 * (kd, ks) in {(1,1), (1,3), (1,5), (1,7), (3,3)}; 2-D: Da = Db = 1. */
int effi_conv_wgrad_f32(const float* a, const float* b, int ca, int cb, int cb_total, int cb_off, int kd, int ks, int Da, int ha,
                        int wa, int Db, int hb, int wb, int sz, int sxy, float* dw, effi_stream_t stream);
/* Input gradient of the feature pyramid's 5x5 / stride-2 / padding-2 convolutions (models/module.py:376-388; scope row n1 in
 * training): grad_out planar [cout][ceil(hin/2)][ceil(win/2)], weight torch layout [cout][cin][5][5] -> grad_in planar [cin][hin][win]. */
int effi_conv2d_k5s2_dgrad_f32(const float* grad_out, const float* weight, int cin, int cout, int hin, int win, float* grad_in,
                               effi_stream_t stream);
/* out[c] = sum over batch and positions of g [B][C][n]  (bias gradients; out is overwritten).  The per-channel reductions of this section take a
 * caller-owned ``scratch`` ([C][nsplit][K] floats; K = 1, for effi_bn_bwd_f32 K = 4: two DOUBLES, 8-byte aligned -- its two sums are
 * accumulated in double because gx is a cancelling projection of them) and ``nsplit``: nsplit workgroups per channel write
 * partial sums and a second small launch adds them in ascending order; scratch == NULL or nsplit <= 1: one workgroup per channel.
 * No atomics either way: the results are bitwise repeatable. */
int effi_channel_sum_f32(const float* g, int B, int C, long n, float* out, float* scratch, int nsplit, effi_stream_t stream);

/* nn.BatchNorm2d / nn.BatchNorm3d on batch statistics (models/module.py:148-157,191-200,217-220), x planar [B][C][n]:
 *   effi_bn_moment_f32: out[c] = sum (x - shift[c])^power   (power 1 with shift = NULL: sum; power 2 with shift = mean);
 *   effi_bn_apply_f32:  y = (x - mean) * invstd * gamma + beta, then ReLU if relu != 0;
 *   effi_bn_bwd_f32:    s1[c] = sum g', s2[c] = sum g' * xhat (overwritten), then
 *                       gx = gamma * invstd * (g' - s1/N - xhat * s2/N), g' = gy masked by y > 0 when relu != 0.
 *                       (grad gamma = s2, grad beta = s1.) */
int effi_bn_moment_f32(const float* x, int B, int C, long n, const float* shift, int power, float* out, float* scratch, int nsplit,
                       effi_stream_t stream);
int effi_bn_apply_f32(const float* x, int B, int C, long n, const float* mean, const float* invstd, const float* gamma,
                      const float* beta, int relu, float* y, effi_stream_t stream);
/* The forward of nn.BatchNorm2d / 3d in training mode as one entry (models/module.py:148-157,191-200,217-220 under model.train()):
 * mean / invstd [C] (outputs, kept for the backward) from the two moment passes, y = BN(x) (+ ReLU), and -- when running_mean /
 * running_var are given -- nn.BatchNorm's running-statistic update with ``momentum`` (unbiased variance) and num_batches_tracked += 1
 * (n_tracked: the module's int64 counter, or NULL).  scratch: 2 x [C][nsplit] floats, nsplit >= 1 workgroups per channel (partials
 * are added in ascending order: bitwise repeatable). */
int effi_bn_train_fwd_f32(const float* x, int B, int C, long n, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, long long* n_tracked, int relu, float* y, float* mean,
                          float* invstd, float* scratch, int nsplit, effi_stream_t stream);
/* nn.Conv2d weights [cout][cin][ks][ks] (+ bias or NULL) -> the operand order effi_conv2d_f32 reads (wpack [ceil(ci/4)][ks*ks]
 * [ceil(co/16)][64] floats, bias_pack [16 ceil(co/16)]); dgrad != 0: the weights of the input-gradient convolution (in / out swapped,
 * taps flipped; co = cin, ci = cout; bias_pack zero unless bias given).  Training re-packs every layer every step
 * (train.py:229-263: the optimizer changed the weights): one launch here. */
int effi_pack_conv2d_mfma_f32(const float* weight, const float* bias, int cout, int cin, int ks, int dgrad, float* wpack,
                              float* bias_pack, effi_stream_t stream);
/* The input gradient of a 5x5 / stride-2 / padding-2 convolution (models/module.py:376-388 under loss.backward()) as a 3x3 convolution
 * over the output gradient + pixel shuffle: packs, for input channels [ci_off, ci_off + ci_n) of weight [cout][cin][5][5], the
 * 3x3 weights with 4 ci_n output channels (4 ci + 2 py + px = input channel ci, pixel parity (py, px)) and cout input channels in
 * effi_conv2d_f32's operand order (wpack [ceil(cout/4)][9][ceil(4 ci_n/16)][64], bias_pack zeros). */
int effi_pack_conv2d_k5s2_dgrad_f32(const float* weight, int cout, int cin, int ci_off, int ci_n, float* wpack, float* bias_pack,
                                    effi_stream_t stream);
int effi_bn_bwd_f32(const float* gy, const float* y, const float* x, int B, int C, long n, const float* mean, const float* invstd,
                    const float* gamma, int relu, float* s1, float* s2, float* gx, float* scratch, int nsplit,
                    effi_stream_t stream);

/* Element-wise pieces of the GRU block and heads, forward and backward (models/update.py:20-27,40-49,86-99;
 * models/Effi_MVS_plus.py:138-148).  n elements; unused pointers NULL. */
#define EFFI_PW_ACT_BWD_RELU     0   /* o0 = a * (b > 0)            a = grad, b = the activation's OUTPUT */
#define EFFI_PW_ACT_BWD_SIGMOID  1   /* o0 = a * b (1 - b) */
#define EFFI_PW_ACT_BWD_TANH     2   /* o0 = a * (1 - b^2) */
#define EFFI_PW_TANH             3   /* o0 = tanh(a) */
#define EFFI_PW_RELU             4   /* o0 = max(a, 0) */
#define EFFI_PW_SIGMOID          5   /* o0 = sigmoid(a) */
#define EFFI_PW_MUL              6   /* o0 = a * b                  (r * h, update.py:46) */
#define EFFI_PW_MUL_BWD          7   /* o0 = a * c, o1 = a * b      a = grad, b = x, c = y of x * y */
#define EFFI_PW_GRU              8   /* o0 = (1 - a) b + a c        a = z, b = h, c = q (update.py:48) */
#define EFFI_PW_GRU_BWD          9   /* o0 = a (d - c) [z], o1 = a (1 - b) [h], o2 = a b [q];  a = grad, b = z, c = h, d = q */
#define EFFI_PW_INV_TO_DEPTH     10  /* o0 = 1 / max(lo + (hi - lo) a, 1e-4); s0 = lo, s1 = hi (scale_inv_depth) */
#define EFFI_PW_INV_TO_DEPTH_BWD 11  /* o0 = grad of the above w.r.t. b = inv; a = grad */
#define EFFI_PW_SCALE_CH         12  /* o0[i] = a[i] * b[(i / inner) % C]  (Dropout2d: per-(sample, channel) factors) */
#define EFFI_PW_COUNT            13
int effi_pointwise_f32(int op, const float* a, const float* b, const float* c, const float* d, float s0, float s1, long n, long inner,
                       int C, float* o0, float* o1, float* o2, effi_stream_t stream);

/* Backward of effi_vol_lookup1d_f32 w.r.t. the looked-up volume (pro_bilinear_sampler, models/Effi_MVS_plus.py:102-134; the query
 * coordinates come from detached depths): gvol (layout as the forward's vol) is ACCUMULATED without atomics (a thread owns a pixel). */
int effi_vol_lookup1d_bwd_f32(float* gvol, long vol_dstride, long vol_pstride, int Dp, const float* query, long q_dstride,
                              long q_ystride, long q_xstride, int nq, const float* dmin, const float* dmax, long range_pstride, int h,
                              int w, const float* gout, effi_stream_t stream);
/* Backward of effi_getcost_f32 w.r.t. the two cached volumes (GetCost.forward, models/Effi_MVS_plus.py:257-303):
 * gcost [2*nq][h*w] -> gcur, greg accumulated. */
int effi_getcost_bwd_f32(const float* inv_depth, const float* disp_range, int n_range, int input_is_depth, const float* interval,
                         float* gcur, long cur_dstride, long cur_pstride, int Dcur, float* greg, long reg_dstride, long reg_pstride,
                         int Dreg, const float* dmin, const float* dmax, long range_pstride, int nq, int h, int w, const float* gcost,
                         effi_stream_t stream);
/* Backward of the soft-argmin (models/Effi_MVS_plus.py:79-81): glogits[d] = gdepth * p_d * (hyp_d - depth). */
int effi_softargmin_bwd_f32(const float* logits, const float* hyp, long depth_dstride, long depth_pstride, int D, int hw,
                            const float* gdepth, float* glogits, effi_stream_t stream);
/* Backward of effi_view_aggregate_f32 (models/Effi_MVS_plus.py:48-53,67): gsim [S][D][hw], gw [S][hw] are written. */
int effi_view_aggregate_bwd_f32(const float* sim_views, const float* weights, int S, int D, int hw, const float* gout, float* gsim,
                                float* gw, effi_stream_t stream);
/* Backward of the convex x2 upsampling (models/Effi_MVS_plus.py:167-178): gup [2h][2w] -> gmask [36][h][w] (written),
 * ginv [h][w] (written).  scratch9: [9][h][w] floats -- every pixel's contribution to its nine neighbours is written there and a
 * second launch gathers them in a fixed order (no atomics: bitwise repeatable). */
int effi_convex_upsample2x_bwd_f32(const float* inv_depth, const float* mask, int h, int w, const float* gup, float* gmask, float* ginv,
                                   float* scratch9, effi_stream_t stream);
/* Backward of effi_warpcorr_dyn_f32 (GetCost_initvolume.forward, models/Effi_MVS_plus.py:184-251): sim = the forward's output,
 * grad_sim [D][h*w]; grad_ref_nhwc is written; grad_src_nhwc[v] (fp32) and grad_view_w [S][h>>k][w>>k] (DOUBLE: a cancelling sum
 * accumulated with 64-bit atomics so that the order of the adds stays below fp32 resolution) must be ZERO on entry. */
int effi_warpcorr_dyn_bwd_f32(const float* ref_nhwc, const float* const* src_nhwc, int S, const float* rt, const float* cur_depth,
                              const float* interval, const float* view_w, int vw_shift, int C, int h, int w, int D, const float* sim,
                              const float* grad_sim, float* grad_ref_nhwc, float* const* grad_src_nhwc, double* grad_view_w,
                              effi_stream_t stream);

/* ---- plain-bf16-operand variants of the split-precision convolution entries -------------------------------------------------
 * BASELINE.json's "bf16 (MFMA 3D-conv path)" configuration: every *_bf16x3_* entry above exists a second time with the suffix
 * _bf16 -- same arguments, same packed weights, same kernels compiled with the two lo terms of each product removed (hi*hi only:
 * bf16 operands, fp32 accumulation).  NOT fp32-grade: the stated tolerance of this variant is a normalised mean depth error
 * <= 1e-2; the reference itself is fp32 only (models/module.py:318). */
int effi_conv2d_k3_bf16x3_pair_f32_bf16(const float* const* srcs_a, const int* src_channels_a, int n_src_a, const void* wpack_a,
                                   const float* bias_a, float* out_a, const float* const* srcs_b,
                                   const int* src_channels_b, int n_src_b, const void* wpack_b, const float* bias_b,
                                   float* out_b, int cout, int h, int w, int act, effi_stream_t stream);
int effi_conv2d_k3_bf16x3_f32_bf16(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                              const float* bias, int cout, int h, int w, int epilogue, int act,
                              const float* aux0, const float* aux1, const float* disp_range, int n_range,
                              float* out0, float* out1, effi_stream_t stream);
int effi_conv2d_k3_k1_bf16x3_f32_bf16(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                 const float* bias, int cout1, int relu1, const float* extra, int c_extra,
                                 const void* w2pack_bf16, const float* bias2, int cout2, int relu, int h, int w, float* out,
                                 effi_stream_t stream);
int effi_conv3d_k3s2_bf16x3_f32_bf16(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int D, int h, int w,
                                int relu, float* out, effi_stream_t stream);
int effi_conv2d_k5s2_bf16x3_f32_bf16(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout, int hin, int win,
                                int act, float* out, effi_stream_t stream);
int effi_conv2d_k3_twice_bf16x3_f32_bf16(const float* in, int cin, const void* w1_bf16, const float* bias1, const void* w2_bf16,
                                    const float* bias2, int cout, int h, int w, float* out, effi_stream_t stream);
int effi_encoder_tail_bf16x3_f32_bf16(const float* cor1, const float* dfm1, int hd, const void* wc2_bf16, const float* bias_c2,
                                 const void* wd2_bf16, const float* bias_d2, const void* wd_bf16, const float* bias_d, int cmix,
                                 const float* extra, int c_extra, const void* w2pack_bf16, const float* bias2, int cout2, int h,
                                 int w, float* out, effi_stream_t stream);
int effi_conv2d_k3_k1_up2x_bf16x3_f32_bf16(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                      const float* bias, int cout1, const void* w2pack_bf16, const float* bias2,
                                      const float* inv_depth, const float* disp_range, int n_range, int h, int w,
                                      float* out_depth, float* out_depth_inv, effi_stream_t stream);
int effi_conv3d_k3s1_bf16x3_f32_bf16(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                const float* bias, int cout, int D, int h, int w, int relu, float* out,
                                effi_stream_t stream);
int effi_conv3d_k3s1_roll_bf16x3_f32_bf16(const float* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                     const float* bias, int cout, int D, int h, int w, int relu, float* out,
                                     effi_stream_t stream);
int effi_conv3d_k3s1_roll_bf16x3_pair_f32_bf16(const float* const* srcs_a, const void* wpack_a, const float* bias_a, float* out_a,
                                          const float* const* srcs_b, const void* wpack_b, const float* bias_b, float* out_b,
                                          const int* src_channels, int n_src, int cout, int D, int h, int w, int relu,
                                          effi_stream_t stream);
int effi_csp_gen_roll_bf16x3_pair_f32_bf16(const float* x, int D, int H, int W, const float* prior_a, const float* w0_a, const float* b0_a,
                                      const float* wc_a, const float* bc_a, const void* w1_a, const float* b1_a, float* out_a,
                                      const float* prior_b, const float* w0_b, const float* b0_b, const float* wc_b, const float* bc_b,
                                      const void* w1_b, const float* b1_b, float* out_b, effi_stream_t stream);
int effi_deconv3d_k3s2_bf16x3_f32_bf16(const float* in, int cin, const void* wpack_bf16, const float* bias, int cout,
                                  int D, int h, int w, int relu, const float* skip, float* out, effi_stream_t stream);

/* ---- scope row n3, DTU branch: reproject_with_depth + check_geometric_consistency + the array part of filter_depth,
 * test_dtu_dypcd.py:164-333 (the reference runs it with numpy / cv2 on the host, one scan per pool worker).  PARITY UNPINNED: cv2
 * (cv2.remap, cv2.resize) is absent from the build image and the reference holds no fixtures; checked against
 * oracle/effi_dtu_filter_oracle.py (numpy lines restated with their dtypes, OpenCV's published INTER_LINEAR remap restated).
 * ref_depth [h][w], src_depths [n_src][h][w] (the SAME size: the reference filters full-resolution maps), cameras as
 * effi_fusion_dynamic_filter_f32 ([2][4][4]: extrinsic, intrinsic in the top-left 3x3), confidence [h][w] ALREADY resized to the
 * depth size (cv2.resize, :259 = effi_resize_linear_f32) or NULL (= 1 everywhere); conf_threshold = args.conf (:261),
 * conf_keep = 0.75 (:300), s / e / dist_base / diff_base the module constants of :33-37 (1, 11, 0.5, 0.25).
 * Outputs: out_depth = depth_est_averaged, masks (or NULL) as bytes, out_points (or NULL) [3][h][w] = world point of every pixel
 * (the reference keeps those where the final mask is set).  mats_scratch: 52 * (n_src + 1) floats. */
int effi_fusion_dtu_filter_f32(const float* ref_depth, const float* src_depths, const float* ref_cam, const float* src_cams, int n_src,
                               int h, int w, const float* confidence, float conf_threshold, float conf_keep, int s, int e,
                               float dist_base, float diff_base, float* mats_scratch, float* out_depth, unsigned char* out_photo_mask,
                               unsigned char* out_geo_mask, unsigned char* out_final_mask, float* out_points, effi_stream_t stream);

/* ---- tuning / A-B switches.  A table of named integers, initialised ONCE per process from the environment (variable EFFI_<NAME in
 * upper case>) and changed afterwards only through effi_set_option; no entry point reads the environment.  Names: warp_lds_kb
 * (stage-1 warp kernel: LDS window in KB; 0 = the window kernel on global loads, -1 = the direct-gather kernel), dyn_form (1 =
 * channel-split lanes in the stage-2/3 warp kernel), dyn_setup_exact (1 = IEEE divisions there), pixnet_mfma (0 = vector-ALU
 * view-weight net), force_mr / mr4_min / mr4_nt2_max / mr2_min / wide_tiles (tile rule of the split-precision 3x3 convolutions),
 * roll_mr / roll_zt / roll_rp / deconv_mr (tile rules of the 3-D convolutions), dyn_xchg (diagnostic builds only).
 * effi_set_option returns EFFI_ERR_BADARG for an unknown name; effi_get_option returns effi_option_unset() for an unknown or unset one;
 * effi_set_option(name, effi_option_unset()) restores "unset" (the built-in rule). */
int effi_set_option(const char* name, long value);
long effi_get_option(const char* name);
long effi_option_unset(void);

/* ================================================================================================================
 * Split-resident ("SR") activation maps of the GRU update block, models/update.py:33-49,69-99,109-141.
 * In split precision every 3x3 convolution reads an fp32 value x as hi = bf16(x), lo = bf16(x - hi).  The layers of the update
 * block feed each other, so their producers store the two halves directly, in the order the consumer's MFMA fragments want:
 *     map[octet o of 8 channels][part: 0 = hi, 1 = lo][row y + 1][column x + 1][channel e]        (bf16)
 * with [hp][wp] pixels per plane, a one-pixel zero border and zeros up to the tile overhang (effi_sr_geometry); the border IS the
 * convolution's zero padding.  A chain through SR maps is bitwise equal to the same chain through fp32 maps (same conversion
 * instructions, applied once by the producer instead of by every consumer).  Maps are caller-owned; the border is zeroed with
 * effi_sr_clear_border once per allocation (producers never write outside rows 1..h, columns 1..w).
 * ================================================================================================================ */
/* plane size the kernels expect for an h x w map: hp = roundup(h, 16) + 2, wp = roundup(w, w >= 512 ? 64 : 16) + 2 (larger is fine) */
int effi_sr_geometry(int h, int w, int* hp, int* wp);
/* zero the border of n_groups (<= 4) blocks of planes[g] consecutive planes of geometry (h, w, hp, wp)[g]: one launch */
int effi_sr_clear_border(void* const* maps, const int* planes, const int* h, const int* w, const int* hp, const int* wp, int n_groups,
                         effi_stream_t stream);
/* fp32 planar [channels][h][w] (channels % 8 == 0) -> SR map (interior only) */
int effi_sr_from_planar_f32(const float* in, int channels, int h, int w, void* sr, int hp, int wp, effi_stream_t stream);
/* effi_split_tanh_relu_stages_f32 (models/Effi_MVS_plus.py:442-452) that ALSO writes each hidden state as an SR map (hd % 8 == 0,
 * cd % 4 == 0): what the first ConvGRU convolution of a stage reads.  hidden_q4[k] != 0 (hidden_q4 == NULL: all 0): the fp32
 * hidden state of stage k is written in the Q4 layout of EFFI_EPI_Q4. */
int effi_split_tanh_relu_stages_sr_f32(const float* const* ctx, const int* hd, const int* cd, const int* h, const int* w,
                                       float* const* hidden, void* const* hidden_sr, const int* hp, const int* wp, float* const* inp,
                                       const int* hidden_q4, int n_stages, effi_stream_t stream);
/* effi_encoder_inputs_bf16x3_f32 (models/update.py:86,90) writing both maps split-resident (sr_c1 = relu(convc1(cost)),
 * sr_d1 = relu(convd1(inv_depth)); [cout/8][2][hp][wp][8]). */
int effi_encoder_inputs_bf16x3_sr(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                  const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds, long rps,
                                  int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h, int w,
                                  const float* weight_c1, const float* bias_c1, const float* weight_d1, const float* bias_d1,
                                  int cout, void* sr_c1, void* sr_d1, int hp, int wp, effi_stream_t stream);
/* effi_encoder_inputs_bf16x3_sr + effi_conv2d_k3_bf16x3_pair_sr in ONE launch (models/update.py:86-91): out_sr_c2 =
 * act(convc2(relu(convc1(GetCost(inv_depth))))), out_sr_d2 = act(convd2(relu(convd1(inv_depth)))).  The 1x1 / 7x7 results are generated
 * tile by tile inside the 3x3 kernel and never reach memory; bitwise equal to the two launches.  nq == 3, hd in {16, 32, 48},
 * cout in {16, 32, 48}; weight_* as for effi_encoder_inputs_bf16x3_sr, wpack_* / bias_* as for the pair. */
int effi_encoder_pair_gen_bf16x3_sr(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                    const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds, long rps,
                                    int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h, int w,
                                    const float* weight_c1, const float* bias_c1, const float* weight_d1, const float* bias_d1, int hd,
                                    const void* wpack_c2, const float* bias_c2, void* out_sr_c2, const void* wpack_d2,
                                    const float* bias_d2, void* out_sr_d2, int cout, int hp, int wp, int act, effi_stream_t stream);
/* effi_conv2d_k3_bf16x3_f32 on SR maps (models/update.py:36-38,43-48,75,77): srcs = SR maps of src_channels[i] channels each
 * (multiples of 16), cout % 16 == 0.  EFFI_EPI_PLAIN: out_sr = act(conv) (out0, if given, the same values as fp32 [cout][h][w]);
 * EFFI_EPI_GRU_ZR: out0 = z (fp32 [cout/2][h][w]), out_sr = r * h, aux0 = h (fp32); EFFI_EPI_GRU_Q: out0 AND out_sr =
 * (1 - z) h + z tanh(conv), aux0 = h, aux1 = z (fp32).  With EFFI_EPI_Q4 or-ed into a GRU epilogue the fp32 maps are Q4 (see the flag). */
int effi_conv2d_k3_bf16x3_sr(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16, const float* bias,
                             int cout, int h, int w, int hp, int wp, int epilogue, int act, const float* aux0, const float* aux1,
                             float* out0, void* out_sr, effi_stream_t stream);
/* effi_conv2d_k3_bf16x3_pair_f32 on SR maps: convc2 / convd2 of the encoder (models/update.py:87,91), SR in, SR out. */
int effi_conv2d_k3_bf16x3_pair_sr(const void* const* srcs_a, const int* src_channels_a, int n_src_a, const void* wpack_a,
                                  const float* bias_a, void* out_sr_a, const void* const* srcs_b, const int* src_channels_b,
                                  int n_src_b, const void* wpack_b, const float* bias_b, void* out_sr_b, int cout, int h, int w,
                                  int hp, int wp, int act, effi_stream_t stream);
/* effi_conv2d_k3_k1_bf16x3_f32 on SR inputs (convd -> convc, models/update.py:78-80,92-96; depth head conv1 + tap projections,
 * :14-15,21): exactly one of out (fp32 [cout2][h][w]) and out_sr (SR, cout2 % 16 == 0) is given. */
int effi_conv2d_k3_k1_bf16x3_sr(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16, const float* bias,
                                int cout1, int relu1, const float* extra, int c_extra, const void* w2pack_bf16, const float* bias2,
                                int cout2, int relu, int h, int w, int hp, int wp, float* out, void* out_sr, effi_stream_t stream);
/* effi_conv2d_k3_k1_up2x_bf16x3_f32 on SR inputs (mask head + convex upsampling, models/update.py:109-112,136-138,
 * models/Effi_MVS_plus.py:167-178). */
int effi_conv2d_k3_k1_up2x_bf16x3_sr(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                     const float* bias, int cout1, const void* w2pack_bf16, const float* bias2, const float* inv_depth,
                                     const float* disp_range, int n_range, int h, int w, int hp, int wp, float* out_depth,
                                     float* out_depth_inv, effi_stream_t stream);
/* ConvGRU (models/update.py:33-49) as ONE launch on split-resident maps: z | r = sigmoid(convzr([h, x])), q = tanh(convq([r * h, x])),
 * h' = (1 - z) h + z q; r * h and z never reach memory (csrc/gru_fused.hpp).  H_in / X: SR maps of hd channels; h_in fp32 [hd][h][w];
 * wzr_pack / bias_zr: convz | convr as one layer (cout 2 hd), wq_pack / bias_q: convq, both in effi_conv2d_k3_bf16x3_f32's operand
 * order; outputs h_out (fp32) and H_out (SR) must NOT alias the inputs (a workgroup reads its neighbours' pixels).  hd in {16, 32}.
 * Bitwise equal to effi_conv2d_k3_bf16x3_sr(GRU_ZR) + (GRU_Q). */
int effi_gru_zr_q_fused_bf16x3_sr(const void* H_in, const void* X, const float* h_in, const void* wzr_pack, const float* bias_zr,
                                  const void* wq_pack, const float* bias_q, int hd, int h, int w, int hp, int wp, float* h_out,
                                  void* H_out, effi_stream_t stream);
/* the convolution entries above with plain bf16 operands (hi only; precision "bf16"): the lo planes are neither read nor written */
int effi_conv2d_k3_bf16x3_sr_bf16(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16, const float* bias,
                                  int cout, int h, int w, int hp, int wp, int epilogue, int act, const float* aux0, const float* aux1,
                                  float* out0, void* out_sr, effi_stream_t stream);
int effi_conv2d_k3_bf16x3_pair_sr_bf16(const void* const* srcs_a, const int* src_channels_a, int n_src_a, const void* wpack_a,
                                       const float* bias_a, void* out_sr_a, const void* const* srcs_b, const int* src_channels_b,
                                       int n_src_b, const void* wpack_b, const float* bias_b, void* out_sr_b, int cout, int h, int w,
                                       int hp, int wp, int act, effi_stream_t stream);
int effi_conv2d_k3_k1_bf16x3_sr_bf16(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                     const float* bias, int cout1, int relu1, const float* extra, int c_extra, const void* w2pack_bf16,
                                     const float* bias2, int cout2, int relu, int h, int w, int hp, int wp, float* out, void* out_sr,
                                     effi_stream_t stream);
int effi_conv2d_k3_k1_up2x_bf16x3_sr_bf16(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                          const float* bias, int cout1, const void* w2pack_bf16, const float* bias2,
                                          const float* inv_depth, const float* disp_range, int n_range, int h, int w, int hp, int wp,
                                          float* out_depth, float* out_depth_inv, effi_stream_t stream);
int effi_gru_zr_q_fused_bf16x3_sr_bf16(const void* H_in, const void* X, const float* h_in, const void* wzr_pack, const float* bias_zr,
                                       const void* wq_pack, const float* bias_q, int hd, int h, int w, int hp, int wp, float* h_out,
                                       void* H_out, effi_stream_t stream);
int effi_encoder_pair_gen_bf16x3_sr_bf16(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                         const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds, long rps,
                                         int Dreg, const float* dmin, const float* dmax, long range_ps, int nq, int h, int w,
                                         const float* weight_c1, const float* bias_c1, const float* weight_d1, const float* bias_d1,
                                         int hd, const void* wpack_c2, const float* bias_c2, void* out_sr_c2, const void* wpack_d2,
                                         const float* bias_d2, void* out_sr_d2, int cout, int hp, int wp, int act, effi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EFFI_MVS_HIP_H */
