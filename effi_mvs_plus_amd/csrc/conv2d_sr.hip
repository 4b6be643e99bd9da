// The 3x3 convolutions of the GRU update block (models/update.py:33-49,69-99,109-141) on SPLIT-RESIDENT maps: the split-precision
// tile of conv2d_x3.hpp reading its inputs as ready-made (hi, lo) bf16 octets and writing its result the same way (see
// effi_sr_store4 there for the layout and why the values are bitwise those of the fp32-map chain).  Entry points mirror the
// planar ones of conv2d.hip (effi_conv2d_k3_bf16x3_f32, ..._pair_f32, ..._k3_k1_..., ..._k3_k1_up2x_...) with map geometry added.
//
// Like conv2d.hip this file is compiled twice: as it stands, and with -DEFFI_BF16_ONLY (entry names + _bf16, hi*hi products only;
// the lo planes of the maps are then neither read nor written).
#include "conv2d_x3.hpp"

namespace {

// host-side validation shared by the entries: sources are SR maps with a multiple of 16 channels each, geometry covers every tile
int fill_sr(Conv2dArgs& a, const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16, const float* bias,
            int cout, int h, int w, int hp, int wp) {
    if (!srcs || !src_channels || n_src < 1 || n_src > EFFI_MAX_SRC || !wpack_bf16 || !bias) return EFFI_ERR_BADARG;
    if (cout < 1 || h < 1 || w < 1) return EFFI_ERR_BADARG;
    // border + overhang of the widest tile the launch rule can pick for this map (effi_sr_geometry): 4 x 64 tiles only from 512 columns
    // on (or when option wide_tiles forces them), 16-column tiles otherwise
    const bool wide_possible = w >= 512 || effi_option(EFFI_OPT_WIDE_TILES) == 1;
    if (hp < ((h + 15) & ~15) + 2 || wp < (wide_possible ? ((w + 63) & ~63) : ((w + 15) & ~15)) + 2) return EFFI_ERR_BADARG;
    if ((long)hp * wp * 4 >= (1L << 31)) return EFFI_ERR_UNSUPPORTED;                          // 32-bit unit offsets inside a chunk
    a.cin = 0;
    for (int i = 0; i < EFFI_MAX_SRC; ++i) {
        a.src[i] = reinterpret_cast<const float*>((i < n_src) ? srcs[i] : srcs[0]);
        a.ch[i] = (i < n_src) ? src_channels[i] : 0;
        if (i < n_src && (!srcs[i] || src_channels[i] < 1)) return EFFI_ERR_BADARG;
        if (i < n_src && (src_channels[i] & 15)) return EFFI_ERR_UNSUPPORTED;                  // a 16-channel chunk lies in one source
        if (i < n_src && (reinterpret_cast<uintptr_t>(srcs[i]) & 15)) return EFFI_ERR_BADARG;
        a.cin += a.ch[i];
    }
    if ((long)cout * hp * wp >= (1L << 31)) return EFFI_ERR_UNSUPPORTED;     // the batched epilogue's 32-bit unit / Q4 offsets
    a.kgroups = (a.cin + 3) / 4;
    a.zeros = nullptr;                       // not read: padding is the maps' zero border
    a.wpack = reinterpret_cast<const float*>(wpack_bf16);
    a.bias = bias;
    a.cout = cout;
    a.h = a.hin = h;
    a.w = a.win = w;
    a.act = EFFI_ACT_NONE;
    a.hd = cout / 2;
    a.aux0 = a.aux1 = a.disp_range = nullptr;
    a.n_range = 0;
    a.out0 = a.out1 = nullptr;
    a.cstride = a.ostride = (long)h * w;
    a.zcount = a.zin = 0;
    a.xptr0 = nullptr;
    a.sr_hp = hp;
    a.sr_wp = wp;
    a.out_sr = nullptr;
    a.aux_q4 = 0;
    return EFFI_OK;
}

}  // namespace

extern "C" int EFFI_FN(effi_conv2d_k3_bf16x3_sr)(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                                 const float* bias, int cout, int h, int w, int hp, int wp, int epilogue, int act,
                                                 const float* aux0, const float* aux1, float* out0, void* out_sr,
                                                 effi_stream_t stream) {
    Conv2dArgs a;
    const int rc = fill_sr(a, srcs, src_channels, n_src, wpack_bf16, bias, cout, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    if (!out_sr || (reinterpret_cast<uintptr_t>(out_sr) & 15) || (cout & 15)) return EFFI_ERR_BADARG;
    a.act = act;
    a.aux0 = aux0;
    a.aux1 = aux1;
    a.out0 = out0;
    a.out_sr = reinterpret_cast<unsigned short*>(out_sr);
    const int nt = cout / 16;
    hipStream_t st = effi_s(stream);
    if (epilogue & EFFI_EPI_Q4) {             // fp32 maps of a GRU epilogue in the Q4 layout (16-byte accesses)
        epilogue &= ~EFFI_EPI_Q4;
        if (epilogue != EFFI_EPI_GRU_ZR && epilogue != EFFI_EPI_GRU_Q) return EFFI_ERR_BADARG;
        if ((reinterpret_cast<uintptr_t>(aux0) | reinterpret_cast<uintptr_t>(aux1) | reinterpret_cast<uintptr_t>(out0)) & 15) return EFFI_ERR_BADARG;
        a.aux_q4 = 1;
    }
    switch (epilogue) {
        case EFFI_EPI_PLAIN:                  // out_sr = act(conv); out0 (or NULL) = the same values as an fp32 map
            if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH) return EFFI_ERR_BADARG;
            return dispatch_bf16x3<EFFI_EPI_PLAIN, true>(a, nt, st);
        case EFFI_EPI_GRU_ZR:                 // out0 = z (fp32), out_sr = r * h; aux0 = h (fp32)
            if (!aux0 || !out0 || (cout % 32) != 0) return EFFI_ERR_BADARG;
            if (nt == 2) return launch_bf16x3<2, EFFI_EPI_GRU_ZR, false, true>(a, st);
            if (nt == 4) return launch_bf16x3<4, EFFI_EPI_GRU_ZR, false, true>(a, st);
            if (nt == 6) return launch_bf16x3<6, EFFI_EPI_GRU_ZR, false, true>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        case EFFI_EPI_GRU_Q:                  // out0 (fp32) and out_sr = (1 - z) h + z tanh(conv); aux0 = h, aux1 = z (fp32)
            if (!aux0 || !aux1 || !out0) return EFFI_ERR_BADARG;
            if (nt == 1) return launch_bf16x3<1, EFFI_EPI_GRU_Q, false, true>(a, st);
            if (nt == 2) return launch_bf16x3<2, EFFI_EPI_GRU_Q, false, true>(a, st);
            if (nt == 3) return launch_bf16x3<3, EFFI_EPI_GRU_Q, false, true>(a, st);
            return EFFI_ERR_UNSUPPORTED;
        default:
            return EFFI_ERR_UNSUPPORTED;
    }
}

extern "C" int EFFI_FN(effi_conv2d_k3_bf16x3_pair_sr)(const void* const* srcs_a, const int* src_channels_a, int n_src_a,
                                                      const void* wpack_a, const float* bias_a, void* out_sr_a,
                                                      const void* const* srcs_b, const int* src_channels_b, int n_src_b,
                                                      const void* wpack_b, const float* bias_b, void* out_sr_b, int cout, int h, int w,
                                                      int hp, int wp, int act, effi_stream_t stream) {
    if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH || (cout & 15) || !out_sr_a || !out_sr_b) return EFFI_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(out_sr_a) | reinterpret_cast<uintptr_t>(out_sr_b)) & 15) return EFFI_ERR_BADARG;
    Conv2dArgs a0, a1;
    int rc = fill_sr(a0, srcs_a, src_channels_a, n_src_a, wpack_a, bias_a, cout, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    rc = fill_sr(a1, srcs_b, src_channels_b, n_src_b, wpack_b, bias_b, cout, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    a0.act = a1.act = act;
    a0.out_sr = reinterpret_cast<unsigned short*>(out_sr_a);
    a1.out_sr = reinterpret_cast<unsigned short*>(out_sr_b);
    hipStream_t st = effi_s(stream);
    switch (cout / 16) {
        case 1: return launch_bf16x3_pair<1, true>(a0, a1, st);
        case 2: return launch_bf16x3_pair<2, true>(a0, a1, st);
        case 3: return launch_bf16x3_pair<3, true>(a0, a1, st);
        case 4: return launch_bf16x3_pair<4, true>(a0, a1, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

// encoder_inputs + the pair above in ONE launch (models/update.py:86-91): relu(convc2(relu(convc1(GetCost(inv_depth))))) -> out_sr_c2,
// relu(convd2(relu(convd1(inv_depth)))) -> out_sr_d2.  The 1x1 / 7x7 results are generated per tile inside the 3x3 kernel (EncGenArgs,
// conv2d_x3.hpp) and never reach memory; arguments as effi_encoder_inputs_bf16x3_sr (volume_ops.hip) + the pair's weights.  Bitwise equal
// to the two launches it replaces.
extern "C" int EFFI_FN(effi_encoder_pair_gen_bf16x3_sr)(const float* inv_depth, const float* disp_range, int n_range, const float* interval,
                                                        const float* cur_vol, long cds, long cps, int Dcur, const float* reg_vol, long rds,
                                                        long rps, int Dreg, const float* dmin, const float* dmax, long range_ps, int nq,
                                                        int h, int w, const float* weight_c1, const float* bias_c1, const float* weight_d1,
                                                        const float* bias_d1, int hd, const void* wpack_c2, const float* bias_c2,
                                                        void* out_sr_c2, const void* wpack_d2, const float* bias_d2, void* out_sr_d2,
                                                        int cout, int hp, int wp, int act, effi_stream_t stream) {
    if (!inv_depth || !interval || !cur_vol || !reg_vol || !dmin || !dmax || !weight_c1 || !bias_c1 || !weight_d1 || !bias_d1 || !disp_range ||
        n_range < 2 || Dcur < 2 || Dreg < 2)
        return EFFI_ERR_BADARG;
    if (act < EFFI_ACT_NONE || act > EFFI_ACT_TANH || (cout & 15) || !out_sr_c2 || !out_sr_d2) return EFFI_ERR_BADARG;
    if ((reinterpret_cast<uintptr_t>(out_sr_c2) | reinterpret_cast<uintptr_t>(out_sr_d2)) & 15) return EFFI_ERR_BADARG;
    if (nq != 3 || (hd != 16 && hd != 32 && hd != 48)) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a0, a1;
    const void* dummy[1] = {out_sr_c2};            // the source list only carries the channel count here: nothing is read through it
    int rc = fill_sr(a0, dummy, &hd, 1, wpack_c2, bias_c2, cout, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    rc = fill_sr(a1, dummy, &hd, 1, wpack_d2, bias_d2, cout, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    a0.act = a1.act = act;
    a0.out_sr = reinterpret_cast<unsigned short*>(out_sr_c2);
    a1.out_sr = reinterpret_cast<unsigned short*>(out_sr_d2);
    const EncGenArgs g{inv_depth, disp_range, n_range, interval, cur_vol, cds, cps, Dcur, reg_vol, rds, rps, Dreg, dmin, dmax, range_ps,
                       weight_c1, bias_c1, weight_d1, bias_d1, hd};
    hipStream_t st = effi_s(stream);
    switch (cout / 16) {
        case 1: return launch_bf16x3_encgen_pair<1>(a0, a1, g, st);
        case 2: return launch_bf16x3_encgen_pair<2>(a0, a1, g, st);
        case 3: return launch_bf16x3_encgen_pair<3>(a0, a1, g, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

extern "C" int EFFI_FN(effi_conv2d_k3_k1_bf16x3_sr)(const void* const* srcs, const int* src_channels, int n_src, const void* wpack_bf16,
                                                    const float* bias, int cout1, int relu1, const float* extra, int c_extra,
                                                    const void* w2pack_bf16, const float* bias2, int cout2, int relu, int h, int w,
                                                    int hp, int wp, float* out, void* out_sr, effi_stream_t stream) {
    if (!w2pack_bf16 || !bias2 || (!out && !out_sr) || (out && out_sr)) return EFFI_ERR_BADARG;
    if (cout1 < 1 || cout2 < 1 || c_extra < 0 || (c_extra > 0 && !extra)) return EFFI_ERR_BADARG;
    if (cout1 > 96 || c_extra > 16 || cout2 > 96) return EFFI_ERR_UNSUPPORTED;
    if (out_sr && ((cout2 & 15) || (reinterpret_cast<uintptr_t>(out_sr) & 15))) return EFFI_ERR_BADARG;
    Conv2dArgs a;
    const int rc = fill_sr(a, srcs, src_channels, n_src, wpack_bf16, bias, cout1, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    a.kgroups = relu1 ? 1 : 0;
    a.act = relu ? EFFI_ACT_RELU : EFFI_ACT_NONE;
    a.hd = c_extra;
    a.aux0 = c_extra ? extra : bias2;
    a.aux1 = reinterpret_cast<const float*>(w2pack_bf16);
    a.disp_range = bias2;
    a.n_range = cout2;
    a.out0 = out;
    a.out_sr = reinterpret_cast<unsigned short*>(out_sr);
    hipStream_t st = effi_s(stream);
    switch ((cout1 + 15) / 16) {
        case 1: return launch_bf16x3<1, EFFI_EPI_K1, false, true>(a, st);
        case 2: return launch_bf16x3<2, EFFI_EPI_K1, false, true>(a, st);
        case 3: return launch_bf16x3<3, EFFI_EPI_K1, false, true>(a, st);
        case 4: return launch_bf16x3<4, EFFI_EPI_K1, false, true>(a, st);
        case 6: return launch_bf16x3<6, EFFI_EPI_K1, false, true>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

extern "C" int EFFI_FN(effi_conv2d_k3_k1_up2x_bf16x3_sr)(const void* const* srcs, const int* src_channels, int n_src,
                                                         const void* wpack_bf16, const float* bias, int cout1, const void* w2pack_bf16,
                                                         const float* bias2, const float* inv_depth, const float* disp_range,
                                                         int n_range, int h, int w, int hp, int wp, float* out_depth,
                                                         float* out_depth_inv, effi_stream_t stream) {
    if (!w2pack_bf16 || !bias2 || !inv_depth || !disp_range || n_range < 2 || !out_depth) return EFFI_ERR_BADARG;
    if (cout1 < 1 || cout1 > 96) return EFFI_ERR_UNSUPPORTED;
    Conv2dArgs a;
    const int rc = fill_sr(a, srcs, src_channels, n_src, wpack_bf16, bias, cout1, h, w, hp, wp);
    if (rc != EFFI_OK) return rc;
    a.kgroups = 1;                          // ReLU between the 3x3 and the 1x1 convolution (models/update.py:110)
    a.hd = 0;
    a.aux0 = inv_depth;
    a.aux1 = reinterpret_cast<const float*>(w2pack_bf16);
    a.disp_range = bias2;
    a.n_range = 36;
    a.out0 = out_depth;
    a.out1 = out_depth_inv;
    a.zin = n_range;
    a.xptr0 = disp_range;
    hipStream_t st = effi_s(stream);
    switch ((cout1 + 15) / 16) {
        case 2: return launch_bf16x3<2, EFFI_EPI_K1UP, false, true>(a, st);
        case 4: return launch_bf16x3<4, EFFI_EPI_K1UP, false, true>(a, st);
        case 6: return launch_bf16x3<6, EFFI_EPI_K1UP, false, true>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

#include "gru_fused.hpp"
