// Split-precision ("bf16x3") 3x3 convolution tile of csrc/conv2d.hip, shared with csrc/conv2d_sr.hip (the same kernel reading
// and writing split-resident maps): argument block, transposed-fragment epilogues, the tile function, its kernels and launch rule.
// Included once per translation unit; everything lives in an anonymous namespace (each unit instantiates what it launches).
#pragma once
#include "common.hpp"

// This file is compiled twice (csrc/Makefile): as it stands, and with -DEFFI_BF16_ONLY, which keeps only the *_bf16x3_* entry points,
// appends _bf16 to their names and drops the two lo terms of every split product (hi*hi only: plain bf16 operands, fp32
// accumulation -- BASELINE.json's "bf16 (MFMA 3D-conv path)" configuration).  A compile-time constant, not a runtime flag: a flag
// tested inside the MFMA loops cost the default build 5 % per view (rolling 3-D conv +17 %).
#ifdef EFFI_BF16_ONLY
#define EFFI_FN(name) name##_bf16
#else
#define EFFI_FN(name) name
#endif

// Diagnostic ablation builds of the split-resident tile (tools/ablate_sr.sh; NEVER the shipped library): -DEFFI_ABL=<bits>
//   1 no global loads of the A image   2 no K loop (no LDS fragment reads, no MFMAs)   4 LDS fragment reads but no MFMAs
//   8 epilogue = raw accumulators stored planar (no bias / activation / auxiliary loads / split-resident store)   16 no epilogue
//   32 generated-input kernel: no volume lookups   64 generated-input kernel: nothing generated
#ifndef EFFI_ABL
#define EFFI_ABL 0
#endif
#ifndef EFFI_PIPE_FRAGS
// 1: the K loop requests its LDS fragments one step ahead (see PIPE in the tile function).  Measured (profiles/r04_f_pipe_ab.txt,
// same box, per launch): -2..-4 % at 148x200, +-0 at 296x400, +10..+14 % on the 592x800 pair / encoder kernels (the second fragment
// set costs a wave per SIMD there: 134 -> 174 registers) and -2 % on z | r: not the default.
#define EFFI_PIPE_FRAGS 0
#endif

#ifndef EFFI_EPI_BATCH
// 1: the split-resident tile's plain / GRU epilogues issue their loads together (see "batched epilogue" in the tile function);
// 0: the per-fragment epilogue.
#define EFFI_EPI_BATCH 1
#endif
#ifndef EFFI_EPI_EARLY_MAX
// the batched epilogue's state quads are requested AHEAD of the K loop when they take at most this many registers per lane
// (held across the loop), else right behind it
#define EFFI_EPI_EARLY_MAX 16
#endif
#ifndef EFFI_B_EARLY
// 1: the next chunk's weight fragments are requested before the K loop of the current one (held in registers across it) instead of
// between the two barriers that separate the chunks.
#define EFFI_B_EARLY 0
#endif

namespace {

#ifdef EFFI_BF16_ONLY
constexpr bool kHiOnly = true;
#else
constexpr bool kHiOnly = false;
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct Conv2dArgs {
    const float* src[EFFI_MAX_SRC];
    int ch[EFFI_MAX_SRC];
    int cin;                 // real input channels (sum of ch[])
    int kgroups;             // ceil(cin / 4)
    const float* wpack;      // [kgroups][KS*KS][NT][64]
    const float* bias;       // [NT*16]
    int cout, h, w, act, hd;
    const float* aux0;
    const float* aux1;
    const float* disp_range;
    int n_range;
    float* out0;
    float* out1;
    // z-batched use (3-D convolution as per-plane 2-D convolutions, effi_conv3d_k3s1_mfma_f32): blockIdx.y = z,
    // source s is plane z + s - 1 of the SAME [cin][D][h][w] tensor (zero when outside), channel strides are D*h*w
    long cstride;            // input channel stride in floats (hin*win for plain 2-D)
    long ostride;            // output channel stride
    int zcount;              // output planes when z-batched, else 0
    int zin;                 // input planes (= zcount for stride 1; stride-2 3-D convs read a 2x deeper volume)
    int hin, win;            // input map size (= h, w for stride 1; stride-2 convs read a 2x larger map)
    const float* zeros;      // >= 64 B of zeros in device memory: where padding is read from (split-precision kernels)
    const float* xptr0;      // EPI_K1UP: the hypotheses' inverse-depth range (first / last entry used), zin = its length
    // split-resident form (SR kernels, conv2d_sr.hip; see effi_sr_store4): src[] point at bf16 maps [ch/8][hi|lo][sr_hp][sr_wp][8]
    // with a zero border (pixel (y, x) at row y + 1, column x + 1; every ch[] a multiple of 16), out_sr (or nullptr) receives the
    // result in the same form.  Unused (and not initialised) by the planar kernels.
    int sr_hp, sr_wp;
    unsigned short* out_sr;
    int aux_q4;              // SR GRU epilogues: aux0 / aux1 / out0 are [channels/4][h][w][4] fp32 maps (EFFI_EPI_Q4)
};

// GENERATED inputs of the encoder's convc2 / convd2 (models/update.py:86-91): instead of reading relu(convc1(GetCost(inv_depth))) /
// relu(convd1(inv_depth)) as maps that another launch wrote, the convolution's workgroup evaluates them for its own tile (+ halo)
// straight into the LDS image -- from the inverse-depth map and the stage's two cached volumes (a few values per pixel).  The
// arithmetic is that of getcost_conv1x1_block (volume_ops.hip) / effi_c1k7_relu_tile_x3 (common.hpp), operation for operation, so
// the result is bitwise the one of encoder_inputs + pair launch; what disappears is one launch and 2 x hd channels written and read
// back per GRU iteration.
struct EncGenArgs {
    const float* inv_depth; const float* disp_range; int n_range; const float* interval;
    const float* cur_vol; long cds, cps; int Dcur;
    const float* reg_vol; long rds, rps; int Dreg;
    const float* dmin; const float* dmax; long range_ps;
    const float* w_c1; const float* b_c1;          // convc1: [6][hd], [hd]
    const float* w_d1; const float* b_d1;          // convd1 (7x7): [49][hd], [hd]
    int hd;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case EFFI_ACT_RELU: return fmaxf(v, 0.0f);
        case EFFI_ACT_SIGMOID: return effi_sigmoid(v);
        case EFFI_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// (effi_sr_store4, the split-resident map layout: common.hpp)
// Epilogue for the TRANSPOSED fragment layout (MFMA called as weights x pixels, D[cout][pixel]): a lane holds output
// channels co0..co0+3 of ONE pixel, and the 16 lanes of a lane group hold 16 consecutive pixels of a row, so every store /
// auxiliary load of a wave instruction is 4 runs of 64 contiguous bytes written by ADJACENT lanes (they coalesce), instead of
// 16 scattered 16-byte pieces per run as in the pixel-major layout above -- the store phase of the planar epilogues was the
// largest single cost of the memory-bound layers.  Channel-last output becomes one float4 per lane, 1 KB contiguous per wave.
template <int EPI, bool SR = false>
__device__ __forceinline__ void conv_epilogue_store_t(const Conv2dArgs& a, const f32x4& acc, int co0, long pix, long hw,
                                                      int zpl, int y = 0, int x = 0, const f32x4* biasq = nullptr) {
    constexpr bool kShuf = (EPI == EFFI_EPI_ADD_SHUF2 || EPI == EFFI_EPI_NHWC_ADD_SHUF2);
    constexpr bool kNhwc = (EPI == EFFI_EPI_NHWC || EPI == EFFI_EPI_NHWC_ADD_SHUF2);
    constexpr bool kPlain = (EPI == EFFI_EPI_PLAIN || EPI == EFFI_EPI_ADD_SHUF2);
    float v[4];
    // biasq: the caller's copy of bias[co0 .. co0 + 3], fetched ahead of the multiplies (the rolling 3-D kernels: once per workgroup
    // instead of a dependent load per plane); same values
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] + (biasq ? (*biasq)[r] : a.bias[co0 + r]);       // bias is padded to 16*NT entries
    if (kNhwc || kPlain) {                                             // activation: one uniform branch for the 4 values
        if (a.act == EFFI_ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
        } else if (a.act != EFFI_ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], a.act);
        }
    }
    if (kShuf) {
        // + a coarser map with the four sub-pixel parities as channel groups (pixel shuffle): aux0 planar [4*cout][h/2][w/2],
        // channel ((y & 1) * 2 + (x & 1)) * cout + co at (y >> 1, x >> 1) -- the nearest-upsampled branch of the pyramid's last
        // head, evaluated at half resolution (models/module.py:407-408, see packing.pack_fpn_head_split)
        const int y = (int)(pix / a.w), x = (int)(pix - (long)y * a.w);
        const long hw4 = (long)(a.h >> 1) * (a.w >> 1);
        const float* up = a.aux0 + (long)(((y & 1) * 2 + (x & 1)) * a.cout + co0) * hw4 + (long)(y >> 1) * (a.w >> 1) + (x >> 1);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co0 + r < a.cout) v[r] = v[r] + up[(long)r * hw4];
    }
    if (kNhwc) {
        if (co0 + 3 < a.cout) {
            *reinterpret_cast<float4*>(a.out0 + pix * a.cout + co0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (co0 + r < a.cout) a.out0[pix * a.cout + co0 + r] = v[r];
        }
        return;
    }
    // channel guard: one test per group of 4 when cout is a multiple of 4 (every layer of the model), else per channel
    const int nvalid = ((a.cout & 3) == 0) ? (co0 < a.cout ? 4 : 0) : a.cout - co0;
    if (kPlain) {
        if (SR) {                                 // cout % 16 == 0 (host): all four channels are real
            if (nvalid > 0) effi_sr_store4(a.out_sr, a.sr_hp, a.sr_wp, co0, y, x, f32x4{v[0], v[1], v[2], v[3]});
            if (!a.out0) return;
        }
        float* dst = a.out0 + (long)co0 * a.ostride + (long)zpl * hw + pix;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) dst[(long)r * a.ostride] = v[r];
    } else if (EPI == EFFI_EPI_GRU_ZR) {         // co0 is a multiple of 4 and hd of 16: the 4 channels are all z or all r
        if (nvalid <= 0) return;                 // cout % 16 == 0 for the GRU epilogues (checked by the host): all or nothing
        const bool is_z = co0 < a.hd;
        if (SR && a.aux_q4) {                     // fp32 state and z in the Q4 layout: one 16-byte access per map (uniform branch)
            const long oq = ((long)((is_z ? co0 : co0 - a.hd) >> 2) * hw + pix) * 4;
            f32x4 g;
            if (is_z) {
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = effi_sigmoid_split(v[r]) * 1.0f;
                *reinterpret_cast<f32x4*>(a.out0 + oq) = g;
            } else {
                const f32x4 h4 = *reinterpret_cast<const f32x4*>(a.aux0 + oq);
#pragma unroll
                for (int r = 0; r < 4; ++r) g[r] = effi_sigmoid_split(v[r]) * h4[r];
                effi_sr_store4(a.out_sr, a.sr_hp, a.sr_wp, co0 - a.hd, y, x, g);
            }
            return;
        }
        const long o = (long)(is_z ? co0 : co0 - a.hd) * hw + pix;
        float* dst = (is_z ? a.out0 : a.out1) + o;
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = a.aux0[o + (long)r * hw];              // r*h needs h; harmless extra read for z
        if (SR) {                                 // z stays fp32 (it gates in the q epilogue); r * h feeds convq: split-resident
            f32x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = effi_sigmoid_split(v[r]) * (is_z ? 1.0f : hv[r]);
            if (is_z) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(long)r * hw] = g[r];
            } else {
                effi_sr_store4(a.out_sr, a.sr_hp, a.sr_wp, co0 - a.hd, y, x, g);
            }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(long)r * hw] = effi_sigmoid_split(v[r]) * (is_z ? 1.0f : hv[r]);
    } else if (EPI == EFFI_EPI_GRU_Q) {
        if (nvalid <= 0) return;
        if (SR && a.aux_q4) {
            const long oq = ((long)(co0 >> 2) * hw + pix) * 4;
            const f32x4 h4 = *reinterpret_cast<const f32x4*>(a.aux0 + oq), z4 = *reinterpret_cast<const f32x4*>(a.aux1 + oq);
            f32x4 g;
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = (1.0f - z4[r]) * h4[r] + z4[r] * effi_tanh_split(v[r]);
            *reinterpret_cast<f32x4*>(a.out0 + oq) = g;
            effi_sr_store4(a.out_sr, a.sr_hp, a.sr_wp, co0, y, x, g);
            return;
        }
        const long o = (long)co0 * hw + pix;
        float hv[4], zv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            hv[r] = a.aux0[o + (long)r * hw];
            zv[r] = a.aux1[o + (long)r * hw];
        }
        f32x4 g;
#pragma unroll
        for (int r = 0; r < 4; ++r) g[r] = (1.0f - zv[r]) * hv[r] + zv[r] * effi_tanh_split(v[r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out0[o + (long)r * hw] = g[r];
        if (SR) effi_sr_store4(a.out_sr, a.sr_hp, a.sr_wp, co0, y, x, g);      // the new hidden state in both forms (fp32: gating, output)
    }
}

// ------------------------------------------------------------------------------------------------
// Split-precision 3x3 convolution ("bf16x3"): every fp32 operand is written as hi + lo with hi = bf16(x),
// lo = bf16(x - hi), and a product is evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32
// accumulation (the dropped lo*lo term and the bf16 rounding of lo are ~2^-17 relative, i.e. ~1e-5 vs 6e-8 for the
// exact fp32 MFMA).  One bf16 MFMA covers K = 32 in 16 cycles where the fp32 MFMA covers K = 4 in 32-40, so the three
// MFMAs per product are still ~6x cheaper; the three run back to back on the same accumulator.
//   * chunk = 16 input channels.  K index inside a chunk = (tap, octet of 8 channels); a K-step of 32 = 4 such items
//     (lane quarter q = lane>>4 owns item 4s+q), 9 taps x 2 octets = 18 items -> 5 K-steps (the last two items are zero).
//   * A tile in LDS is bf16 [octet][pixel][8 ch] for hi and for lo, so a lane's fragment (8 consecutive channels of one
//     pixel at one tap) is ONE ds_read_b128 and the 16 pixels of a lane group sit in 16 consecutive 16-byte slots = all 64
//     banks (conflict-free whatever the tap shift).  Staging does the transposition in registers: a thread owns (4 pixels,
//     8 channels) = 8 coalesced float4 loads from the planar fp32 map, splits them and writes 2 x 4 ds_write_b128; pixel
//     slot p is stored at p ^ ((p >> 3) & 1), which keeps the reads conflict-free and makes the stores 2-way (13 vs 16
//     LDS cycles) instead of 4-way.
//   * B fragments are pre-split and pre-ordered by the host ([chunk][K-step][N-tile][hi|lo][lane][8] bf16) and copied to LDS.
// Same tiling (16 x 4*MR pixels per workgroup), prefetch-into-registers structure and epilogues as the fp32 kernel.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ZB = z-batched (3-D convolution, effi_conv3d_k3s1_bf16x3_f32): blockIdx.y = output plane z, the effective input channels
// are (dz, c) = cat over dz of the sources' channels at plane z + dz - 1 (zero outside), channel strides D*h*w.
//
// These layers are bound by the vector-ALU instruction count around the MFMAs (measured: 751 VALU instructions per wave and
// tile for 60 MFMAs before this form), so the staging is written for few instructions:
//   * every source but the last has a multiple of 8 channels (checked by the host), so a thread's octet of 8 channels lies in
//     one source: ONE source/plane selection per chunk, then 8 loads at p + e*cstride;
//   * padding outside the map / the volume is read from a zero page (base and channel step are selected once per chunk)
//     instead of being masked; channels beyond cin re-read the last real channel (their weights are zero);
//   * fp32 -> (hi, lo) uses the packed conversion (v_cvt_pk_bf16_f32) on channel pairs;
//   * A fragments sit at lane_base + koff[s] + m*row with the row term as an immediate (no address swizzle: the stores are then
//     4-way instead of 2-way conflicted, ~0.5k LDS cycles per tile, against ~150 address instructions per wave).
#define EFFI_EPI_K1 6        // internal: the 3x3 result (+ extra channels) goes through a fused 1x1 convolution (see below)
#define EFFI_EPI_K1UP 8      // internal: K1 producing the 36-channel convex-upsampling mask, consumed in registers (see below)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 fp32 channels of 4 pixels (pa[e] = 4 pixels of channel e) -> hi/lo bf16x8 of pixel px
__device__ __forceinline__ void split_octet(const f32x4 (&pa)[8], int px, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const f32x2 x = {pa[e][px], pa[e + 1][px]};
        const bf16x2 h2 = __builtin_convertvector(x, bf16x2);
        const f32x2 hf = __builtin_convertvector(h2, f32x2);
        const bf16x2 l2 = __builtin_convertvector(x - hf, bf16x2);
        hi[e] = h2[0];
        hi[e + 1] = h2[1];
        lo[e] = l2[0];
        lo[e + 1] = l2[1];
    }
}

// WIDE: the workgroup's tile is 4 rows x 16*MR columns (one row per wave, MR column groups per wave) instead of 4*MR rows x 16
// columns: same LDS image size and halo factor, but every row segment the tile reads (64*MR + 32 bytes) and writes (64*MR bytes)
// is MR times longer, which is what HBM likes once the working set no longer fits the 256 MB MALL.
//
// SR: the sources are split-resident maps (see effi_sr_store4): staging is a 16-byte copy per (pixel, octet, part) -- no conversion,
// no bounds tests (zero border), halo of one pixel instead of the four a float4-aligned fp32 row needs (18 instead of 24 staged
// columns per 16 outputs).  The LDS image is ONE array of 16-byte units [part][octet][pixel] (octet planes padded to a multiple of
// 16 pixels: a ds_read_b128 lane group mixes two lane quarters, i.e. two octet planes, and stays conflict-free only if the planes
// are congruent mod 16 slots), filled by unit u = tid + 256 j; units beyond the image (the last pass) land in its padding.
// NW: waves per workgroup (4, or 8 for SR kernels: twice the rows behind ONE copy of the weight fragments -- at the same LDS budget
// per CU six waves per SIMD instead of four, and half the L2 -> LDS weight traffic).
// GEN (with SR): the A image is generated (EncGenArgs); gmode (workgroup-uniform) 0 = relu(convc1(GetCost(inv_depth))), 1 =
// relu(convd1(inv_depth)) (7x7) -- ONE instantiation for both so that the two halves of the pair kernel share their LDS.
template <int NT, int MR, int EPI, bool ZB, bool WIDE, bool SR = false, int NW = 4, bool GEN = false>
__device__ __forceinline__ void conv2d_k3_bf16x3_tile(const Conv2dArgs a, int tiles_x, int ntiles, int bid, int nbid, int bidy,
                                                      const EncGenArgs* gp = nullptr, int gmode = 0) {
    constexpr int NTHR = NW * 64;
    static_assert(!GEN || SR, "generated inputs use the split-resident LDS image");
    static_assert(NW == 4 || (SR && NW == 8), "eight waves: split-resident kernels only");
    constexpr int TR = WIDE ? NW : NW * MR, TW = WIDE ? 16 * MR : 16;
    constexpr int AR = TR + 2, AW = SR ? TW + 2 : TW + 8, AQ = AW / 4, XOFF = SR ? 0 : 3, XLEFT = 4, CCH = 16, NKS = 5;
    constexpr int MROW = WIDE ? 0 : AW * 8, MCOL = WIDE ? 16 * 8 : 0;   // LDS element step between a wave's MR sub-tiles
    constexpr int APIX = AR * AW, NITEMS = (APIX / 4) * 2;             // staging work items: (pixel quad, octet)
    constexpr int APIXP = SR ? ((APIX + 15) & ~15) : APIX;             // pixels per octet plane of the LDS image
    constexpr int NPART = kHiOnly ? 1 : 2;
    constexpr int NUA = SR ? (NPART * 2 * APIXP + NTHR - 1) / NTHR : 1; // SR: staging passes of NTHR 16-byte units
    constexpr int NBF = NKS * NT * 2 * 64;                             // 16-byte units of B per chunk
    constexpr int NB4 = (NBF + NTHR - 1) / NTHR;
    static_assert(SR || NITEMS <= 256, "one staging item per thread");
    static_assert(!(SR && ZB), "split-resident maps are 2-D");
    static_assert(EPI != EFFI_EPI_HEAD && EPI != EFFI_EPI_ADD_UP2, "epilogue not instantiated for the split-precision kernel");
    static_assert((EPI != EFFI_EPI_K1 && EPI != EFFI_EPI_K1UP) || !ZB, "the fused 1x1 epilogue is 2-D only");
    __shared__ __attribute__((aligned(16))) unsigned short lds_ah[SR ? NUA * NTHR * 8 : APIX * CCH];
    __shared__ __attribute__((aligned(16))) unsigned short lds_al[SR ? 8 : APIX * CCH];
    __shared__ __attribute__((aligned(16))) unsigned short lds_b[NB4 * NTHR * 8];
    const unsigned short* const lds_al_rd = SR ? lds_ah + 2 * APIXP * 8 : lds_al;      // SR: part 1 of the one image
    // gmode 1: the inverse-depth tile of the staged region grown by the 7x7 kernel's reach (+ one zero-weighted row; 8 columns read)
    constexpr int IW7 = (AW + 7 + 1) & ~1, IH7 = AR + 7;
    __shared__ float lds_inv[GEN ? IH7 * IW7 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int h = a.h, w = a.w;
    const long hw = (long)h * w;
    const int tile = effi_xcd_remap(bid, nbid);
    if (tile >= ntiles) return;
    const int ty_ = tile / tiles_x;
    const int x0 = (tile - ty_ * tiles_x) * TW, y0 = ty_ * TR;
    const int zpl = ZB ? bidy : 0;

    // staging item of this thread
    const bool stager = tid < NITEMS;
    const int pq = stager ? tid % (APIX / 4) : 0, soct = stager ? tid / (APIX / 4) : 0;
    const int srow = pq / AQ, sqx = pq - srow * AQ;
    const int sgy = y0 - 1 + srow, sgx = x0 - XLEFT + 4 * sqx;
    const bool s_in = stager & (sgy >= 0) & (sgy < h) & (sgx >= 0) & (sgx < w);
    const int s_off = sgy * w + sgx;
    const int s_lds = (soct * APIX + srow * AW + 4 * sqx) * 8;         // bf16 index of the quad in its octet plane
    const int s_nv = w - sgx;                                          // elements of the quad inside its row (ZB: w % 4 may be != 0)
    const bool s_part = s_nv < 4;

    f32x4 pa[SR ? NUA : 8];                           // raw loads: pa[e] = 4 pixels of channel e of the octet (SR: unit tid + 256 e)
    const int cin_eff = ZB ? 3 * a.cin : a.cin;
    const int nchunks = (cin_eff + CCH - 1) / CCH;
    const unsigned short* wbf = reinterpret_cast<const unsigned short*>(a.wpack);
    // SR: where unit tid + 256 j of a chunk's image lies in the chunk's four planes of the map (16-byte units; constant per tile)
    int goff[NUA];
    if (SR && !GEN) {
#pragma unroll
        for (int j = 0; j < NUA; ++j) {
            const int u = tid + j * NTHR;
            const int part = u / (2 * APIXP), r_ = u - part * (2 * APIXP);
            const int oct = r_ / APIXP, p = r_ - oct * APIXP;
            const int row = p / AW, col = p - row * AW;
            const bool valid = (part < NPART) & (p < APIX);
            goff[j] = valid ? ((oct * 2 + part) * a.sr_hp + (y0 + row)) * a.sr_wp + (x0 + col) : 0;
        }
    }
    auto prefetch = [&](int ch) {
        if constexpr (GEN) {
            (void)ch;
        } else if constexpr (SR) {
            // every source has a multiple of 16 channels (host): the chunk lies in one source, at its local chunk index
            const int cb = ch * CCH;
            const int c1 = cb - a.ch[0], c2 = c1 - a.ch[1];
            const float* src = (c1 < 0) ? a.src[0] : (c2 < 0 ? a.src[1] : a.src[2]);
            const int cl = (c1 < 0) ? cb : (c2 < 0 ? c1 : c2);
            const f32x4* base = reinterpret_cast<const f32x4*>(src) + (long)(cl >> 2) * ((long)a.sr_hp * a.sr_wp);   // 4 planes per chunk
#pragma unroll
            for (int j = 0; j < NUA; ++j) pa[j] = (EFFI_ABL & 1) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : base[goff[j]];
        } else {
        int cb = ch * CCH + soct * 8;                                  // first channel of this thread's octet
        const int emax = min(cin_eff - cb, 8) - 1;                     // last real channel of the octet (< 0: none)
        bool in = s_in & (emax >= 0);
        long zoff = 0;
        if (ZB) {
            const int dz = (cb >= a.cin) + (cb >= 2 * a.cin);          // a.cin % 8 == 0: the octet lies in one plane
            cb -= dz * a.cin;
            const int zz = zpl + dz - 1;
            in &= (zz >= 0) & (zz < a.zcount);
            zoff = (long)zz * hw;
        }
        const int c1 = cb - a.ch[0], c2 = c1 - a.ch[1];
        const float* src = (c1 < 0) ? a.src[0] : (c2 < 0 ? a.src[1] : a.src[2]);
        const int cl = (c1 < 0) ? cb : (c2 < 0 ? c1 : c2);
        const float* base = in ? src + ((long)cl * a.cstride + zoff + s_off) : a.zeros;
        const long step = in ? a.cstride : 0;
        const float* q = base;
        if (ZB && in && s_part) {
            // rows that are not a multiple of 4 long (the U-Net's coarsest level is 50 wide at 1600x1184): the quad that straddles the
            // row end takes its 1-3 valid elements one by one (the vector load would pick up the next row and could leave the tensor)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t0 = q[0], t1 = (s_nv > 1) ? q[1] : 0.0f, t2 = (s_nv > 2) ? q[2] : 0.0f;
                pa[e] = f32x4{t0, t1, t2, 0.0f};
                q += (e < emax) ? step : 0;
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pa[e] = *reinterpret_cast<const f32x4*>(q);
            q += (e < emax) ? step : 0;                                // channels past the last real one re-read it
        }
        }
    };
    // ---- generated A image (GEN): per-pixel set-up once per tile, one 16-channel chunk per call ----
    constexpr int PPT = GEN ? (APIX + NTHR - 1) / NTHR : 1;          // gmode 0: staged pixels per thread
    float gcost[PPT][6];
    bool gin[PPT];
    if (GEN && gmode == 0 && (EFFI_ABL & 32)) {       // ablation: no lookups
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            gin[i] = true;
#pragma unroll
            for (int k = 0; k < 6; ++k) gcost[i][k] = 0.25f * k;
        }
    } else
    if (GEN && gmode == 0) {
        const EncGenArgs& g = *gp;
        const float itv = g.interval[0];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int p = tid + i * NTHR;
            const int row = p / AW, col = p - row * AW;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            gin[i] = (p < APIX) & (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w);
            const long pix = gin[i] ? (long)gy * w + gx : 0;
            effi_getcost_pixel<3>(g.inv_depth[pix], 0, g.disp_range, g.n_range, itv, g.cur_vol + pix * g.cps, g.cds, g.Dcur,
                                  g.reg_vol + pix * g.rps, g.rds, g.Dreg, g.dmin[pix * g.range_ps], g.dmax[pix * g.range_ps], gcost[i]);
        }
    }
    if (GEN && gmode != 0) {
        const EncGenArgs& g = *gp;
        for (int e = tid; e < IH7 * IW7; e += NTHR) {
            const int yy = e / IW7, xx = e - yy * IW7;
            const int gy = y0 - 1 - 3 + yy, gx = x0 - 1 - 3 + xx;
            lds_inv[e] = (yy < IH7 - 1 && gy >= 0 && gy < h && gx >= 0 && gx < w) ? g.inv_depth[(long)gy * w + gx] : 0.0f;
        }
        __syncthreads();
    }
    auto generate = [&](int ch) {
        if (EFFI_ABL & 64) return;                     // ablation: no generated image at all (LDS left as it is)
        if (GEN && gmode == 0) {
            const EncGenArgs& g = *gp;
            const int hd = g.hd;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int p = tid + i * NTHR;
                if (p >= APIX) continue;
#pragma unroll
                for (int oct = 0; oct < 2; ++oct) {
                    const int c0 = ch * CCH + oct * 8;
                    float acc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = g.b_c1[c0 + j];
#pragma unroll
                    for (int k = 0; k < 6; ++k)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(gcost[i][k], g.w_c1[k * hd + c0 + j], acc[j]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = gin[i] ? fmaxf(acc[j], 0.0f) : 0.0f;        // outside the map: the 3x3's zero padding
                    effi_bf16x8_t hi, lo;
                    effi_split8(acc, hi, lo);
                    *reinterpret_cast<effi_bf16x8_t*>(&lds_ah[((0 * 2 + oct) * APIXP + p) * 8]) = hi;
                    if (!kHiOnly) *reinterpret_cast<effi_bf16x8_t*>(&lds_ah[((1 * 2 + oct) * APIXP + p) * 8]) = lo;
                }
            }
        }
        if (GEN && gmode != 0) {
            const EncGenArgs& g = *gp;
            const int hd = g.hd;
            // weight fragments of this chunk's 16 channels: lane (cout j = li, quarter lk) holds W[16 ch + j][ky = 4 s + lk][kx = 0..7]
            effi_bf16x8_t wh[2], wl[2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const int ky = 4 * s_ + lk;
                float wv8[8];
#pragma unroll
                for (int kx = 0; kx < 8; ++kx) wv8[kx] = (ky < 7 && kx < 7) ? g.w_d1[(ky * 7 + kx) * hd + CCH * ch + li] : 0.0f;
                effi_split8(wv8, wh[s_], wl[s_]);
            }
            float b4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b4[r] = g.b_d1[CCH * ch + 4 * lk + r];
            constexpr int NSEG = (APIX + 15) / 16;                    // groups of 16 consecutive staged pixels (a group may wrap a row)
            for (int seg = wv; seg < NSEG; seg += NW) {
                const int pr = seg * 16 + li, p = min(pr, APIX - 1);
                const int row = p / AW, col = p - row * AW;
                const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                const bool inside = (gy >= 0) & (gy < h) & (gx >= 0) & (gx < w);
                f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_) {
                    const float* rowp = lds_inv + (row + 4 * s_ + lk) * IW7 + col;
                    float x8[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) x8[i] = rowp[i];
                    effi_bf16x8_t ah, al;
                    effi_split8(x8, ah, al);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s_], ah, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[s_], ah, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[s_], al, acc, 0, 0, 0);
                }
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inside ? fmaxf(acc[r] + b4[r], 0.0f) : 0.0f;
                const bf16x4 h4 = __builtin_convertvector(v, bf16x4);
                const bf16x4 l4 = __builtin_convertvector(v - __builtin_convertvector(h4, f32x4), bf16x4);
                if (pr < APIX) {
                    const int e = ((lk >> 1) * APIXP + p) * 8 + (lk & 1) * 4;      // octet lk >> 1 of the chunk, half lk & 1
                    *reinterpret_cast<bf16x4*>(&lds_ah[e]) = h4;
                    if (!kHiOnly) *reinterpret_cast<bf16x4*>(&lds_ah[2 * APIXP * 8 + e]) = l4;
                }
            }
        }
    };

    // A: split + transpose out of the prefetch registers.  B (pre-split by the host, L2-resident, identical for every
    // workgroup) is copied global -> LDS (all its loads are issued before the first use; the LDS image is padded to whole
    // 256-thread passes so the copy needs no predicate).
    f32x4 tb[NB4];
    auto fetch_b = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NB4; ++j) {
            const int u = min(tid + j * NTHR, NBF - 1);
            tb[j] = *reinterpret_cast<const f32x4*>(wbf + ((long)ch * NBF + u) * 8);
        }
    };
    constexpr bool kBEarly = SR && !GEN && EFFI_B_EARLY;
    auto stash = [&](int ch) {
        if (!kBEarly || ch == 0) fetch_b(ch);
        if constexpr (GEN) {
            generate(ch);
        } else if constexpr (SR) {
#pragma unroll
            for (int j = 0; j < NUA; ++j) *reinterpret_cast<f32x4*>(&lds_ah[(tid + j * NTHR) * 8]) = pa[j];
        } else if (stager) {
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                bf16x8 hi, lo;
                split_octet(pa, px, hi, lo);
                *reinterpret_cast<bf16x8*>(&lds_ah[s_lds + px * 8]) = hi;
                if (!kHiOnly) *reinterpret_cast<bf16x8*>(&lds_al[s_lds + px * 8]) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < NB4; ++j) *reinterpret_cast<f32x4*>(&lds_b[(tid + j * NTHR) * 8]) = tb[j];
    };

    // fragment addressing: lane (pixel li, quarter lk) owns item 4s + lk = (tap, octet) of K-step s
    int koff[NKS];
#pragma unroll
    for (int s_ = 0; s_ < NKS; ++s_) {
        const int item = 4 * s_ + lk;
        const int tap = min(item >> 1, 8), oct = item & 1;               // items 18, 19 are padding (B is zero there)
        koff[s_] = (((WIDE ? wv : wv * MR) + tap / 3) * AW + li + XOFF + tap % 3 + oct * APIXP) * 8;
    }

    f32x4 acc[MR][NT];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // BATCHED EPILOGUE (split-resident plain / GRU forms).  Written per fragment (conv_epilogue_store_t) the epilogue is a chain of
    // dependent memory round trips -- per (m, n) fragment: the bias quad, wait, the state quad(s), wait, store -- under divergent
    // bounds tests the compiler cannot move loads across: 8 serialised round trips per wave for the stage-3 z | r kernel, 10-24 us of
    // the 30-44 us such a launch takes (ablation, profiles/r04_e_ablate_sr.txt: "no epilogue").  Here every load of the epilogue is
    // issued at once from clamped addresses (out-of-map lanes read pixel 0), the bias quads before the first chunk, the state quads
    // (Q4 layout: one 16-byte load per fragment and map) ahead of the K loop when few (EFFI_EPI_EARLY_MAX), else right behind it; the unit /
    // offset of every store derives from ONE base per sub-tile.  Same operations on the same values: bitwise the per-fragment form.
    constexpr bool kBatchT = SR && EFFI_EPI_BATCH != 0 && !(EFFI_ABL & 24) &&
                             (EPI == EFFI_EPI_PLAIN || EPI == EFFI_EPI_GRU_ZR || EPI == EFFI_EPI_GRU_Q);
    constexpr int NZ = (EPI == EFFI_EPI_GRU_ZR) ? NT / 2 : 0;                                    // a.hd = 8 NT (host): tiles [0, NZ) are z
    constexpr int NH = !kBatchT ? 0 : (EPI == EFFI_EPI_GRU_ZR ? NT - NZ : (EPI == EFFI_EPI_GRU_Q ? NT : 0));
    constexpr int NZQ = (kBatchT && EPI == EFFI_EPI_GRU_Q) ? NT : 0;
    constexpr bool kStateEarly = MR * (NH + NZQ) * 4 <= EFFI_EPI_EARLY_MAX;
    const bool ebatch = kBatchT && (EPI == EFFI_EPI_PLAIN || a.aux_q4);
    f32x4 eb[kBatchT ? NT : 1], eh[MR][NH > 0 ? NH : 1], ez[MR][NZQ > 0 ? NZQ : 1];
    unsigned eqb[MR];                                 // Q4 float offset of (channel group lk, the sub-tile's pixel), clamped
    bool einside[MR];
    auto epi_issue_bias = [&]() {
        if (!ebatch) return;
#pragma unroll
        for (int n = 0; n < NT; ++n) eb[n] = *reinterpret_cast<const f32x4*>(a.bias + n * 16 + 4 * lk);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int x = x0 + li + (WIDE ? 16 * m : 0), y = y0 + (WIDE ? wv : wv * MR + m);
            einside[m] = (y < h) & (x < w);
            eqb[m] = ((unsigned)lk * (unsigned)hw + (einside[m] ? (unsigned)(y * w + x) : 0u)) * 4u;
        }
    };
    auto epi_issue_state = [&]() {
        if (!ebatch) return;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int j = 0; j < NH; ++j) eh[m][j] = *reinterpret_cast<const f32x4*>(a.aux0 + (size_t)(eqb[m] + (unsigned)j * 16u * (unsigned)hw));
#pragma unroll
            for (int j = 0; j < NZQ; ++j) ez[m][j] = *reinterpret_cast<const f32x4*>(a.aux1 + (size_t)(eqb[m] + (unsigned)j * 16u * (unsigned)hw));
        }
    };

    // Fused-1x1 epilogues (EPI_K1 / EPI_K1UP): the 3x3 bias quads and (K1) the extra-channel values of the wave's pixels are requested
    // here, unconditionally from clamped addresses, instead of element by element behind bounds branches after the K loop (as the
    // compiler laid that out: 16 single-dword loads in 8 dependent round trips per wave, profiles/r04_e_ablate_sr.txt: 10-13 us of a
    // 21-32 us launch).  Same values: bitwise.
    constexpr bool kK1 = (EPI == EFFI_EPI_K1 || EPI == EFFI_EPI_K1UP) && !(EFFI_ABL & 24) && EFFI_EPI_BATCH != 0;
    f32x4 kb[kK1 ? NT : 1], kex[kK1 ? MR : 1];
    if constexpr (kK1) {
#pragma unroll
        for (int n = 0; n < NT; ++n) kb[n] = *reinterpret_cast<const f32x4*>(a.bias + n * 16 + 4 * lk);
#pragma unroll
        for (int m = 0; m < MR; ++m) kex[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (EPI == EFFI_EPI_K1 && a.hd > 0) {                     // uniform: the layer has extra (context) channels
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int x = x0 + li + (WIDE ? 16 * m : 0), y = y0 + (WIDE ? wv : wv * MR + m);
                const unsigned pix = ((y < h) & (x < w)) ? (unsigned)(y * w + x) : 0u;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * lk + r;
                    const float t_ = a.aux0[(size_t)((unsigned)min(c, a.hd - 1) * (unsigned)hw + pix)];
                    kex[m][r] = (c < a.hd) ? t_ : 0.0f;
                }
            }
        }
    }

    // every other epilogue (planar maps, channel-last heads, z-batched 3-D): the bias quads of the lane's N-tiles, fetched here
    constexpr bool kBiasQ = !kBatchT && !kK1 && !(EFFI_ABL & 24) && EFFI_EPI_BATCH != 0;
    f32x4 gb[kBiasQ ? NT : 1];
    if constexpr (kBiasQ) {
#pragma unroll
        for (int n = 0; n < NT; ++n) gb[n] = *reinterpret_cast<const f32x4*>(a.bias + n * 16 + 4 * lk);
    }

    prefetch(0);
    epi_issue_bias();
    if (kStateEarly) epi_issue_state();
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) {
            prefetch(ch + 1);
            if (kBEarly) fetch_b(ch + 1);
        }
        // SOFTWARE-PIPELINED FRAGMENT READS (PIPE).  As the compiler schedules the plain loop, every K-step opens with its 2 MR + 2 NT
        // ds_read_b128 and waits for them in front of its first MFMA; an ablation (tools/ablate_sr.sh, profiles/r04_e_ablate_sr.txt)
        // put 18 us of the stage-3 z | r kernel's 44 on these exposed reads and only 9 on the MFMAs -- the phases add up instead of
        // overlapping.  Here the pixel fragments of step s + 1 are requested at the start of step s (second register set) and the
        // weight fragments of (s + 1, n) right after the multiplies of (s, n) (into the registers those just freed); LDS returns in
        // order, so each request sits BEHIND what the next multiplies need.  sched_barrier pins the order (the scheduler otherwise
        // sinks every read back to its first use).  Same MFMAs on the same accumulators in the same order: bitwise the plain loop.
        // EFFI_PIPE_FRAGS = 2: only the FRONT half of a wave's pixel fragments gets the second register set; the back half is requested
        // at the start of its own step, behind the front half's multiplies (16 instead of 32 extra registers at MR = 4: the z | r kernel
        // stays at three waves per SIMD).
        constexpr int PIPE = SR ? EFFI_PIPE_FRAGS : 0;
        constexpr int MF = (PIPE == 2) ? (MR + 1) / 2 : MR, MB = MR - MF;          // double-buffered / single-buffered fragments
        bf16x8 ahf[PIPE ? 2 : 1][MF], alf[PIPE ? 2 : 1][MF], ahk[MB > 0 ? MB : 1], alk[MB > 0 ? MB : 1], bhb[NT], blb[NT];
        auto load_front = [&](int s_, int buf) {
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                ahf[buf][m] = *reinterpret_cast<const bf16x8*>(&lds_ah[koff[s_] + m * (MROW + MCOL)]);
                if (!kHiOnly) alf[buf][m] = *reinterpret_cast<const bf16x8*>(&lds_al_rd[koff[s_] + m * (MROW + MCOL)]);
            }
        };
        auto load_back = [&](int s_) {
#pragma unroll
            for (int m = MF; m < MR; ++m) {
                ahk[m - MF] = *reinterpret_cast<const bf16x8*>(&lds_ah[koff[s_] + m * (MROW + MCOL)]);
                if (!kHiOnly) alk[m - MF] = *reinterpret_cast<const bf16x8*>(&lds_al_rd[koff[s_] + m * (MROW + MCOL)]);
            }
        };
        auto load_b = [&](int s_, int n) {
            bhb[n] = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NT + n) * 2 + 0) * 64 + lane) * 8]);
            blb[n] = bhb[n];
            if (!kHiOnly) blb[n] = *reinterpret_cast<const bf16x8*>(&lds_b[(((s_ * NT + n) * 2 + 1) * 64 + lane) * 8]);
        };
        constexpr int NKS_RUN = (EFFI_ABL & 2) ? 0 : NKS;
        if (PIPE && NKS_RUN > 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n) load_b(0, n);
            load_front(0, 0);
        }
#pragma unroll
        for (int s_ = 0; s_ < NKS_RUN; ++s_) {
            const int cur = PIPE ? (s_ & 1) : 0;
            if (PIPE) {
                load_back(s_);
                if (s_ + 1 < NKS) load_front(s_ + 1, cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                load_front(s_, 0);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (!PIPE) load_b(s_, n);
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const bf16x8 ah_m = (m < MF) ? ahf[cur][m < MF ? m : 0] : ahk[m >= MF ? m - MF : 0];
                    const bf16x8 al_m = (m < MF) ? alf[cur][m < MF ? m : 0] : alk[m >= MF ? m - MF : 0];
                    if (EFFI_ABL & 4) {                  // keep the fragment reads alive, issue no MFMA
                        asm volatile("" ::"v"(bhb[n]), "v"(blb[n]), "v"(ah_m), "v"(al_m));
                        continue;
                    }
                    // weights x pixels: D[cout][pixel] (transposed fragment, see conv_epilogue_store_t)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhb[n], ah_m, acc[m][n], 0, 0, 0);
                    if (!kHiOnly) {
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(blb[n], ah_m, acc[m][n], 0, 0, 0);
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bhb[n], al_m, acc[m][n], 0, 0, 0);
                    }
                }
                if (PIPE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s_ + 1 < NKS) load_b(s_ + 1, n);
                }
            }
        }
        if (ch + 1 < nchunks) {
            __syncthreads();
            stash(ch + 1);
            __syncthreads();
        }
    }

    if (EFFI_ABL & 24) {                                 // ablation: raw planar store of the accumulators, or (16) nothing but a sink
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int x = x0 + li + (WIDE ? 16 * m : 0), y = y0 + (WIDE ? wv : wv * MR + m);
            if (y >= h || x >= w) continue;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (EFFI_ABL & 16) {
                        if (acc[m][n][r] == 1.2345e38f) a.out0[0] = 1.0f;
                    } else if (n * 16 + 4 * lk + r < min(a.cout, a.hd > 0 ? a.hd : a.cout)) {
                        a.out0[(long)(n * 16 + 4 * lk + r) * hw + (long)y * w + x] = acc[m][n][r];
                    }
                }
        }
        return;
    }
    if (EPI == EFFI_EPI_K1 || EPI == EFFI_EPI_K1UP) {
        // Fused 1x1 convolution (convd -> convc of the encoder, models/update.py:78-80,93-96): the 3x3 result of a lane
        // -- channels 4*lk..4*lk+3 of pixel li, per N-tile -- is exactly the B fragment of v_mfma_f32_16x16x16_bf16 (K = 16
        // channels), so out2[co2][px] = sum_k W2[co2][k] * cat(conv3x3 + b1, extra)[k][px] needs no data movement: one K = 16 step
        // per N-tile plus one for the extra (context) channels, weights W2 as A fragments from a small L2-resident table.
        // Fields reused: aux0 = extra [c_extra][h][w], hd = c_extra, aux1 = W2 fragments (bf16 [NT2][NT+1][hi|lo][64][4]),
        // disp_range = bias2 (padded to 16*NT2), n_range = cout2, act = activation of the 1x1 result, kgroups = 1 if the 3x3
        // result passes through a ReLU first (mask head, models/update.py:112-114).
        const unsigned short* w2 = reinterpret_cast<const unsigned short*>(a.aux1);
        const int nt2 = (a.n_range + 15) >> 4;
        const bool relu1 = a.kgroups != 0;
        // B fragments of all the wave's pixels first (the extra-channel loads are issued together), then per output tile
        // the W2 fragments are fetched once and reused for the MR sub-tiles
        bf16x4 xh[MR][NT + 1], xl[MR][NT + 1];
        bool inside[MR];
        long pixm[MR];
        f32x4 ex[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const int x = x0 + li + (WIDE ? 16 * m : 0);
            const int y = y0 + (WIDE ? wv : wv * MR + m);
            inside[m] = (y < h) & (x < w);
            pixm[m] = inside[m] ? (long)y * w + x : 0;
            if constexpr (kK1) {
                ex[m] = kex[m];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 4 * lk + r;
                    ex[m][r] = (c < a.hd) ? a.aux0[(long)c * hw + pixm[m]] : 0.0f;   // a.hd == 0: aux0 is a valid dummy, never read
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
#pragma unroll
            for (int n = 0; n <= NT; ++n) {
                f32x4 vf;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    vf[r] = (n < NT) ? acc[m][n][r] + (kK1 ? kb[n < NT ? n : 0][r] : a.bias[n * 16 + 4 * lk + r]) : (inside[m] ? ex[m][r] : 0.0f);
                if (relu1 && n < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) vf[r] = fmaxf(vf[r], 0.0f);
                }
                xh[m][n] = __builtin_convertvector(vf, bf16x4);
                xl[m][n] = __builtin_convertvector(vf - __builtin_convertvector(xh[m][n], f32x4), bf16x4);
            }
        }
        if constexpr (EPI == EFFI_EPI_K1UP) {
            // Mask head + convex upsampling (models/update.py:109-112,136-138 + upsample_depth, models/Effi_MVS_plus.py:167-178 +
            // scale_inv_depth): the 36 mask values of a pixel never reach HBM.  The host orders the rows of the 1x1 convolution so that
            // row 16 t + 4 lk + r is mask entry (tap k = 4 t + r, sub-pixel lk) (packing.pack_mask_taps_per_lane; rows with k > 8 are
            // zero): a lane holds all nine taps of ONE sub-pixel of its pixel, so the softmax over the taps and the weighted sum of the
            // 3x3 inverse-depth neighbourhood need no cross-lane traffic, and every lane stores one output value (the 16 pixels x 2
            // columns of a sub-pixel row are 32 consecutive floats).  aux0 = inverse depth [h][w], out0 / out1 = depth /
            // depth_to_disp(depth) [2h][2w], xptr0 / zin = the hypotheses' range.
            f32x4 om[MR][3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bf16x4 wh[NT + 1], wl[NT + 1];
#pragma unroll
                for (int n = 0; n <= NT; ++n) {
                    const long f = ((long)(t * (NT + 1) + n) * 2) * 64 + lane;
                    wh[n] = *reinterpret_cast<const bf16x4*>(w2 + f * 4);
                    wl[n] = *reinterpret_cast<const bf16x4*>(w2 + (f + 64) * 4);
                }
                const int co = t * 16 + 4 * lk;
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                    for (int n = 0; n <= NT; ++n) {
                        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[n], xh[m][n], o, 0, 0, 0);
                        if (!kHiOnly) {
                            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wl[n], xh[m][n], o, 0, 0, 0);
                            o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[n], xl[m][n], o, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) om[m][t][r] = o[r] + a.disp_range[co + r];
                }
            }
            const float lo = a.xptr0[0], hi = a.xptr0[a.zin - 1];
            const int W2 = 2 * w;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                if (!inside[m]) continue;
                const int x = x0 + li + (WIDE ? 16 * m : 0);
                const int y = y0 + (WIDE ? wv : wv * MR + m);
                float v[9], nbv[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    v[k] = om[m][k >> 2][k & 3];
                    const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
                    const bool ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w);
                    nbv[k] = ok ? a.aux0[ok ? (long)yy * w + xx : 0] : 0.0f;         // F.unfold zero padding
                }
                float mx = v[0];
#pragma unroll
                for (int k = 1; k < 9; ++k) mx = fmaxf(mx, v[k]);
                float e[9], sm = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
#ifdef EFFI_EXACT_EPILOGUES
                    e[k] = expf(v[k] - mx);
#else
                    e[k] = effi_exp_fast(v[k] - mx);                 // argument <= 0: no overflow; see effi_sigmoid_split
#endif
                    sm = sm + e[k];
                }
                float ac = 0.0f;
#ifdef EFFI_EXACT_EPILOGUES
#pragma unroll
                for (int k = 0; k < 9; ++k) ac = ac + (e[k] / sm) * nbv[k];
#else
                const float rsm = effi_rcp_refined(sm);              // 1 <= sm <= 9
#pragma unroll
                for (int k = 0; k < 9; ++k) ac = ac + (e[k] * rsm) * nbv[k];
#endif
                const long o = (long)(2 * y + (lk >> 1)) * W2 + 2 * x + (lk & 1);
                const float dep = effi_inv_to_depth(ac, lo, hi);
                a.out0[o] = dep;
                if (a.out1) a.out1[o] = effi_depth_to_inv(dep, lo, hi);
            }
            return;
        }
        for (int t = 0; t < nt2; ++t) {
            bf16x4 wh[NT + 1], wl[NT + 1];
#pragma unroll
            for (int n = 0; n <= NT; ++n) {
                const long f = ((long)(t * (NT + 1) + n) * 2) * 64 + lane;
                wh[n] = *reinterpret_cast<const bf16x4*>(w2 + f * 4);
                wl[n] = *reinterpret_cast<const bf16x4*>(w2 + (f + 64) * 4);
            }
            const int co = t * 16 + 4 * lk;
            float b2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) b2[r] = a.disp_range[co + r];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int n = 0; n <= NT; ++n) {
                    o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[n], xh[m][n], o, 0, 0, 0);
                    if (!kHiOnly) {
                        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wl[n], xh[m][n], o, 0, 0, 0);
                        o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh[n], xl[m][n], o, 0, 0, 0);
                    }
                }
                if (inside[m]) {
                    f32x4 vo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = o[r] + b2[r];
                        if (a.act == EFFI_ACT_RELU) v = fmaxf(v, 0.0f);
                        vo[r] = v;
                    }
                    if (SR && a.out_sr) {                          // cout2 % 16 == 0 (host)
                        effi_sr_store4(a.out_sr, a.sr_hp, a.sr_wp, co, y0 + (WIDE ? wv : wv * MR + m), x0 + li + (WIDE ? 16 * m : 0), vo);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (co + r < a.n_range) a.out0[(long)(co + r) * hw + pixm[m]] = vo[r];
                    }
                }
            }
        }
        return;
    }
    if constexpr (kBatchT) if (ebatch) {
        if (!kStateEarly) epi_issue_state();
        const unsigned plane = (unsigned)a.sr_hp * (unsigned)a.sr_wp;            // 16-byte units per (octet, part) plane
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (!einside[m]) continue;
            const int x = x0 + li + (WIDE ? 16 * m : 0), y = y0 + (WIDE ? wv : wv * MR + m);
            // channels 16 n + 4 lk ..: octet 2 n + (lk >> 1), part lk & 1 -> plane 4 n + lk
            const unsigned ub = (unsigned)lk * plane + (unsigned)(y + 1) * (unsigned)a.sr_wp + (unsigned)(x + 1);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[m][n][r] + eb[n][r];
                if constexpr (EPI == EFFI_EPI_PLAIN) {
                    if (a.act == EFFI_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f);
                    } else if (a.act != EFFI_ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], a.act);
                    }
                    effi_sr_store4_at(a.out_sr, ub + (unsigned)(4 * n) * plane, lk & 1, v);
                    if (a.out0) {
                        float* dst = a.out0 + (long)(n * 16 + 4 * lk) * a.ostride + (long)y * w + x;
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[(long)r * a.ostride] = v[r];
                    }
                } else if constexpr (EPI == EFFI_EPI_GRU_ZR) {
                    f32x4 g;
                    if (n < NZ) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[r] = effi_sigmoid_split(v[r]) * 1.0f;
                        *reinterpret_cast<f32x4*>(a.out0 + (size_t)(eqb[m] + (unsigned)n * 16u * (unsigned)hw)) = g;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) g[r] = effi_sigmoid_split(v[r]) * eh[m][n >= NZ ? n - NZ : 0][r];
                        effi_sr_store4_at(a.out_sr, ub + (unsigned)(4 * (n - NZ)) * plane, lk & 1, g);
                    }
                } else {
                    f32x4 g;
#pragma unroll
                    for (int r = 0; r < 4; ++r) g[r] = (1.0f - ez[m][n][r]) * eh[m][n][r] + ez[m][n][r] * effi_tanh_split(v[r]);
                    *reinterpret_cast<f32x4*>(a.out0 + (size_t)(eqb[m] + (unsigned)n * 16u * (unsigned)hw)) = g;
                    effi_sr_store4_at(a.out_sr, ub + (unsigned)(4 * n) * plane, lk & 1, g);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        const int x = x0 + li + (WIDE ? 16 * m : 0);                   // lane = (pixel li, channels 4*lk .. 4*lk+3)
        const int y = y0 + (WIDE ? wv : wv * MR + m);
        if (y >= h || x >= w) continue;
        const long pix = (long)y * w + x;
#pragma unroll
        for (int n = 0; n < NT; ++n)
            conv_epilogue_store_t<((EPI == EFFI_EPI_K1 || EPI == EFFI_EPI_K1UP) ? EFFI_EPI_PLAIN : EPI), SR>(a, acc[m][n], n * 16 + 4 * lk, pix, hw, zpl, y, x,
                                                                                                          kBiasQ ? &gb[n] : nullptr);
    }
}

template <int NT, int MR, int EPI, bool ZB = false, bool WIDE = false, bool SR = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv2d_k3_bf16x3_kernel(const Conv2dArgs a, int tiles_x, int ntiles) {
#ifdef EFFI_STAGGER
    // experiment: workgroups that start together on one CU (first come first served: blocks b, b + 256, b + 512 of the first round)
    // run their load / multiply / epilogue phases in step; delay the second and third by a fraction of a tile's time
    if (SR && gridDim.x > 512) {
        const int slot = (blockIdx.x >> 8) % 3;
        for (int i = 0; i < slot * EFFI_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    conv2d_k3_bf16x3_tile<NT, MR, EPI, ZB, WIDE, SR, NW>(a, tiles_x, ntiles, blockIdx.x, gridDim.x, blockIdx.y);
}

// Two independent convolutions of the same shape (NT, h, w) in one launch: blockIdx.y picks the argument set.  Used for the
// update block's convc2 / convd2 (models/update.py:87,91), which would otherwise be forked onto two streams.
template <int NT, int MR, bool WIDE, bool SR = false, int NW = 4>
__global__ __launch_bounds__(NW * 64) void conv2d_k3_bf16x3_pair_kernel(const Conv2dArgs a0, const Conv2dArgs a1, int tiles_x, int ntiles) {
    if (blockIdx.y == 0) conv2d_k3_bf16x3_tile<NT, MR, EFFI_EPI_PLAIN, false, WIDE, SR, NW>(a0, tiles_x, ntiles, blockIdx.x, gridDim.x, 0);
    else conv2d_k3_bf16x3_tile<NT, MR, EFFI_EPI_PLAIN, false, WIDE, SR, NW>(a1, tiles_x, ntiles, blockIdx.x, gridDim.x, 0);
}

// convc2 | convd2 with GENERATED inputs (EncGenArgs): blockIdx.y = 0 -> relu(convc2(relu(convc1(GetCost(inv_depth))))),
// 1 -> relu(convd2(relu(convd1(inv_depth)))); split-resident outputs.  Replaces encoder_inputs + the pair launch.
template <int NT, int MR, bool WIDE>
__global__ __launch_bounds__(256) void conv2d_k3_bf16x3_encgen_pair_kernel(const Conv2dArgs a0, const Conv2dArgs a1, const EncGenArgs g,
                                                                           int tiles_x, int ntiles) {
    // (the two halves stacked along blockIdx.y: interleaving lookup-bound and matrix-bound workgroups along x measured 24 % SLOWER
    // at 592x800)
    conv2d_k3_bf16x3_tile<NT, MR, EFFI_EPI_PLAIN, false, WIDE, true, 4, true>(blockIdx.y == 0 ? a0 : a1, tiles_x, ntiles, blockIdx.x, gridDim.x, 0, &g,
                                                                             (int)blockIdx.y);
}

// ---- split-bf16 3x3 convolution entry --------------------------------------------------------------------------
// Thresholds of the rows-per-wave choice below (workgroup counts); the environment overrides are for A/B runs of the rule
// (with several views in flight the chip is filled by other views' kernels, which favours the larger tiles earlier).
static long effi_opt_or(int id, long dflt) {
    const long v = effi_option(id);
    return v == EFFI_OPT_UNSET ? dflt : v;
}
static long effi_mr4_min() { return effi_opt_or(EFFI_OPT_MR4_MIN, 400); }
static long effi_mr4_nt2_max() { return effi_opt_or(EFFI_OPT_MR4_NT2_MAX, 1024); }
static long effi_mr2_min() { return effi_opt_or(EFFI_OPT_MR2_MIN, 400); }
// Rows per wave (MR): 4 rows amortise the B fragments best, but the grid must still cover the 256 CUs (>= ~400 workgroups),
// and with two N-tiles the 4-row variant drops to 2 workgroups per CU where the 2-row one keeps 4: on large maps the latter
// wins.  (Persistent workgroups with cross-tile prefetch were built and measured twice: no gain, more registers.)
template <int NT, int EPI, bool ZB = false, bool SR = false>
static int launch_bf16x3(const Conv2dArgs& a, hipStream_t st) {
    const long cols = effi_cdiv(a.w, 16);
    const long planes = ZB ? a.zcount : 1;
    const long t4 = cols * effi_cdiv(a.h, 16) * planes, t2 = cols * effi_cdiv(a.h, 8) * planes;
    int mr;
    if (t4 >= effi_mr4_min() && !(NT == 2 && t4 >= effi_mr4_nt2_max())) mr = 4;
    else if (t2 >= effi_mr2_min()) mr = 2;
    else mr = 1;
    const long force = effi_opt_or(EFFI_OPT_FORCE_MR, 0);
    if (force) mr = (int)force;
    // wide tiles (4 rows x 64 columns) pay off on the large maps only (measured with an HBM-cold working set, 592x800:
    // 16->16 23.8 -> 22.6 us, 32->12 33.3 -> 30.7, 32->16 GRU update 41.9 -> 38.6; 296x400: 38 -> 49 us for 64->64)
    const long wide_env = effi_opt_or(EFFI_OPT_WIDE_TILES, -1);
    const bool wide = wide_env >= 0 ? wide_env != 0 : (mr == 4 && a.w >= 512 && !ZB);
    if (wide) {
        const int tiles_x = effi_cdiv(a.w, 16 * mr), ntiles = tiles_x * effi_cdiv(a.h, 4);
        const dim3 grid(ntiles, (unsigned)planes);
        if (mr == 4) hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 4, EPI, ZB, true, SR>), grid, dim3(256), 0, st, a, tiles_x, ntiles);
        else if (mr == 2) hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 2, EPI, ZB, true, SR>), grid, dim3(256), 0, st, a, tiles_x, ntiles);
        else hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 1, EPI, ZB, true, SR>), grid, dim3(256), 0, st, a, tiles_x, ntiles);
        return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
    }
    if constexpr (SR) {
        // eight waves per workgroup: the tile of 4-row waves' workgroup at half the rows per wave, twice the pixels behind one copy of
        // the weight fragments.  Rule (option sr_waves unset): one row per wave AND six N-tiles -- the stage-1 z | r layer, whose 61 KB
        // of weight fragments per chunk and 64-pixel workgroup are what its launch moves (84.7-87.0 -> 78.4-79.0 us per view,
        // profiles/r04_ai_sr_waves_ab2.txt); every other layer is equal or slower with eight waves (stage-3 z | r: +12 us).
        // sr_waves = 8: wherever the rule picks 1 or 2 rows per wave (A/B); sr_waves = 4: never.
        const long srw = effi_option(EFFI_OPT_SR_WAVES);
        if ((srw == 8 && mr <= 2) || (srw == EFFI_OPT_UNSET && mr == 1 && NT >= 6)) {
            const int tiles_x = (int)cols, ntiles = tiles_x * effi_cdiv(a.h, 8 * mr);
            const dim3 grid(ntiles, 1);
            if (mr == 2) hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 2, EPI, false, false, true, 8>), grid, dim3(512), 0, st, a, tiles_x, ntiles);
            else hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 1, EPI, false, false, true, 8>), grid, dim3(512), 0, st, a, tiles_x, ntiles);
            return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
        }
    }
    const int tiles_x = (int)cols, ntiles = tiles_x * effi_cdiv(a.h, 4 * mr);
    const dim3 grid(ntiles, (unsigned)planes);
    if (mr == 4) hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 4, EPI, ZB, false, SR>), grid, dim3(256), 0, st, a, tiles_x, ntiles);
    else if (mr == 2) hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 2, EPI, ZB, false, SR>), grid, dim3(256), 0, st, a, tiles_x, ntiles);
    else hipLaunchKernelGGL((conv2d_k3_bf16x3_kernel<NT, 1, EPI, ZB, false, SR>), grid, dim3(256), 0, st, a, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

template <int EPI, bool SR = false>
static int dispatch_bf16x3(const Conv2dArgs& a, int nt, hipStream_t st) {
    switch (nt) {
        case 1: return launch_bf16x3<1, EPI, false, SR>(a, st);
        case 2: return launch_bf16x3<2, EPI, false, SR>(a, st);
        case 3: return launch_bf16x3<3, EPI, false, SR>(a, st);
        case 4: return launch_bf16x3<4, EPI, false, SR>(a, st);
        case 6: return launch_bf16x3<6, EPI, false, SR>(a, st);
        default: return EFFI_ERR_UNSUPPORTED;
    }
}

// Pair launch (see conv2d_k3_bf16x3_pair_kernel): tile shape chosen as launch_bf16x3 does for two planes.
template <int NT, bool SR = false>
static int launch_bf16x3_pair(const Conv2dArgs& a0, const Conv2dArgs& a1, hipStream_t st) {
    const long cols = effi_cdiv(a0.w, 16);
    const long t4 = cols * effi_cdiv(a0.h, 16) * 2, t2 = cols * effi_cdiv(a0.h, 8) * 2;
    int mr;
    if (t4 >= effi_mr4_min() && !(NT == 2 && t4 >= effi_mr4_nt2_max())) mr = 4;
    else if (t2 >= effi_mr2_min()) mr = 2;
    else mr = 1;
    const long force = effi_opt_or(EFFI_OPT_FORCE_MR, 0);
    if (force) mr = (int)force;
    if (mr == 4 && a0.w >= 512) {
        const int tiles_x = effi_cdiv(a0.w, 64), ntiles = tiles_x * effi_cdiv(a0.h, 4);
        hipLaunchKernelGGL((conv2d_k3_bf16x3_pair_kernel<NT, 4, true, SR>), dim3(ntiles, 2), dim3(256), 0, st, a0, a1, tiles_x, ntiles);
        return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
    }
    if constexpr (SR) {
        if (effi_option(EFFI_OPT_SR_WAVES) == 8 && mr <= 2) {
            const int tiles_x = (int)cols, ntiles = tiles_x * effi_cdiv(a0.h, 8 * mr);
            const dim3 grid(ntiles, 2);
            if (mr == 2) hipLaunchKernelGGL((conv2d_k3_bf16x3_pair_kernel<NT, 2, false, true, 8>), grid, dim3(512), 0, st, a0, a1, tiles_x, ntiles);
            else hipLaunchKernelGGL((conv2d_k3_bf16x3_pair_kernel<NT, 1, false, true, 8>), grid, dim3(512), 0, st, a0, a1, tiles_x, ntiles);
            return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
        }
    }
    const int tiles_x = (int)cols, ntiles = tiles_x * effi_cdiv(a0.h, 4 * mr);
    const dim3 grid(ntiles, 2);
    if (mr == 4) hipLaunchKernelGGL((conv2d_k3_bf16x3_pair_kernel<NT, 4, false, SR>), grid, dim3(256), 0, st, a0, a1, tiles_x, ntiles);
    else if (mr == 2) hipLaunchKernelGGL((conv2d_k3_bf16x3_pair_kernel<NT, 2, false, SR>), grid, dim3(256), 0, st, a0, a1, tiles_x, ntiles);
    else hipLaunchKernelGGL((conv2d_k3_bf16x3_pair_kernel<NT, 1, false, SR>), grid, dim3(256), 0, st, a0, a1, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

// Generated-input pair (conv2d_k3_bf16x3_encgen_pair_kernel): the MR rule of launch_bf16x3_pair, 4 waves.
template <int NT>
static int launch_bf16x3_encgen_pair(const Conv2dArgs& a0, const Conv2dArgs& a1, const EncGenArgs& g, hipStream_t st) {
    const long cols = effi_cdiv(a0.w, 16);
    const long t4 = cols * effi_cdiv(a0.h, 16) * 2, t2 = cols * effi_cdiv(a0.h, 8) * 2;
    int mr;
    if (t4 >= effi_mr4_min() && !(NT == 2 && t4 >= effi_mr4_nt2_max())) mr = 4;
    else if (t2 >= effi_mr2_min()) mr = 2;
    else mr = 1;
    const long force = effi_opt_or(EFFI_OPT_FORCE_MR, 0);
    if (force) mr = (int)force;
    // always 16-column tiles: the generated image pays per STAGED pixel (24 volume taps each), and a 4 x 64 tile stages 1.55 x its
    // pixels against 1.27 x for 16 x 16 (592x800, hd 16: 55 us per launch with wide tiles, 47 us without; the loaded form gains 5 %
    // from wide tiles)
    // "4 rows per wave" is 3 here: 14 x 18 = 252 staged pixels are ONE lookup per thread; 18 x 18 = 324 make 68 threads do a second
    // one while 188 wait (option enc_gen_mr3 = 0: 4 rows)
    if (mr == 4 && effi_opt_or(EFFI_OPT_ENC_GEN_MR3, 1) != 0) mr = 3;
    const int tiles_x = (int)cols, ntiles = tiles_x * effi_cdiv(a0.h, 4 * mr);
    const dim3 grid(ntiles, 2);
    if (mr == 4) hipLaunchKernelGGL((conv2d_k3_bf16x3_encgen_pair_kernel<NT, 4, false>), grid, dim3(256), 0, st, a0, a1, g, tiles_x, ntiles);
    else if (mr == 3) hipLaunchKernelGGL((conv2d_k3_bf16x3_encgen_pair_kernel<NT, 3, false>), grid, dim3(256), 0, st, a0, a1, g, tiles_x, ntiles);
    else if (mr == 2) hipLaunchKernelGGL((conv2d_k3_bf16x3_encgen_pair_kernel<NT, 2, false>), grid, dim3(256), 0, st, a0, a1, g, tiles_x, ntiles);
    else hipLaunchKernelGGL((conv2d_k3_bf16x3_encgen_pair_kernel<NT, 1, false>), grid, dim3(256), 0, st, a0, a1, g, tiles_x, ntiles);
    return hipPeekAtLastError() == hipSuccess ? EFFI_OK : EFFI_ERR_LAUNCH;
}

}  // namespace
