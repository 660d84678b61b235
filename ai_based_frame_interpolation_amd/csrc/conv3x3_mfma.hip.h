// conv3x3_mfma.hip.h -- fused 3x3 conv (pad 1, no bias) + per-channel scale/shift (eval BatchNorm)
// + ReLU as an implicit GEMM on the CDNA4 matrix cores, for gfx950 only.
//
// Replaces the aten conv2d + batch_norm + relu triple behind DoubleConv
// (/root/reference/model/unet.py:11-18) and, fused around it, the other ops of the UNet forward:
//   * MaxPool2d(2) of Down (unet.py:28)           -> EPI_POOL: the producer also writes the pooled copy
//   * Upsample + F.pad + torch.cat of Up (:46-54) -> SRC_CONCAT_UP: interpolated inside the gather
//   * OutConv 1x1 + bias (:60)                    -> EPI_HEAD / EPI_HEAD3: reduced in the epilogue
//   * the 2->64 stem conv (:72), bf16 gray path   -> SRC_STEM: evaluated inside the gather
// so none of the pooled-input re-reads, upsampled, concatenated, stem-output or last-activation
// tensors is ever materialised in HBM.  The same kernel runs precision "bf16x2" (gather mode SRC_DIRECT_X2: two bf16
// pieces per activation and weight, three MFMAs per product - the reference's fp32 tolerance on the bf16 pipe).
//
// Mapping onto the hardware (one workgroup = 4 waves = 256 threads, 2 workgroups per CU):
//   * GEMM view: D[cout][pixel] += W[cout][k] * X[k][pixel], k = (channel plane, kx, ky).
//     MFMA A operand = weights (16 couts x 8k per lane-row), B operand = pixels.  With that
//     orientation a lane of the 16x16 accumulator holds consecutive couts of one pixel, so the
//     epilogue stores 16 contiguous bytes per lane into the blocked activation layout
//     [B][C/PL][H][W][PL] (see blk_off below).
//   * A "plane" is 64 bytes of channels per pixel (32 bf16 or 16 fp32).  The input tile
//     (TH+2)x(TW+2) pixels of one plane is staged in LDS ONCE and reused by all 9 taps -- the
//     shifted windows are just different LDS addresses (base + immediate offset).
//   * A step is one (plane, kx): its three taps ky = 0..2 read in-tile rows r, r+1, r+2 of the same
//     columns, so a wave fetches each (row, 16-pixel) fragment ONCE per step and uses it for up to
//     three taps (rolling window of ROWS_W + 2 rows): 24 instead of 36 ds_read_b128 per 96 MFMAs.
//   * Weights for (plane, kx, ky=0..2) are streamed per step: 3*BN rows of 64 B, by LDS-DMA
//     (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPRs) into a 2-deep ring, one
//     step ahead of the MFMAs, so the L2 latency of the weight stream hides under the previous
//     step's 96 MFMAs per wave.  The next plane's input tile is gathered at the plane boundary
//     behind one extra barrier (LDS-DMA for stored planes; LDS-staged bilinear interpolation or
//     the split-bf16 stem conv for computed ones); the co-resident second workgroup of the CU
//     covers that gap.
//   * Both LDS images are [row][64 B] with the 16-B chunk index XOR-ed by ((row>>2)&1)<<1, which
//     makes every ds_read_b128 of 16 consecutive rows x 4 chunks conflict-free for any row
//     alignment (the 16-lane groups of ds_read_b128 are listed in MI355X_MICROARCH.md, LDS).
//     The in-tile row pitch is padded to a multiple of 8 pixels so the swizzle term of a lane
//     does not depend on the tile row.
//   * Each wave owns 64 couts x 128 pixels = 4 x 8 accumulator tiles (128 VGPRs).
//   * bf16: v_mfma_f32_16x16x32_bf16 (one per A/B chunk pair); fp32: 4 x v_mfma_f32_16x16x4_f32
//     per chunk pair (exact fp32 FMA chain) -- same kernel body, same LDS images.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fiunet {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

enum SrcMode { SRC_DIRECT = 0, SRC_POOL = 1 /* host-side tag only */, SRC_CONCAT_UP = 2,
               SRC_STEM = 3 /* input = stem conv of the raw frame pair, computed in the gather */,
               // precision "bf16x2" (the fp32 contract on the bf16 pipe): direct sources whose activations are TWO bf16
               // pieces, x = hi + lo (16 significant bits), stored [hi planes | lo planes] (4 B per element, fp32's
               // traffic), weights [wh planes | wl planes]; a product is wh*xh + wl*xh + wh*xl with fp32 accumulation
               // (the dropped wl*xl term is 2^-16 relative).  The K loop runs three "virtual planes" per real plane of 32
               // channels - (xh, wh), (xh, wl), (xl, wh) - and the second one reuses the in-tile of the first: every piece
               // is stored once and gathered once.  Every epilogue of this mode writes the two pieces of its output.
               SRC_DIRECT_X2 = 4,
               // bf16x2 + fused stem (gray): the stem conv is evaluated inside the gather as in SRC_STEM, once for the hi
               // piece and once more for the lo piece of each of its two 32-channel planes (same split-bf16 MFMAs, ~2^-16
               // relative: this precision's own accuracy class); the 64-channel two-piece stem output never goes to HBM
               SRC_STEM_X2 = 5 };
constexpr bool src_is_stem(int mode) { return mode == SRC_STEM || mode == SRC_STEM_X2; }
constexpr bool src_is_x2(int mode) { return mode == SRC_DIRECT_X2 || mode == SRC_STEM_X2; }

// Activation layout in HBM: plane-major blocked channels-last, [B][C/PL][H][W][PL] with one
// 64-byte "plane" record per pixel (PL = 32 bf16 / 16 fp32 channels).  A tile row of one plane is
// one contiguous run of (TW+2)*64 bytes, so the LDS-DMA gather of a plane touches every 128-B line
// once and in full (with plain NHWC the two 64-B halves of a 64-channel pixel were fetched by two
// gathers microseconds apart and the line was re-read from HBM when the store stream had evicted
// it).  Byte offset of (plane, y, x) inside one image:
__device__ __forceinline__ size_t blk_off(int plane, int y, int x, int H, int W)
{
    return (((size_t)plane * H + y) * W + x) * 64;
}

struct ConvArgs {
    const void* src0;    // [B][C0/PL][H][W][PL]
    const void* src1;    // CONCAT_UP: low-res [B][C1/PL][lowH][lowW][PL], bilinearly upsampled on the fly
    const void* wgt;     // [Cin/PL][9][Cout][PL]  (plane-major, then tap, cout, channel-in-plane)
    const float* scale;  // [Cout]  gamma / sqrt(var + eps)
    const float* shift;  // [Cout]  beta - mean * scale
    void* dst;           // [B][Cout/PL][H][W][PL] or nullptr (fused head only)
    void* pool_dst;      // EPI_POOL: [B][Cout/PL][H/2][W/2][PL], MaxPool2d(2) of dst
    int B, H, W;         // conv input == output spatial size
    int C0, C1, Cout;    // (SRC_DIRECT_X2: REAL channel counts; every tensor then holds its hi planes followed by its lo planes)
    int lowH, lowW;      // CONCAT_UP: spatial size of src1
    int padT, padL;      // CONCAT_UP: F.pad top/left (unet.py:52-53)
    float sy, sx;        // CONCAT_UP: (low-1)/(2*low-1), align_corners=True scale
    // CONCAT_UP on a horizontal band of a taller image (fiunet_forward_strip); un-tiled: 0, 0, lowH
    int upOffY;          //   global row of this tensor's row 0
    int lowOffY;         //   global row of src1's row 0
    int lowHg;           //   rows of the WHOLE image's low-res tensor (padT, sy are global too)
    int tilesX, tilesY, nct;
    int relu;
    const void* zero_page; // >= 64 zero bytes: padding source of the tile-pair kernel's gather (this kernel zeroes
                           // the padding slots of its in-tile once and masks those lanes out of the DMAs)
    // SRC_STEM (bf16, gray): the 2->64 stem conv + BN + ReLU (unet.py:72) is evaluated inside the
    // in-tile gather of the NEXT conv, so its 64-channel output never goes to HBM.
    const float* f1;          // frame1 [B][1][H][W] fp32
    const float* f2;          // frame2
    const void* stem_w;       // [2 (hi, lo)][64 packed rows][32 k] bf16 of w * bn_scale; k = lane group*8 + dx*2 + frame
                              // with lane groups 0, 1, 2 <-> dy = 0, 2, 1; zero for dx = 3; k = 24 (lane group 3) holds
                              // the BatchNorm shift (its operand is 1.0); packed row R <-> cout bf16_row_to_cout(R)
    float dither;             // bf16 stem only: amplitude of the ordered input dither (stem_dither), 0 = off
    // EPI_SPLITK: the K loop (planes) is cut into `ksplit` slices handled by different workgroups; slice s stores its
    // raw fp32 partial sums to kslab[s][tile][wave][fragment][lane]; splitk_finalize_tile_kernel adds the slices in
    // index order (deterministic) on top of the BatchNorm shift and runs the epilogue.
    int ksplit;
    int pair;            // host side only: 8-wave tile-pair kernel where it applies (FIUNET_OPT_PAIR_TILES)
    int force_tile;      // host side only (diagnostic, fiunet_debug_force_cfg): 0 = choose, 1 = the big tile, 2 = the small one, 3 = conv3x3_kwave_kernel
    int force_ksplit;    // host side only (diagnostic): 0 = choose, k >= 1 = cut the K loop k ways
    int concat_origin;   // host side only: 1 = an fp32 concat conv run through upcat_kernel (ablation / read-back path) whose fused
                         // counterpart keeps the in-gather form: it takes that form's launch configuration (same K cut, never
                         // conv3x3_kwave_kernel), so that the two paths stay bit-identical per stage
    float* kslab;
    unsigned long long* stamp;  // diagnostic builds (-DFIUNET_STAMP / -DFIUNET_CLOCK) only: one 128-B record (16 cycle sums) per wave
    unsigned stamp_cap;         //   records the buffer holds (waves beyond it do not write)
    const float* head_w; // fused 1x1 head: [head_nc][64]
    const float* head_b; // [head_nc]
    float* head_out;     // fp32 NCHW [B][head_nc][H][W]
    // fiunet_forward_u8: the reference's pre/post-processing (model/inference.py:31-35, :54-61) applied where the
    // frames are read and where the output is written, so no fp32 frame buffer exists
    const uint8_t* u1;        // SRC_STEM: uint8 frames [B][1][H][W] instead of f1 / f2 (nullptr: fp32 frames)
    const uint8_t* u2;
    uint8_t* head_out_u8;     // fused head: uint8 NCHW output instead of head_out (nullptr: fp32 logits)
    int head_nc;
    size_t head_img_stride;   // elements from one image of head_out / head_out_u8 to the next (contiguous: head_nc * H * W;
                              // fiunet_forward_u8_strided: the video loop's interleaved destination, every second frame)
};

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PL = 16;  // channels per 64-B plane
    static constexpr int NE = 4;   // elements per 16-B chunk
};
template <> struct Elem<__bf16> {
    static constexpr int PL = 32;
    static constexpr int NE = 8;
};

// ---- 16-byte chunk helpers -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void chunk_unpack(const uint4& c, float* f);
template <> __device__ __forceinline__ void chunk_unpack<float>(const uint4& c, float* f)
{
    f[0] = __uint_as_float(c.x); f[1] = __uint_as_float(c.y);
    f[2] = __uint_as_float(c.z); f[3] = __uint_as_float(c.w);
}
template <> __device__ __forceinline__ void chunk_unpack<__bf16>(const uint4& c, float* f)
{
    f[0] = __uint_as_float(c.x << 16); f[1] = __uint_as_float(c.x & 0xffff0000u);
    f[2] = __uint_as_float(c.y << 16); f[3] = __uint_as_float(c.y & 0xffff0000u);
    f[4] = __uint_as_float(c.z << 16); f[5] = __uint_as_float(c.z & 0xffff0000u);
    f[6] = __uint_as_float(c.w << 16); f[7] = __uint_as_float(c.w & 0xffff0000u);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)lo, (__bf16)hi};  // v_cvt_pk_bf16_f32, round-to-nearest-even
    return __builtin_bit_cast(unsigned, v);
}
// The same conversion spelled as a vector convert: always ONE v_cvt_pk_bf16_f32 (with the SLP
// vectoriser off, the two scalar casts above sometimes come out as two converts and a v_perm).
// Not used everywhere: in the fused-head kernels the shorter form lets hipcc hoist all the packs
// ahead of the head's fp32 math, and spill.
__device__ __forceinline__ unsigned pack_bf16x2_pk(float lo, float hi)
{
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
template <typename T> __device__ __forceinline__ uint4 chunk_pack(const float* f);
template <> __device__ __forceinline__ uint4 chunk_pack<float>(const float* f)
{
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]),
                      __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 chunk_pack<__bf16>(const float* f)
{
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]),
                      pack_bf16x2(f[6], f[7]));
}

typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b)
{
    const s16x2 x = __builtin_bit_cast(s16x2, a), y = __builtin_bit_cast(s16x2, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));  // v_pk_max_i16
}

template <typename T>
__device__ __forceinline__ uint4 chunk_max4(const uint4& a, const uint4& b, const uint4& c,
                                            const uint4& d);

// bf16 max-pool of NON-NEGATIVE values (every pooled tensor in this network is a ReLU output):
// for x >= 0 the bf16 order equals the signed 16-bit integer order of the bit patterns, and a
// stray -0.0 (0x8000 = INT16_MIN) loses against everything, so a packed integer max is exact.
template <>
__device__ __forceinline__ uint4 chunk_max4<__bf16>(const uint4& a, const uint4& b, const uint4& c,
                                                    const uint4& d)
{
    return make_uint4(pk_max_i16(pk_max_i16(a.x, b.x), pk_max_i16(c.x, d.x)),
                      pk_max_i16(pk_max_i16(a.y, b.y), pk_max_i16(c.y, d.y)),
                      pk_max_i16(pk_max_i16(a.z, b.z), pk_max_i16(c.z, d.z)),
                      pk_max_i16(pk_max_i16(a.w, b.w), pk_max_i16(c.w, d.w)));
}

template <>
__device__ __forceinline__ uint4 chunk_max4<float>(const uint4& a, const uint4& b, const uint4& c,
                                                   const uint4& d)
{
    using T = float;
    constexpr int NE = Elem<T>::NE;
    float fa[NE], fb[NE], fc[NE], fd[NE], r[NE];
    chunk_unpack<T>(a, fa); chunk_unpack<T>(b, fb); chunk_unpack<T>(c, fc); chunk_unpack<T>(d, fd);
#pragma unroll
    for (int i = 0; i < NE; ++i) r[i] = fmaxf(fmaxf(fa[i], fb[i]), fmaxf(fc[i], fd[i]));
    return chunk_pack<T>(r);  // exact for bf16: the max is one of the (bf16) inputs
}

// hy*(hx*a + lx*b) + ly*(hx*c + lx*d): the association aten's upsample_bilinear2d uses.
template <typename T>
__device__ __forceinline__ uint4 chunk_bilerp(const uint4& a, const uint4& b, const uint4& c,
                                              const uint4& d, float hx, float lx, float hy,
                                              float ly)
{
    constexpr int NE = Elem<T>::NE;
    float fa[NE], fb[NE], fc[NE], fd[NE], r[NE];
    chunk_unpack<T>(a, fa); chunk_unpack<T>(b, fb); chunk_unpack<T>(c, fc); chunk_unpack<T>(d, fd);
    // explicit fma pattern: the same bits in every kernel this is inlined into (hipcc's default
    // fp-contract=fast would otherwise pick a different contraction per context)
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const float top = fmaf(lx, fb[i], __fmul_rn(hx, fa[i]));
        const float bot = fmaf(lx, fd[i], __fmul_rn(hx, fc[i]));
        r[i] = fmaf(ly, bot, __fmul_rn(hy, top));
    }
    return chunk_pack<T>(r);
}

// The two halves of chunk_bilerp, for callers that walk down a column and reuse the horizontal
// result of a low-res row for every output row that touches it.  Same operations on the same
// operands in the same order as chunk_bilerp, hence the same bits.
template <typename T>
__device__ __forceinline__ void chunk_hlerp(const uint4& a, const uint4& b, float hx, float lx, float* h)
{
    [[maybe_unused]] constexpr int NE = Elem<T>::NE;
#ifdef FIUNET_DIAG_NO_HLERP
    // timing diagnostic (results are garbage): what the consumer's gather would cost if the HORIZONTAL lerp had been
    // done by the producer (review item 6): one staged chunk per low-res row, unpacked, no arithmetic
    (void)b; (void)hx; (void)lx;
    chunk_unpack<T>(a, h);
#else
    float fa[NE], fb[NE];
    chunk_unpack<T>(a, fa); chunk_unpack<T>(b, fb);
#pragma unroll
    for (int i = 0; i < NE; ++i) h[i] = fmaf(lx, fb[i], __fmul_rn(hx, fa[i]));
#endif
}
template <typename T>
__device__ __forceinline__ uint4 chunk_vlerp(const float* top, const float* bot, float hy, float ly)
{
    constexpr int NE = Elem<T>::NE;
    float r[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) r[i] = fmaf(ly, bot[i], __fmul_rn(hy, top[i]));
    return chunk_pack<T>(r);
}


__device__ __forceinline__ uint4 ldg16(const char* p) { return *reinterpret_cast<const uint4*>(p); }

// model/inference.py:31-35: image.astype(float32) / 255.0 ; 2.0 * image - 1.0   (numpy fp32 arithmetic)
__device__ __forceinline__ float preprocess_u8_value(unsigned char u)
{
    // u / 255 correctly rounded without the division sequence: q0 = u * RN(1/255), one Newton correction
    // with the exact remainder (fma); equal to the IEEE quotient for all 256 codes (asserted bit for bit
    // against numpy by tests/test_gpu_parity.py over every code)
    const float x = (float)u, r = 1.0f / 255.0f;
    const float q0 = __fmul_rn(x, r);
    const float q = fmaf(fmaf(-q0, 255.0f, x), r, q0);
    return __fsub_rn(__fmul_rn(2.0f, q), 1.0f);
}
// model/inference.py:54-61: (x + 1) / 2 ; clamp(0, 1) ; (x * 255).astype(uint8) -- truncation
__device__ __forceinline__ unsigned char postprocess_u8_value(float x)
{
    float v = __fdiv_rn(__fadd_rn(x, 1.0f), 2.0f);
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return (unsigned char)(int)__fmul_rn(v, 255.0f);
}

// Ordered dither of the bf16 path's INPUT (bf16 stem only; the fp32 path never sees it).
// Why: a bf16 activation has 8 significant bits, the frames have 8 too (x = 2*u8/255 - 1), and a network
// that interpolates carries the frames themselves through the full-resolution skip (x1 -> up4).  Rounded
// to nearest, the error of such a carried value is a fixed sawtooth of the pixel's intensity, i.e. a
// small intensity-dependent gain/offset error that is the same in every smooth region of the image: it
// correlates with the content and moves PSNR-vs-truth by 0.03-0.05 dB on an interpolating checkpoint,
// ten times what its energy alone would (tools/bf16_emulate.py isolates it: rounding the stem output /
// x1 / up4.0 accounts for nearly all of the bf16 path's PSNR difference; all other layers together for
// < 0.01 dB).  The classic remedy is dither: frame 1 gets +d(y, x), frame 2 gets -d(y, x) before the
// stem conv, d = the 8x8 Bayer matrix scaled to a quarter of an 8-bit input step peak-to-peak/2
// (amplitude 2^-8 = +-2^-9, below the frames' own quantisation noise).  The rounding error of a carried
// value then averages to zero over every 8x8 block whatever the intensity, and in the blend
// 0.5*(f1 + f2) the two dithers cancel exactly.  Measured (emulation, 5 checkpoints x 5 scenes):
// |PSNR_bf16 - PSNR_fp32| 0.017-0.039 dB -> <= 0.0125 dB, rel-L2 of the output error slightly lower.
// Deterministic and position-based: (y & 7, x & 7) of the GLOBAL pixel (strip origins are multiples of
// 16 rows), so tiled == un-tiled and batch invariance are unaffected.
__device__ __forceinline__ float stem_dither(int y, int x)
{
    // Bayer index matrix M8: bits of (x ^ y) and y interleaved, bit 0 most significant
    const unsigned a = (unsigned)(x ^ y) & 7u, b = (unsigned)y & 7u;
    const unsigned m = ((a & 1u) << 5) | ((b & 1u) << 4) | ((a & 2u) << 2) | ((b & 2u) << 1) | ((a & 4u) >> 1) | ((b & 4u) >> 2);
    return ((float)m + 0.5f) * (1.0f / 64.0f) - 0.5f;  // (-0.5, 0.5), 64 levels, zero mean over a period
}

// Bilinear x2 (align_corners=True) source coordinates and weights of one upsampled+padded pixel
// (unet.py:40,49-53), shared by every kernel that upsamples so they agree bit for bit.
struct UpAxis {
    int i0, i1;   // low-res source indices (local to the tensor at hand)
    float h, l;   // weights of i0, i1
    bool ok;      // inside the upsampled extent (false: F.pad zero)
};
// One axis of the mapping.  c = conv-input coordinate (already clamped to the local tensor).
// Strip-tiled forwards (fiunet_forward_strip) run on a horizontal band of a taller image: the
// mapping is then evaluated in GLOBAL coordinates (c + off; low-res extent lowNg; pad and scale of
// the whole image) and only the resulting source indices are moved back into the band (- lowOff,
// clamped to its lowN rows), so a band computes exactly what the whole image would.  Un-tiled:
// off = lowOff = 0 and lowNg = lowN.
__device__ __forceinline__ UpAxis up_axis(int c, int off, int pad, int lowNg, float scale, int lowOff,
                                          int lowN)
{
    UpAxis u;
    int cu = c + off - pad;
    u.ok = (cu >= 0) & (cu < 2 * lowNg);
    cu = min(max(cu, 0), 2 * lowNg - 1);
    // f is the ROUNDED product, as in aten (the library is built with -ffp-contract=off)
    const float f = scale * (float)cu;
    const int g0 = (int)f;
    const int g1 = g0 < lowNg - 1 ? g0 + 1 : g0;
    u.l = f - (float)g0;
    u.h = 1.0f - u.l;
    u.i0 = min(max(g0 - lowOff, 0), lowN - 1);
    u.i1 = min(max(g1 - lowOff, 0), lowN - 1);
    return u;
}
__device__ __forceinline__ UpAxis up_axis_y(const ConvArgs& a, int y)
{
    return up_axis(y, a.upOffY, a.padT, a.lowHg, a.sy, a.lowOffY, a.lowH);
}
__device__ __forceinline__ UpAxis up_axis_x(const ConvArgs& a, int x)
{
    return up_axis(x, 0, a.padL, a.lowW, a.sx, 0, a.lowW);
}
struct UpCoord {
    int y0, y1, x0, x1;
    float hy, ly, hx, lx;
    bool ok;
};
__device__ __forceinline__ UpCoord up_coord(const ConvArgs& a, int y, int x)
{
    const UpAxis v = up_axis_y(a, y), h = up_axis_x(a, x);
    UpCoord u;
    u.y0 = v.i0; u.y1 = v.i1; u.hy = v.h; u.ly = v.l;
    u.x0 = h.i0; u.x1 = h.i1; u.hx = h.h; u.lx = h.l;
    u.ok = v.ok & h.ok;
    return u;
}

// One 16-byte chunk (plane `plane`, chunk `ch`) of conv-input pixel (b, y, x) straight from
// global memory; zero outside the image (conv padding) and outside the upsampled extent (F.pad).
// Used by the standalone (ablation) upsample+concat kernel; the fused conv computes the same
// values from an LDS-staged low-res tile.
template <typename T, int MODE>
__device__ __forceinline__ uint4 gather_chunk(const ConvArgs& a, int b, int y, int x, int plane,
                                              int ch)
{
    constexpr int PL = Elem<T>::PL;
    bool ok = (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
    y = min(max(y, 0), a.H - 1);
    x = min(max(x, 0), a.W - 1);
    const int p0 = a.C0 / PL;
    uint4 v;
    if (MODE == SRC_DIRECT || plane < p0) {
        const char* p = (const char*)a.src0 + (size_t)b * a.H * a.W * a.C0 * sizeof(T) +
                        blk_off(plane, y, x, a.H, a.W) + ch * 16;
        v = ldg16(p);
    } else {
        const UpCoord u = up_coord(a, y, x);
        ok = ok & u.ok;
        const char* base = (const char*)a.src1 + (size_t)b * a.lowH * a.lowW * a.C1 * sizeof(T) + ch * 16;
        const int q = plane - p0;
        const uint4 v00 = ldg16(base + blk_off(q, u.y0, u.x0, a.lowH, a.lowW));
        const uint4 v01 = ldg16(base + blk_off(q, u.y0, u.x1, a.lowH, a.lowW));
        const uint4 v10 = ldg16(base + blk_off(q, u.y1, u.x0, a.lowH, a.lowW));
        const uint4 v11 = ldg16(base + blk_off(q, u.y1, u.x1, a.lowH, a.lowW));
        v = chunk_bilerp<T>(v00, v01, v10, v11, u.hx, u.lx, u.hy, u.ly);
    }
    return ok ? v : make_uint4(0u, 0u, 0u, 0u);
}

// LDS-DMA: 16 bytes per lane, global -> LDS, destination = wave-uniform LDS byte address (in M0)
// + lane*16.  Issued from inline asm ON PURPOSE: hipcc treats the builtin form as an LDS store
// that may alias every later ds_read and drains it with s_waitcnt vmcnt(0) before the first
// fragment read of the step, which would serialise the weight stream with the MFMAs.  The asm
// form is invisible to its wait bookkeeping, so completion is waited for by hand
// (lds_dma_wait_all) before the barrier that publishes the data.  M0 is saved/restored in
// the same statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_wave_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_wave_base)
                 : "memory");
}
// The same with a wave-uniform 64-bit base in SGPRs and a 32-bit per-lane byte offset: one VGPR per
// address instead of two.  Lanes switched off by EXEC neither load nor write their 16 LDS bytes.
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, unsigned lds_wave_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const char* p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }

template <typename T>
__device__ __forceinline__ void mma_chunk(f32x4& acc, const uint4& wa, const uint4& xb);
template <>
__device__ __forceinline__ void mma_chunk<__bf16>(f32x4& acc, const uint4& wa, const uint4& xb)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa),
                                                  __builtin_bit_cast(bf16x8, xb), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_chunk<float>(f32x4& acc, const uint4& wa, const uint4& xb)
{
    // lane group g = lane>>4 holds channels 4g..4g+3 of the plane in both operands; MFMA i
    // consumes element i of every group, i.e. k-slot g <-> channel 4g+i on both sides.
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.x), __uint_as_float(xb.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.y), __uint_as_float(xb.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.z), __uint_as_float(xb.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa.w), __uint_as_float(xb.w), acc, 0, 0, 0);
}

enum Epilogue { EPI_PLAIN = 0, EPI_HEAD = 1 /* 1x1 head, 1 class */, EPI_POOL = 2, EPI_HEAD3 = 3 /* 3 classes */,
                EPI_SPLITK = 4 /* raw fp32 partial sums of a K slice -> slab (small problems); splitk_finalize_tile_kernel reduces */ };
constexpr bool epi_is_splitk(int epi) { return epi == EPI_SPLITK; }

// value of the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS crossbar
__device__ __forceinline__ unsigned dpp_swap_pairs(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

// number of row segments for the column-walk upsample (see gather_plane_up): minimise
// passes over the 256 threads x (rows per segment + ~1 row of segment start-up)
constexpr int up_segments(int rows, int ncol)
{
    int best = 1, best_cost = 1 << 30;
    for (int n = 1; n <= 6; ++n) {
        const int passes = (n * ncol + 255) / 256, segr = (rows + n - 1) / n;
        const int cost = passes * (segr * 4 + 3);
        if (cost < best_cost) { best_cost = cost; best = n; }
    }
    return best;
}

// LDS map of one workgroup (<= 80 KiB so that two fit on a CU):
//   [ in-tile | W slot 0 | spare | W slot 1 ]
// The spare region sits between the two ring slots so that, whichever slot is idle at a plane
// boundary, idle slot + spare form one contiguous staging area for the low-res tile of an
// upsampled plane.
template <int BN, int TH, int TW, int MODE> struct ConvTile {
    static constexpr int TWP = ((TW + 2 + 7) / 8) * 8;  // in-tile row pitch, multiple of 8 pixels
    static constexpr int THP = TH + 2;
    static constexpr int IN_BYTES = THP * TWP * 64;
    static constexpr int W_BYTES = 3 * BN * 64;          // one (plane, kx) step: 3 taps
    // low-res tile of an upsampled plane: rows/cols that a (TH+2)x(TW+2) window can touch
    static constexpr int LRH = (TH + 1) / 2 + 3, LRW = (TW + 1) / 2 + 3, LRP = ((LRW + 3) / 4) * 4;
    static constexpr int LR_PIECES = (LRH * LRP + 15) / 16;
    // spare bytes between the ring slots: only what the staging area needs beyond one slot
    static constexpr int SPARE_BYTES =
        (MODE == SRC_CONCAT_UP && LR_PIECES * 1024 > W_BYTES) ? LR_PIECES * 1024 - W_BYTES : 0;
    static constexpr int W_STRIDE = W_BYTES + SPARE_BYTES;  // slot 1 = slot 0 + W_STRIDE
    // (Two ring slots.  A 4-deep ring for the small tiles - W(step+3) requested at the top of `step`, counted waits - was built in
    // round 6 and measured NULL on the small problems it was meant for (a lone workgroup's step is issue / LDS latency, not L2
    // latency: profiles/r06_cfg_sweep_b1_256_bf16_first_version.txt against r06_cfg_sweep_b1_256_bf16.txt), and its 24 KiB cost the
    // small tile its third workgroup per CU, which IS worth 5-17 % on the deep levels of one to four 1080p pairs.)
    // SRC_STEM: raw patch of both frames, (TH+4) x (TW+4) pixels, after the ring: bf16 dwords {frame1,
    // frame2}, the hi and the lo part of a patch row side by side, [row][hi | lo][PATCH_W].  The row pitch
    // of 2 * PATCH_W = 72 dwords makes rows py and py+2 - read together by lane groups 0 and 1 of a
    // ds_read2_b32 (lanes 0-31, 32 banks) - sit 144 = 16 (mod 32) banks apart: no conflicts (separate hi
    // and lo images of pitch 36 put rows py, py+1 only 4 banks apart: 2-way conflicts on 12 of 16 lanes).
    static constexpr int PATCH_W = TW + 4, PATCH_H = TH + 4, PATCH_PITCH = 2 * PATCH_W;
    static_assert(!src_is_stem(MODE) || (2 * PATCH_PITCH) % 32 == 16, "patch pitch: lane groups 0/1 must not share banks");
    static constexpr int PATCH_OFF = IN_BYTES + 2 * W_BYTES + SPARE_BYTES;
    // The operand of the bias k-slot - PATCH_W dwords {1.0, 0} followed by PATCH_W zero dwords (its lo
    // part), which all lanes of lane group 3 read at the same address (a broadcast) - lives in the
    // row-pitch filler of in-tile row 0 (pixels TW+2 .. TWP-1: never written or read in this mode), so
    // that the workgroup stays at 63 LDS allocation granules of 1280 B: at 64 (= 80 KiB, two workgroups
    // filling the CU's LDS exactly) this kernel ran 5 % slower.
    static constexpr int PATCH_TAIL_OFF = (TW + 2) * 64;                 // byte offset inside the in-tile
    static constexpr int PATCH_TAIL = 2 * PATCH_W * 4;
    static_assert(!src_is_stem(MODE) || PATCH_TAIL_OFF + PATCH_TAIL <= TWP * 64, "bias operand must fit the row-pitch filler");
    static constexpr int PATCH_BYTES = src_is_stem(MODE) ? PATCH_H * PATCH_PITCH * 4 : 0;
    // CONCAT_UP: the bilinear mapping of this tile, one 16-B entry per in-tile row and per in-tile
    // pixel column (same for every plane, so it is evaluated once per tile, not per plane)
    static constexpr int TAB_OFF = PATCH_OFF + ((PATCH_BYTES + 15) / 16) * 16;
    static constexpr int TAB_BYTES = MODE == SRC_CONCAT_UP ? ((THP + TW + 2) * 16 + 255) / 256 * 256 : 0;
    // SRC_STEM: plane 1's stem weights (hi and lo halves of two 16-cout tiles), parked here by
    // LDS-DMA at kernel start so the plane boundary does not wait on a global load
    static constexpr int STEMW_OFF = TAB_OFF + TAB_BYTES;
    static constexpr int STEMW_BYTES = src_is_stem(MODE) ? 4096 : 0;
    static constexpr int LDS_BYTES = STEMW_OFF + STEMW_BYTES;
    static_assert(LDS_BYTES <= 80 * 1024, "two workgroups must fit in the CU's 160 KiB of LDS");
    static_assert(!src_is_stem(MODE) || LDS_BYTES <= 63 * 1280, "fused-stem kernel: stay below 64 LDS granules");
};

// 16-pixel fragments per wave: 8 (wave tile 64 couts x 128 pixels, 128 accumulator registers, two
// workgroups per CU) or 4 (64 x 64, 64 accumulator registers, three workgroups per CU: for the
// short-K full-resolution layers, whose prologue/epilogue share is large, a third resident wave per
// SIMD keeps the MFMA pipe fed while the other two gather or store).
constexpr int conv_wave_frags(int BN, int TH, int TW) { return TH * TW / 16 / (4 / (BN / 64)); }
// (a 64 x 64 wave tile whose workgroup needs more than a third of the CU's LDS - the fused-stem gather - stays at two)
constexpr int conv_occupancy(int BN, int TH, int TW, int lds_bytes = 0)
{
    return conv_wave_frags(BN, TH, TW) == 4 && lds_bytes * 3 <= 160 * 1024 ? 3 : 2;
}

// ---- accumulator set-up and epilogue shared by the conv kernels ---------------------------------
// Eval-mode BatchNorm is folded on both sides of the K loop: its scale into the packed weights
// (fiunet_load_weights) and its shift into the INITIAL value of the accumulators, so the epilogue is
// just y = relu(acc).  Which couts a lane holds: accumulator tile m, register j of lane group lc is
// MFMA row lc*4+j of that tile.  fp32: row r of tile m <-> cout m*16+r, so a lane's 4 registers are
// one 16-B quarter of the tile's 64-B plane record.  bf16: the packed weight rows are permuted on the
// host (fiunet.hip, `bf16_row_to_cout`) so that tiles 2g and 2g+1 together give the lane the 8
// consecutive couts g*32+lc*8 .. +7: ONE 16-B store per lane and tile pair, and the 4 lane groups of
// a pixel write its whole 64-B record (1 KiB contiguous per store instruction).
template <typename T> __device__ __forceinline__ int conv_cout_ofs(int m, int lc)
{
    return sizeof(T) == 2 ? (m >> 1) * 32 + lc * 8 + (m & 1) * 4 : m * 16 + lc * 4;  // cout (within the wave) of register j = 0
}

template <typename T, int BN, int TH, int TW, int EPI>
__device__ __forceinline__ void conv_acc_init(const ConvArgs& a, f32x4 (&acc)[4][conv_wave_frags(BN, TH, TW)],
                                              int ct, int wc, int lc)
{
    constexpr int NF = conv_wave_frags(BN, TH, TW);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (!epi_is_splitk(EPI)) {  // K slices start from zero; the reduction adds the shift
            const float4 sh = *reinterpret_cast<const float4*>(a.shift + ct * BN + wc * 64 + conv_cout_ofs<T>(m, lc));
            v = f32x4{sh.x, sh.y, sh.z, sh.w};
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = v;
    }
}

// bf16 pair -> relu on the packed pair: for x < 0 (sign bit set, incl. -0.0) the int16 pattern is negative
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned p) { return pk_max_i16(p, 0u); }
// sum += max(x, lo) * w: v_max_f32 + v_fmac_f32 as one block (fmaxf() costs a canonicalising
// v_max(x, x) in front, and a lone inline-asm v_max an s_nop behind)
__device__ __forceinline__ void relu_fma(float& sum, float x, float lo, float w)
{
    float t;
    asm("v_max_f32 %1, %3, %2\n\tv_fmac_f32 %0, %1, %4" : "+v"(sum), "=&v"(t) : "v"(x), "v"(lo), "v"(w));
}

//      y = relu(acc) -> blocked activation records (+ fused pool / head / split-K slab).  acc[m][n]:
//      accumulator tile m (16 couts) x pixel fragment n of the wave (wc = cout half, wp = pixel
//      group) of workgroup tile (b, y0, x0), cout tile ct, K slice `split`.
template <typename T, int BN, int TH, int TW, int EPI, bool X2 = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[4][conv_wave_frags(BN, TH, TW)],
                                              int b, int y0, int x0, int ct, int split, int wc, int wp,
                                              int l15, int lc)
{
    constexpr int PL = Elem<T>::PL;
    constexpr int FR = TW / 16;
    constexpr int NF = conv_wave_frags(BN, TH, TW);
    constexpr int ROWS_W = NF / FR;
    constexpr int HNC = EPI == EPI_HEAD ? 1 : (EPI == EPI_HEAD3 ? 3 : 0);
    const int aH = a.H, aW = a.W;
    constexpr bool PERM = sizeof(T) == 2;
    const int wbase_c = ct * BN + wc * 64;  // first cout of this wave
    if constexpr (epi_is_splitk(EPI)) {
        // raw fp32 partial sums of this K slice, in FRAGMENT order: slab[split][tile][wave][m * NF + n][lane] as float4,
        // so every store instruction of a wave writes 1 KiB contiguous (round 6; the pixel-major slab of rounds 1-5 was
        // written 16 B at a time with a stride of Cout floats: 29 % of a K-split kernel's wave time was this epilogue,
        // profiles/r06_stamp_phases_b1_256_before_small_tiles.txt).  splitk_finalize_tile_kernel reads it back in the same order.
        const int tile = (((b * a.tilesY) + y0 / TH) * a.tilesX + x0 / TW) * a.nct + ct;
        const int ntile = a.B * a.tilesY * a.tilesX * a.nct;
        const int wave = wp * (BN / 64) + wc;
        float4* o = reinterpret_cast<float4*>(a.kslab) + ((((size_t)split * ntile + tile) * 4 + wave) * (4 * NF)) * 64 + (lc * 16 + l15);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n)
                o[(m * NF + n) * 64] = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
        return;
    }
    float hw[HNC > 0 ? HNC : 1][4][4];
#pragma unroll
    for (int c = 0; c < HNC; ++c)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float4 v = *reinterpret_cast<const float4*>(a.head_w + c * 64 + conv_cout_ofs<T>(m, lc));
            hw[c][m][0] = v.x; hw[c][m][1] = v.y; hw[c][m][2] = v.z; hw[c][m][3] = v.w;
        }
    // record address = image base + plane * plane_stride + pixel * 64 + byte within the record
    const size_t plane_stride = (size_t)aH * aW * 64;
    const int plane0 = wbase_c / PL;                       // first output plane of this wave
    const int rec_byte = lc * 16;                          // this lane's 16 B of a 64-B record
    char* const out_img = a.dst ? (char*)a.dst + (size_t)b * plane_stride * (a.Cout / PL) +
                                  (size_t)plane0 * plane_stride + rec_byte : nullptr;
    const int pH = aH >> 1, pW = aW >> 1;  // EPI_POOL: MaxPool2d(2) output size (floor)
    const size_t pplane_stride = (size_t)pH * pW * 64;
    char* const pool_img = EPI == EPI_POOL ? (char*)a.pool_dst + (size_t)b * pplane_stride * (a.Cout / PL) +
                                             (size_t)plane0 * pplane_stride + rec_byte : nullptr;
    if constexpr (HNC > 0) {
        // Fused 1x1 head on the fp32 post-activation values (never rounded to bf16).  Every lane sums
        // its 16 couts for each of the 8 fragments; the four lane groups of a pixel are then combined
        // by a transposing reduction (v_permlane16_swap / v_permlane32_swap: 6 swaps instead of 16
        // bpermutes, same (lc0 + lc1) + (lc2 + lc3) order), which leaves fragments nb, nb + 1 complete
        // in lane group lc: all 64 lanes store, 2 dwords each.
        static_assert(HNC == 0 || NF == 8 || NF == 4, "head reduction is written for 8 or 4 fragments per wave");
        const float floor_v = a.relu ? 0.f : -__builtin_inff();
        float hb[HNC];
#pragma unroll
        for (int c = 0; c < HNC; ++c) hb[c] = a.head_b[c];
        const int nb = ((lc & 1) ? 4 : 0) + ((lc & 2) ? 2 : 0);
#pragma unroll
        for (int c = 0; c < HNC; ++c) {
            float hs[NF];
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                float sum = 0.f;
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) relu_fma(sum, acc[m][n][j], floor_v, hw[c][m][j]);
                hs[n] = sum;
            }
            if constexpr (NF == 4) {
                // 64 x 64 wave tiles (small problems): four fragments, one per lane group after the reduction.  Two
                // xor-shuffles per fragment give every lane (g0 + g1) + (g2 + g3) of its pixel - the association of
                // the transposing reduction below (fp addition commutes), so the output bits do not depend on the tile
                float full[NF];
#pragma unroll
                for (int n = 0; n < NF; ++n) {
                    const float p = hs[n] + __shfl_xor(hs[n], 16);
                    full[n] = p + __shfl_xor(p, 32);
                }
                const float mine = lc == 0 ? full[0] : (lc == 1 ? full[1] : (lc == 2 ? full[2] : full[3]));
                const int n = lc;
                const int y = y0 + wp * ROWS_W + n / FR;
                const int x = x0 + (n % FR) * 16 + l15;
                if (y < aH && x < aW) {
                    const size_t o = (size_t)b * a.head_img_stride + ((size_t)c * aH + y) * aW + x;
                    if (a.head_out_u8) a.head_out_u8[o] = postprocess_u8_value(mine + hb[c]);
                    else a.head_out[o] = mine + hb[c];
                }
            } else {
                float t[4], u[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {  // rows 0, 2: fragment i; rows 1, 3: fragment i + 4
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(hs[i]), __float_as_uint(hs[(i + 4) % NF]), false, false);
                    t[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {  // rows 0, 1: t[j]; rows 2, 3: t[j + 2]
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[j]), __float_as_uint(t[j + 2]), false, false);
                    u[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = nb + j;
                    const int y = y0 + wp * ROWS_W + n / FR;
                    const int x = x0 + (n % FR) * 16 + l15;
                    if (y < aH && x < aW) {
                        const size_t o = (size_t)b * a.head_img_stride + ((size_t)c * aH + y) * aW + x;
                        if (a.head_out_u8) a.head_out_u8[o] = postprocess_u8_value(u[j] + hb[c]);
                        else a.head_out[o] = u[j] + hb[c];
                    }
                }
            }
        }
    }
    if constexpr (X2) {
        // precision "bf16x2": relu(acc) in fp32 -> two bf16 pieces, stored [hi planes | lo planes] (two 16-B stores per
        // lane and tile pair; a fused-head conv stores only when the read-back keeps its activation)
        static_assert(sizeof(T) == 2, "the two-piece epilogue belongs to the bf16 kernels");
        if (HNC > 0 && !a.dst) return;
        const size_t ps = (size_t)aH * aW * 64;                  // bytes of one 32-channel plane
        const int npo = a.Cout / 32;                             // planes per piece; the tensor has 2 * npo
        const size_t blk = (size_t)npo * ps;                     // bytes from the hi piece to the lo piece
        char* const img = (char*)a.dst + (size_t)b * 2 * blk + (size_t)(wbase_c / 32) * ps + lc * 16;
        const int pH2 = aH >> 1, pW2 = aW >> 1;
        const size_t pps = (size_t)pH2 * pW2 * 64, pblk = (size_t)npo * pps;
        char* const pimg = EPI == EPI_POOL ? (char*)a.pool_dst + (size_t)b * 2 * pblk + (size_t)(wbase_c / 32) * pps + lc * 16
                                           : nullptr;
        // 8 consecutive couts of this lane (accumulator tiles 2g, 2g + 1) -> hi and lo chunk
        auto put = [&](char* o, size_t block_bytes, const float (&v)[8]) __attribute__((always_inline)) {
            unsigned h[4], l[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h[i] = pack_bf16x2_pk(v[2 * i], v[2 * i + 1]);
                l[i] = pack_bf16x2_pk(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
            }
            *reinterpret_cast<uint4*>(o) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(o + block_bytes) = make_uint4(l[0], l[1], l[2], l[3]);
        };
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            const int y = y0 + wp * ROWS_W + n / FR;
            const int x = x0 + (n % FR) * 16 + l15;
            if (y < aH && x < aW) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = a.relu ? fmaxf(acc[2 * g][n][j], 0.f) : acc[2 * g][n][j];
                        v[4 + j] = a.relu ? fmaxf(acc[2 * g + 1][n][j], 0.f) : acc[2 * g + 1][n][j];
                    }
                    put(img + (size_t)(y * aW + x) * 64 + g * ps, blk, v);
                }
            }
        }
        if constexpr (EPI == EPI_POOL) {
            // MaxPool2d(2) on the fp32 values (relu is monotonic): rows n / n + FR of this wave, column partner = lane ^ 1
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                if (((n / FR) & 1) != 0) continue;
                const int y = y0 + wp * ROWS_W + n / FR;
                const int x = x0 + (n % FR) * 16 + l15;
                const int py = y >> 1, px = x >> 1;
                const bool okp = (py < pH2) && (px < pW2) && ((l15 & 1) == 0);
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float cm = fmaxf(acc[2 * g + h][n][j], acc[2 * g + h][n + FR][j]);
                            float r = fmaxf(cm, __uint_as_float(dpp_swap_pairs(__float_as_uint(cm))));
                            if (a.relu) r = fmaxf(r, 0.f);
                            v[4 * h + j] = r;
                        }
                    if (okp) put(pimg + (size_t)(min(py, pH2 - 1) * pW2 + min(px, pW2 - 1)) * 64 + g * pps, pblk, v);
                }
            }
        }
        return;
    }
    if constexpr (EPI == EPI_POOL && PERM) {
        // bf16 + fused MaxPool2d(2) (unet.py:28): a wave owns whole row pairs (fragments n and n + FR), the
        // column partner is the neighbouring lane (l15 ^ 1).  The packed, relu'd pairs are built ONCE and
        // used for both the full-resolution store and the pooled one (relu and the bf16 rounding are
        // monotonic and every value is >= 0 after the relu, so the packed int16 max of the stored bits IS
        // the pooled tensor; the epilogue's VALU instructions are paid at the partner wave's MFMA cadence).
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            if (((n / FR) & 1) != 0) continue;  // upper row of each pair
            uint4 pk[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int nn = n + r * FR;
                const int y = y0 + wp * ROWS_W + nn / FR;
                const int x = x0 + (nn % FR) * 16 + l15;
                const bool ok = (y < aH) && (x < aW);
                char* o = out_img + (size_t)(y * aW + x) * 64;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint4 v = make_uint4(pack_bf16x2_pk(acc[2 * g][nn][0], acc[2 * g][nn][1]),
                                         pack_bf16x2_pk(acc[2 * g][nn][2], acc[2 * g][nn][3]),
                                         pack_bf16x2_pk(acc[2 * g + 1][nn][0], acc[2 * g + 1][nn][1]),
                                         pack_bf16x2_pk(acc[2 * g + 1][nn][2], acc[2 * g + 1][nn][3]));
                    if (a.relu) v = make_uint4(relu_pk_bf16(v.x), relu_pk_bf16(v.y), relu_pk_bf16(v.z), relu_pk_bf16(v.w));
                    pk[r][g] = v;
                    if (ok && out_img) *reinterpret_cast<uint4*>(o + g * plane_stride) = v;
                }
            }
            const int y = y0 + wp * ROWS_W + n / FR;
            const int x = x0 + (n % FR) * 16 + l15;
            const int py = y >> 1, px = x >> 1;
            const bool okp = (py < pH) && (px < pW) && ((l15 & 1) == 0);
            char* o = pool_img + (size_t)(min(py, pH - 1) * pW + min(px, pW - 1)) * 64;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint4 t = pk[0][g], u = pk[1][g];
                const unsigned c0 = pk_max_i16(t.x, u.x), c1 = pk_max_i16(t.y, u.y);
                const unsigned c2 = pk_max_i16(t.z, u.z), c3 = pk_max_i16(t.w, u.w);
                const uint4 m = make_uint4(pk_max_i16(c0, dpp_swap_pairs(c0)), pk_max_i16(c1, dpp_swap_pairs(c1)),
                                           pk_max_i16(c2, dpp_swap_pairs(c2)), pk_max_i16(c3, dpp_swap_pairs(c3)));
                if (okp) *reinterpret_cast<uint4*>(o + g * pplane_stride) = m;
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NF; ++n) {
        const int y = y0 + wp * ROWS_W + n / FR;
        const int x = x0 + (n % FR) * 16 + l15;
        const bool ok = (y < aH) && (x < aW);
        if (ok && out_img) {
            char* o = out_img + (size_t)(y * aW + x) * 64;
            if constexpr (PERM) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    uint4 pk = make_uint4(pack_bf16x2(acc[2 * g][n][0], acc[2 * g][n][1]),
                                          pack_bf16x2(acc[2 * g][n][2], acc[2 * g][n][3]),
                                          pack_bf16x2(acc[2 * g + 1][n][0], acc[2 * g + 1][n][1]),
                                          pack_bf16x2(acc[2 * g + 1][n][2], acc[2 * g + 1][n][3]));
                    if (a.relu) pk = make_uint4(relu_pk_bf16(pk.x), relu_pk_bf16(pk.y), relu_pk_bf16(pk.z), relu_pk_bf16(pk.w));
                    *reinterpret_cast<uint4*>(o + g * plane_stride) = pk;
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float4 v = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
                    if (a.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    *reinterpret_cast<float4*>(o + m * plane_stride) = v;
                }
            }
        }
    }
    if (EPI == EPI_POOL) {
        // MaxPool2d(2) of this conv's output (unet.py:28), fused here so the consumer conv reads a
        // ready tensor by LDS-DMA: tile origins are even, a wave owns whole row pairs (fragments n
        // and n+FR) and the column partner is the neighbouring lane (l15 ^ 1).
        // relu and the bf16 rounding are monotonic, so max-then-relu-then-round of the raw sums
        // equals pooling the stored tensor.
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            if (((n / FR) & 1) != 0) continue;  // upper row of each pair only
            const int y = y0 + wp * ROWS_W + n / FR;
            const int x = x0 + (n % FR) * 16 + l15;
            const int py = y >> 1, px = x >> 1;
            const bool okp = (py < pH) && (px < pW) && ((l15 & 1) == 0);
            char* o = pool_img + (size_t)(min(py, pH - 1) * pW + min(px, pW - 1)) * 64;
            if constexpr (!PERM) {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    float r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float cm = fmaxf(acc[m][n][j], acc[m][n + FR][j]);
                        r[j] = fmaxf(cm, __uint_as_float(dpp_swap_pairs(__float_as_uint(cm))));
                        if (a.relu) r[j] = fmaxf(r[j], 0.f);
                    }
                    if (okp) *reinterpret_cast<float4*>(o + m * pplane_stride) = make_float4(r[0], r[1], r[2], r[3]);
                }
            } else {
                // Packed int16 max on the rounded bf16 pairs.  Among non-negative patterns it is the
                // float max; a negative pattern (sign bit) loses against every non-negative one, so
                // the result is the true maximum whenever that is >= 0 and SOME negative value
                // otherwise -- which the final relu turns into 0 either way: relu(max) exactly.
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    unsigned r[4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int m = 2 * g + h;
                        const unsigned a0 = pk_max_i16(pack_bf16x2_pk(acc[m][n][0], acc[m][n][1]),
                                                       pack_bf16x2_pk(acc[m][n + FR][0], acc[m][n + FR][1]));
                        const unsigned a1 = pk_max_i16(pack_bf16x2_pk(acc[m][n][2], acc[m][n][3]),
                                                       pack_bf16x2_pk(acc[m][n + FR][2], acc[m][n + FR][3]));
                        r[2 * h] = pk_max_i16(a0, dpp_swap_pairs(a0));
                        r[2 * h + 1] = pk_max_i16(a1, dpp_swap_pairs(a1));
                        if (a.relu) { r[2 * h] = relu_pk_bf16(r[2 * h]); r[2 * h + 1] = relu_pk_bf16(r[2 * h + 1]); }
                    }
                    if (okp) *reinterpret_cast<uint4*>(o + g * pplane_stride) = make_uint4(r[0], r[1], r[2], r[3]);
                }
            }
        }
    }
}

template <typename T, int BN, int TH, int TW, int MODE, int EPI>
__global__ __launch_bounds__(256, conv_occupancy(BN, TH, TW, ConvTile<BN, TH, TW, MODE>::LDS_BYTES)) void conv3x3_mfma_kernel(const ConvArgs a)
{
    using Tile = ConvTile<BN, TH, TW, MODE>;
    constexpr int PL = Elem<T>::PL;
    constexpr int TWP = Tile::TWP, THP = Tile::THP;
    constexpr int WAVES_C = BN / 64, WAVES_P = 4 / WAVES_C;
    constexpr int FR = TW / 16;       // 16-pixel fragments per tile row
    constexpr int NF = conv_wave_frags(BN, TH, TW);  // fragments per wave
    constexpr int ROWS_W = NF / FR;   // tile rows per wave
    static_assert(NF == 8 || NF == 4, "wave tile must be 64 couts x 128 or 64 pixels");
    static_assert(TH == ROWS_W * WAVES_P && ROWS_W * FR == NF, "tile does not split over the waves");
    static_assert(EPI != EPI_POOL || ROWS_W % 2 == 0, "pooled epilogue: a wave owns whole row pairs");
    static_assert(Tile::LDS_BYTES * conv_occupancy(BN, TH, TW, Tile::LDS_BYTES) <= 160 * 1024, "LDS per CU");
    constexpr int HNC = EPI == EPI_HEAD ? 1 : (EPI == EPI_HEAD3 ? 3 : 0);  // fused-head classes
    static_assert(HNC == 0 || BN == 64, "fused head needs all 64 couts in one wave");
    static_assert(MODE == SRC_DIRECT || MODE == SRC_CONCAT_UP || MODE == SRC_STEM || MODE == SRC_DIRECT_X2 ||
                  MODE == SRC_STEM_X2, "pooling is fused into the producer");
    // precision "bf16x2" (SRC_DIRECT_X2 / SRC_STEM_X2): two-piece activations and weights, three virtual planes per real plane
    constexpr bool X2 = src_is_x2(MODE);
    constexpr bool STEM = src_is_stem(MODE);
    constexpr bool DIRECT = MODE == SRC_DIRECT || MODE == SRC_DIRECT_X2;
    static_assert(!X2 || sizeof(T) == 2, "two-piece operands are bf16");
    static_assert(!STEM || (sizeof(T) == 2 && BN == 64), "fused stem: bf16, 64 couts");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_in = smem;
    char* const lds_w = smem + Tile::IN_BYTES;           // slot 0; slot 1 at + W_STRIDE
    const unsigned lds_in_addr = lds_addr_of(lds_in);
    const unsigned lds_w_addr = lds_addr_of(lds_w);

    // XCD-aware, bijective block remap: blocks sharing an input tile (different cout tiles) and
    // neighbouring tiles get consecutive logical ids on ONE XCD so the re-reads hit its L2.
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ksplit = epi_is_splitk(EPI) ? a.ksplit : 1;
    const int split = lid % ksplit;
    lid /= ksplit;
    const int ct = lid % a.nct;
    int t = lid / a.nct;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty * TH, x0 = tx * TW;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lc = lane >> 4;
    const int wc = wave % WAVES_C, wp = wave / WAVES_C;

#ifdef FIUNET_STAMP
    // diagnostic build: per-wave s_memtime sums per phase.  Slots: [0] total [1] prologue (whole) [2] MFMA phases
    // [3] end-of-step wait + barrier [4] plane-boundary gather (rest) [5] epilogue [6] upsample staging DMA + wait
    // [7] upsample interpolation [8] record count; the prologue again, split (round 6): [9] kernel entry -> accumulators
    // initialised (argument + BatchNorm-shift loads) [10] W(0) issue + LDS read offsets [11] gather walk (per-tile
    // offsets / masks, padding zeroed; CONCAT_UP: the lerp tables) [12] fused stem: weights + patch staging + barrier
    // [13] first in-tile: DMA issue (fused stem: the stem's evaluation) [14] wait for W(0) + the in-tile (vmcnt(0))
    // [15] the barrier that publishes them.  PSTAMP pins the code of a sub-phase in place (sched_barrier), so the
    // prologue of this build is not scheduled as the shipped one is: [1] / [0] of a build WITHOUT the split is the
    // share to quote, the split says where inside it the time sits.
    unsigned long long st_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_prev = st_t0;
#ifdef FIUNET_STAMP_PROLOG
    unsigned long long st_pprev = st_t0;
#define PSTAMP(slot) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                          st_sum[slot] += t_ - st_pprev; st_pprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSTAMP(slot) do {} while (0)
#endif
#define STAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         st_sum[slot] += t_ - st_prev; st_prev = t_; } while (0)
#define STAMP_UP(slot) STAMP(slot)
#else
#define STAMP(slot) do {} while (0)
#define STAMP_UP(slot) do {} while (0)
#define PSTAMP(slot) do {} while (0)
#endif
    f32x4 acc[4][NF];
    conv_acc_init<T, BN, TH, TW, EPI>(a, acc, ct, wc, lc);  // BatchNorm shift (scale is in the weights)
#ifdef FIUNET_STAMP_PROLOG
    asm volatile("" :: "v"(acc[3][NF - 1][3]), "v"(acc[0][0][0]));  // the shift loads have landed
#endif
    PSTAMP(9);

    // per-lane LDS read offsets (everything else is an immediate)
    const int a_off = (wc * 64 + l15) * 64 + ((lc ^ swz(l15)) << 4);  // within a weight ring slot
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        b_off[kx] = (wp * ROWS_W * TWP + kx + l15) * 64 + ((lc ^ swz(kx + l15)) << 4);

    const int nplanes_real = (a.C0 + a.C1) / PL;
    // X2: virtual plane v = 3 * real plane + j, j = 0: (xh, wh), 1: (xh, wl) on the SAME in-tile, 2: (xl, wh)
    const int nplanes_all = X2 ? 3 * nplanes_real : nplanes_real;
    const int p0 = a.C0 / PL;
    // this workgroup's K slice: (virtual) planes [pbeg, pend)
    const int pbeg = split * nplanes_all / ksplit, pend = (split + 1) * nplanes_all / ksplit;
    const int nsteps = (pend - pbeg) * 3;
    const char* wbase = (const char*)a.wgt + (size_t)ct * BN * 64;

    // ---- weight stream: each wave moves NW 1-KiB pieces (16 LDS rows) per step by LDS-DMA.  The
    //      LDS image is lane-linear, so the XOR swizzle goes on the per-lane SOURCE chunk. --------
    constexpr int NW = Tile::W_BYTES / 1024 / 4;
    // Piece j of this wave = packed rows r0 .. r0+15 with r0 = (wave * NW + j) * 16; 16 divides BN, so a
    // piece lies inside one tap: its source offset is a wave-uniform part (added to the SGPR base) plus
    // ONE per-lane register shared by all pieces (row within the piece, swizzled 16-B chunk).
    static_assert(BN % 16 == 0, "a weight piece must not straddle two taps");
    const unsigned w_lane_off = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ swz(lane >> 2)) << 4));
    auto issue_w = [&](int step) __attribute__((always_inline)) {  // step counts from this slice's start
        const int lp = step / 3, kx = step - lp * 3;
        int pl = pbeg + lp;
        if constexpr (X2) {  // weight planes: [wh of every real plane | wl of every real plane]
            const int real = pl / 3;
            pl = pl - real * 3 == 1 ? nplanes_real + real : real;
        }
        const char* wsrc = wbase + ((size_t)(pl * 9 + kx * 3) * a.Cout) * 64;  // packed [plane][kx][ky][cout]
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            lds_w_addr + (unsigned)((step & 1) * Tile::W_STRIDE + wave * NW * 1024));
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int r0 = (wave * NW + j) * 16, tap = r0 / BN;  // tap = ky within the step
            glds16s(wsrc + (size_t)(tap * (a.Cout - BN) + r0) * 64, w_lane_off, dst + j * 1024);
        }
    };
    // W(0) goes out first (slot 0; slot 1 + spare is the idle staging area for plane 0), ahead of the
    // in-tile address math below, so its round trip runs under those ~200 instructions
#ifndef FIUNET_DIAG_FREE_PROLOGUE
    issue_w(0);
#endif
#ifdef FIUNET_DIAG_NO_WSTREAM
    if (nsteps > 1) issue_w(1);   // both ring slots hold real weights (realistic operand data, hence realistic power); nothing streams afterwards
#endif
    PSTAMP(10);

    // ---- in-tile gather (a): planes stored as-is in an NHWC tensor go by LDS-DMA: piece j = 16
    //      consecutive LDS rows (pixels) x 64 B, address = plane base in SGPRs + a 32-bit per-lane
    //      offset; lanes on pixels outside the image / on the row-pitch padding are switched off
    //      (their slots were zeroed once).  No data VGPRs, one memory round trip per plane. ---------
    constexpr int NPIECE = THP * TWP / 16;
    static_assert(THP * TWP % 16 == 0, "in-tile must be a whole number of 1-KiB pieces");
    const int aH = a.H, aW = a.W;
    const char* const dma_src = (const char*)a.src0 + (size_t)b * a.H * a.W * a.C0 * sizeof(T) * (X2 ? 2 : 1);
    // SRC_DIRECT with C1 > 0: planes >= p0 come from a second full-resolution tensor (the upsampled
    // half of a concat input, materialised by upsample_kernel when several cout tiles share it)
    const char* const dma_src1 = DIRECT && a.C1 > 0
                                     ? (const char*)a.src1 + (size_t)b * a.H * a.W * a.C1 * sizeof(T) * (X2 ? 2 : 1) : nullptr;
    // The per-lane source offset of every piece is plane-invariant in the blocked layout.  Where
    // registers allow (the plain 128-cout direct kernels) it is computed once (NPW registers, 32-bit)
    // and a plane's gather costs ~8 instructions per piece; the pooled / concat / head / split-K
    // variants and the 64-cout tiles (12-14 pieces per wave), which are at the 256-VGPR limit,
    // rebuild it per plane from three registers (see pm_* below; hoisting there spills).
    // (64 x 64 wave tiles - the small-problem configuration - have 64 accumulator registers less: every direct variant hoists)
    constexpr bool HOIST = (MODE == SRC_DIRECT && BN == 128 && EPI == EPI_PLAIN) ||   // (the two-piece epilogue with it: 3 spilled registers)
                           (NF == 4 && DIRECT);
    // Rolling window of in-tile rows across the three ky taps of a step (24 instead of 36 fragment
    // reads per step): bf16 32-wide tiles.  The fp32 instantiations (4 MFMAs per fragment pair keep
    // more operands in flight) and the 16-wide tiles have no registers to spare for the extra row
    // and re-read every row for every tap instead; same tap order, same sums.
    constexpr bool ROLL = sizeof(T) == 2 && FR == 2 && (!epi_is_splitk(EPI) || NF == 4);
    constexpr int NPW = (NPIECE + 3) / 4;
    unsigned in_off[HOIST ? NPW : 1];
    const unsigned plane_bytes = (unsigned)(aH * aW) * 64u;
    // Walk over this wave's pieces: f(j, ok, off) with off = byte offset of the lane's 16 B inside a
    // plane of the source image (valid when ok).  Hoisted kernels read the stored offsets.  The others
    // keep THREE registers instead of one per piece: a wave's pieces are 64 in-tile pixels apart, so
    // (py, px) and the linear pixel index advance by constants with one conditional row wrap (64 rows
    // keep the swizzle phase; NPIECE * 16 == THP * TWP exactly, so py stays inside) - the walk is done
    // once per tile and leaves the first piece's offset plus two bit masks (piece jj valid / row wrap
    // after piece jj); a plane's gather then needs no coordinate arithmetic at all.
    unsigned pm_lin0 = 0, pm_valid = 0, pm_wrap = 0;
    static_assert(NPW <= 32, "one mask bit per piece of a wave");
    constexpr int PM_DY = 64 / TWP, PM_DX = 64 % TWP;
    const unsigned pm_dlin = (unsigned)(PM_DY * aW + PM_DX) * 64u, pm_dwrap = (unsigned)(aW - TWP) * 64u;
    if constexpr (!STEM) {
        // one division per tile; every further piece is the previous one + 64 in-tile pixels (the hoisted
        // kernels keep the resulting offsets, the others the first offset and two bit masks)
        const int row0 = wave * 16 + (lane >> 2);
        int py = row0 / TWP, px = row0 - py * TWP;
        int y = y0 - 1 + py, x = x0 - 1 + px;
        pm_lin0 = (unsigned)(y * aW + x) * 64u + (((lane & 3) ^ swz(row0)) << 4);
        unsigned lin = pm_lin0;
#pragma unroll
        for (int jj = 0; jj < NPW; ++jj) {
            const bool ok = (px < TW + 2) & ((unsigned)y < (unsigned)aH) & ((unsigned)x < (unsigned)aW);
            if constexpr (HOIST) in_off[jj] = ok ? lin : ~0u;
            pm_valid |= (ok ? 1u : 0u) << jj;
            px += PM_DX; x += PM_DX; y += PM_DY;
            lin += pm_dlin;
            if (px >= TWP) { px -= TWP; x -= TWP; y += 1; pm_wrap |= 1u << jj; lin += pm_dwrap; }
        }
    }
    auto for_pieces = [&](auto&& f) __attribute__((always_inline)) {
        if constexpr (HOIST) {
#pragma unroll
            for (int jj = 0; jj < NPW; ++jj) {
                const int j = wave + 4 * jj;
                if (j < NPIECE) f(j, in_off[jj] != ~0u, in_off[jj]);
            }
        } else {
            // opaque copies: without them hipcc hoists every piece's offset and mask out of the K loop
            // (they are plane-invariant), i.e. rebuilds the hoisted form and spills
            unsigned lin = pm_lin0, valid = pm_valid, wrap = pm_wrap;
            asm volatile("" : "+v"(lin), "+v"(valid), "+v"(wrap));
#pragma unroll
            for (int jj = 0; jj < NPW; ++jj) {
                const int j = wave + 4 * jj;
                if (j < NPIECE) f(j, ((valid >> jj) & 1u) != 0u, lin);
                // (pm_dwrap is negative mod 2^32 when the level is narrower than the in-tile pitch)
                lin += pm_dlin + (((wrap >> jj) & 1u) ? pm_dwrap : 0u);
            }
        }
    };
    // Padding slots of the in-tile (pixels outside the image, row-pitch filler) are the same for every
    // plane of the tile: they are zeroed ONCE here, and the per-plane DMAs simply leave those lanes
    // switched off (the interpolated planes of a concat conv write zeros there themselves).
    if constexpr (!STEM) {
        for_pieces([&](int j, bool ok, unsigned) __attribute__((always_inline)) {
            if (!ok) *reinterpret_cast<uint4*>(lds_in + j * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
        });
    }
    auto gather_plane_dma = [&](int plane) __attribute__((always_inline)) {
        // wave-uniform plane base in SGPRs + the lane's 32-bit offset
        const char* base;
        if constexpr (X2) {  // virtual plane -> (real plane, piece): the lo planes follow the hi planes of their tensor
            const int real = plane / 3, lo = plane - real * 3 == 2;
            base = real >= p0 ? dma_src1 + (size_t)(real - p0 + (lo ? nplanes_real - p0 : 0)) * plane_bytes
                              : dma_src + (size_t)(real + (lo ? p0 : 0)) * plane_bytes;
        } else {
            base = (MODE == SRC_DIRECT && plane >= p0) ? dma_src1 + (size_t)(plane - p0) * plane_bytes
                                                       : dma_src + (size_t)plane * plane_bytes;
        }
        for_pieces([&](int j, bool ok, unsigned off) __attribute__((always_inline)) {
            if (ok) glds16s(base, off, __builtin_amdgcn_readfirstlane(lds_in_addr + (unsigned)j * 1024u));
        });
    };

    // ---- in-tile gather (b): bilinearly upsampled planes.  The low-res source tile (<= LRH x LRW
    //      pixels of this plane) is DMA-ed into the idle weight slot + spare region, then every
    //      thread interpolates its chunks LDS -> LDS: one memory round trip per plane instead of
    //      one per batch of register loads, and no VGPRs held across it. ---------------------------
    // Work split of the interpolation.  A column = (in-tile pixel column, 16-B chunk); a wave walks
    // DOWN 64 columns at once, all lanes on the same row, so the row mapping is wave-uniform (scalar
    // branches) and the horizontally interpolated low-res rows are reused from one output row to the
    // next.  The (TW+2)*4 columns are cut into groups of 64 ("main" groups, NMAIN of them) whose rows
    // are shared out over 4/NMAIN waves each, plus a tail of NTC < 64 columns whose NTC*THP elements
    // are interpolated independently, a quarter per wave: all four waves get the same amount of work
    // (the round-1 split left waves 2-3 idle and gave waves 0-1 every row).
    constexpr int UP_NCOL = (TW + 2) * 4;
    constexpr int UP_NMAIN = UP_NCOL / 64;            // 2 (TW = 32) or 1 (TW = 16)
    constexpr int UP_NTC = UP_NCOL - UP_NMAIN * 64;   // 8 tail columns
    constexpr int UP_WPG = 4 / UP_NMAIN;              // waves per main group
    constexpr int UP_SEGR = (THP + UP_WPG - 1) / UP_WPG;
    constexpr int UP_NTAIL = UP_NTC * THP;            // tail elements
    constexpr int UP_TAILW = (UP_NTAIL + 3) / 4;      // ... per wave
    static_assert(MODE != SRC_CONCAT_UP || (UP_NMAIN >= 1 && UP_NMAIN <= 2), "interpolation work split");
    int lr_y = 0, lr_x = 0;  // low-res origin of this tile's staging window
    char* const ytab = smem + Tile::TAB_OFF;          // [THP]   {off0, off1 (bytes into staging; < 0: zero row), hy, ly}
    char* const xtab = ytab + THP * 16;               // [TW+2]  {s0, s1 (bytes; < 0: zero column), hx, lx}
    uint4 yrow = make_uint4(0u, 0u, 0u, 0u);         // lane l < THP: ytab[l] (CONCAT_UP)
    if constexpr (MODE == SRC_CONCAT_UP) {
        const UpCoord u = up_coord(a, y0 - 1, x0 - 1);
        lr_y = u.y0; lr_x = u.x0;
        // the mapping of in-tile row py / column px, evaluated ONCE per tile (it is the same for
        // every plane); published by the barrier that ends the prologue
        {   // every wave keeps the row table in its own lanes 0 .. THP-1 as well: the column walk fetches
            // row py's entry with v_readlane (no LDS round trip per row; the walk is latency-bound)
            const int y = y0 - 1 + min(lane, THP - 1);
            const UpAxis uy = up_axis_y(a, min(max(y, 0), a.H - 1));
            const bool ok = uy.ok & (y >= 0) & (y < a.H);
            yrow = make_uint4(ok ? (unsigned)((uy.i0 - lr_y) * (Tile::LRP * 64)) : 0x80000000u,
                              (unsigned)((uy.i1 - lr_y) * (Tile::LRP * 64)), __float_as_uint(uy.h),
                              __float_as_uint(uy.l));
            if (tid < THP) *reinterpret_cast<uint4*>(ytab + tid * 16) = yrow;
        }
        if (tid >= 64 && tid < 64 + TW + 2) {
            const int px = tid - 64, x = x0 - 1 + px;
            const UpAxis ux = up_axis_x(a, min(max(x, 0), a.W - 1));
            const bool ok = ux.ok & (x >= 0) & (x < a.W);
            *reinterpret_cast<uint4*>(xtab + px * 16) =
                make_uint4(ok ? (unsigned)((ux.i0 - lr_x) * 64) : 0x80000000u, (unsigned)((ux.i1 - lr_x) * 64),
                           __float_as_uint(ux.h), __float_as_uint(ux.l));
        }
    }
    auto gather_plane_up = [&](int plane, int idle_slot) __attribute__((always_inline)) {
        constexpr int LRP = Tile::LRP, LRH = Tile::LRH;
        // idle slot 0: [slot0 | spare]; idle slot 1: [spare | slot1]
        const int stg_off = idle_slot == 0 ? 0 : Tile::W_BYTES;
        const char* const lsrc = (const char*)a.src1 + (size_t)b * a.lowH * a.lowW * a.C1 * sizeof(T) +
                                 (size_t)(plane - p0) * a.lowH * a.lowW * 64;
        int opq = 0;
        asm volatile("" : "+s"(opq));
        int l4 = (lane & 3) << 4;          // opaque: hipcc would otherwise form lsrc + l4 as a 64-bit
        asm volatile("" : "+v"(l4));       // per-lane pointer ahead of the K loop (and spill it)
#pragma unroll 1
        for (int j = wave; j < Tile::LR_PIECES; j += 4) {
            const int row = j * 16 + (lane >> 2) + opq;
            const int r = row / LRP, c = row - r * LRP;
            const int gy = min(lr_y + min(r, LRH - 1), a.lowH - 1), gx = min(lr_x + c, a.lowW - 1);
            const char* src = lsrc + ((unsigned)(gy * a.lowW + gx) * 64u + (unsigned)l4);
            glds16(src, __builtin_amdgcn_readfirstlane(lds_w_addr + (unsigned)(stg_off + j * 1024)));
        }
        lds_dma_wait_all();
        __syncthreads();
        STAMP_UP(6);
        const char* const stg = lds_w + stg_off;
        constexpr int NE = Elem<T>::NE;
        // ---- main groups: column walk over this wave's row segment ------------------------------
        {
            const int grp = UP_NMAIN == 1 ? 0 : (wave & 1), seg = UP_NMAIN == 1 ? wave : (wave >> 1);
            const int col = grp * 64 + lane, px = col >> 2, ch = col & 3;
            const uint4 xt = *reinterpret_cast<const uint4*>(xtab + px * 16);
            const bool okx = (int)xt.x >= 0;
            const char* const s0 = stg + (okx ? (int)xt.x : 0) + ch * 16;
            const char* const s1 = stg + (okx ? (int)xt.y : 0) + ch * 16;
            // columns outside the upsampled extent (F.pad / conv padding) get zero weights instead of a
            // per-row select: every source value is a ReLU output (>= +0), so 0 * a + 0 * b = +0 exactly
            const float hx = okx ? __uint_as_float(xt.z) : 0.f, lx = okx ? __uint_as_float(xt.w) : 0.f;
            float h0[NE], h1[NE];
#pragma unroll
            for (int i = 0; i < NE; ++i) h0[i] = h1[i] = 0.f;
            int c0 = -1, c1 = -1;  // staging row offsets currently held in h0 / h1 (wave-uniform)
            const int rbeg = seg * UP_SEGR, rend = min(THP, rbeg + UP_SEGR);
            // Row py's table entry comes from the wave's own lane py (v_readlane: no LDS round trip per row;
            // up3.0 -1.5 %, up4.0 -2.3 % against the LDS read, profiles/r03_ab_lerp_variants.txt).  Tried on
            // top of it and dropped, same file: the one new low-res row of the NEXT output row requested an
            // iteration early (1-1.5 % slower: the walk is not bound by LDS latency), the arithmetic on
            // float pairs (v_pk_fma_f32: null), s_setprio 3 for the walk (null).
#pragma unroll 1
            for (int py = rbeg; py < rend; ++py) {
#ifdef FIUNET_YTAB_LDS  // A/B: the row entry through LDS
                const uint4 yt = *reinterpret_cast<const uint4*>(ytab + py * 16);  // same address in every lane
                const int o0 = __builtin_amdgcn_readfirstlane((int)yt.x);
                const int o1 = __builtin_amdgcn_readfirstlane((int)yt.y);
#else
                const uint4 yt = make_uint4((unsigned)__builtin_amdgcn_readlane((int)yrow.x, py),
                                            (unsigned)__builtin_amdgcn_readlane((int)yrow.y, py),
                                            (unsigned)__builtin_amdgcn_readlane((int)yrow.z, py),
                                            (unsigned)__builtin_amdgcn_readlane((int)yrow.w, py));
                const int o0 = (int)yt.x, o1 = (int)yt.y;
#endif
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (o0 >= 0) {  // rows outside the image / the upsampled extent stay zero (conv pad, F.pad)
                    if (o0 != c0) {
                        if (o0 == c1) {
                            // (swapping the roles of h0 / h1 instead of copying - two instantiations of the
                            // row step chosen by a wave-uniform flag - measured 4-9 % SLOWER on up3.0 / up4.0)
#pragma unroll
                            for (int i = 0; i < NE; ++i) h0[i] = h1[i];
                        } else {
                            chunk_hlerp<T>(*reinterpret_cast<const uint4*>(s0 + o0),
                                           *reinterpret_cast<const uint4*>(s1 + o0), hx, lx, h0);
                        }
                        c0 = o0;
                    }
                    if (o1 != c1) {
                        if (o1 == c0) {
#pragma unroll
                            for (int i = 0; i < NE; ++i) h1[i] = h0[i];
                        } else {
                            chunk_hlerp<T>(*reinterpret_cast<const uint4*>(s0 + o1),
                                           *reinterpret_cast<const uint4*>(s1 + o1), hx, lx, h1);
                        }
                        c1 = o1;
                    }
                    v = chunk_vlerp<T>(h0, h1, __uint_as_float(yt.z), __uint_as_float(yt.w));
                }
                const int row = py * TWP + px;
                *reinterpret_cast<uint4*>(lds_in + row * 64 + ((ch ^ swz(row)) << 4)) = v;
            }
        }
        // ---- tail columns: UP_NTC x THP elements, a quarter per wave, each interpolated on its own
        //      (chunk_bilerp = chunk_hlerp + chunk_vlerp operation for operation: the same bits)
        for (int k = lane; k < UP_TAILW; k += 64) {
            const int e = wave * UP_TAILW + k;
            if (e < UP_NTAIL) {
                const int py = e / UP_NTC, col = UP_NMAIN * 64 + (e - py * UP_NTC), px = col >> 2, ch = col & 3;
                const uint4 xt = *reinterpret_cast<const uint4*>(xtab + px * 16);
                const uint4 yt = *reinterpret_cast<const uint4*>(ytab + py * 16);
                const bool ok = ((int)xt.x >= 0) & ((int)yt.x >= 0);
                const char* const q0 = stg + (ok ? (int)xt.x : 0) + ch * 16;
                const char* const q1 = stg + (ok ? (int)xt.y : 0) + ch * 16;
                const int o0 = ok ? (int)yt.x : 0, o1 = ok ? (int)yt.y : 0;
                uint4 v = chunk_bilerp<T>(*reinterpret_cast<const uint4*>(q0 + o0), *reinterpret_cast<const uint4*>(q1 + o0),
                                          *reinterpret_cast<const uint4*>(q0 + o1), *reinterpret_cast<const uint4*>(q1 + o1),
                                          __uint_as_float(xt.z), __uint_as_float(xt.w), __uint_as_float(yt.z),
                                          __uint_as_float(yt.w));
                if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
                const int row = py * TWP + px;
                *reinterpret_cast<uint4*>(lds_in + row * 64 + ((ch ^ swz(row)) << 4)) = v;
            }
        }
        STAMP_UP(7);
    };

    // ---- in-tile gather (c): SRC_STEM.  The plane's 32 channels of relu(bn(conv3x3(frames))) are
    //      computed right here for the (TH+2)x(TW+2) window, from a raw fp32 patch of the two frames
    //      staged once per tile.  bf16 MFMA with the operands split hi+lo (x = xh + xl, w = wh + wl;
    //      xh*wh + xl*wh + xh*wl, fp32 accumulate) keeps ~2^-16 relative accuracy, far below the
    //      bf16 rounding of the result.  k = tap*2 + frame (18 of 32 slots used).
    constexpr int NIN = THP * (TW + 2);  // in-tile pixels
    // patch image: bf16, [row][hi | lo][col][frame]; the 8 k-slots of a lane group are the 16 contiguous
    // bytes (4 columns x 2 frames) at (row + dy, col): k = lane group*8 + dx*2 + frame, zero weights for dx = 3.
    char* const patch = smem + Tile::PATCH_OFF;
    constexpr int PW = Tile::PATCH_W, PH = Tile::PATCH_H, PP = Tile::PATCH_PITCH;
    auto stage_patch = [&]() __attribute__((always_inline)) {
        if constexpr (STEM) {
            // one thread = one patch pixel (both frames): one address, one dither value, one dword store
            // for the hi pair and one for the lo pair (the prologue is VALU-bound: SQ_INSTS_VALU per wave
            // tracks its time, profiles/r03_pmc_ab_stem.txt)
            unsigned* const pd = reinterpret_cast<unsigned*>(patch);  // dwords {frame1, frame2}; lo part: + PW
            constexpr int NP = PH * PW;               // patch pixels
            constexpr int NB = (NP + 255) / 256;      // all loads are issued before the first use
            float v0[NB], v1[NB];
            unsigned okm = 0;
            // unconditional loads from clamped coordinates (all issued back to back, one wait), the
            // out-of-image pixels are zeroed afterwards
            size_t at[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = tid + k * 256;
                const int py = r / PW, px = r - py * PW;
                const int y = y0 - 2 + py, x = x0 - 2 + px;
                const bool ok = (r < NP) & (y >= 0) & (y < aH) & (x >= 0) & (x < aW);
                okm |= (ok ? 1u : 0u) << k;
                at[k] = ((size_t)b * aH + min(max(y, 0), aH - 1)) * aW + min(max(x, 0), aW - 1);
            }
            if (a.u1) {  // wave-uniform: fiunet_forward_u8 reads the uint8 frames right here
                unsigned char q0[NB], q1[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) { q0[k] = a.u1[at[k]]; q1[k] = a.u2[at[k]]; }
#pragma unroll
                for (int k = 0; k < NB; ++k) { v0[k] = preprocess_u8_value(q0[k]); v1[k] = preprocess_u8_value(q1[k]); }
            } else {
#pragma unroll
                for (int k = 0; k < NB; ++k) { v0[k] = a.f1[at[k]]; v1[k] = a.f2[at[k]]; }
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = tid + k * 256;
                const int py = r / PW, px = r - py * PW;
                // ordered input dither, +d on frame 1 and -d on frame 2 (stem_dither); the conv's zero
                // padding stays exactly zero
                const float d = a.dither * stem_dither(y0 - 2 + py, x0 - 2 + px);
                const bool ok = (okm >> k) & 1u;
                v0[k] = ok ? v0[k] + d : 0.f;
                v1[k] = ok ? v1[k] - d : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = tid + k * 256;
                const int py = r / PW, px = r - py * PW;
                const unsigned hi = pack_bf16x2_pk(v0[k], v1[k]);
                const unsigned lo = pack_bf16x2_pk(v0[k] - __uint_as_float(hi << 16), v1[k] - __uint_as_float(hi & 0xffff0000u));
                if (r < NP) {
                    pd[py * PP + px] = hi;
                    pd[py * PP + px + PW] = lo;
                }
            }
            // bias operand (in the in-tile's row-pitch filler): PW dwords {1.0, 0}, then PW zero dwords (lo part)
            if (tid < 2 * PW) reinterpret_cast<unsigned*>(smem + Tile::PATCH_TAIL_OFF)[tid] = tid < PW ? 0x00003f80u : 0u;
        }
    };
    // The plane's A operands: two 16-cout tiles, hi and lo parts, BatchNorm scale folded in on the
    // host and the BatchNorm shift riding in k-slot 24 (lane group 3, whose data operand is 1.0).
    // Plane 0's come straight from global memory, requested before the patch is staged; plane 1's
    // are parked in LDS by DMA at kernel start and picked up at the plane boundary.
    struct StemW { uint4 wh[2], wl[2]; };
    StemW stem_w0;
    auto stem_park = [&](int plane) __attribute__((always_inline)) {   // LDS-DMA of one plane's stem weights into STEMW
        if constexpr (STEM && X2)   // issued again inside the K loop: scalar base + 32-bit lane offset, so that no 64-bit
            glds16s((const char*)a.stem_w + (wave >> 1) * 4096 + (2 * plane + (wave & 1)) * 1024, (unsigned)lane * 16u,   // per-lane pointer lives across it
                    __builtin_amdgcn_readfirstlane(lds_in_addr + (unsigned)(Tile::STEMW_OFF + wave * 1024)));
        else if constexpr (STEM)
            glds16((const char*)a.stem_w + (wave >> 1) * 4096 + (2 * plane + (wave & 1)) * 1024 + lane * 16,
                   __builtin_amdgcn_readfirstlane(lds_in_addr + (unsigned)(Tile::STEMW_OFF + wave * 1024)));
    };
    auto stem_load = [&](StemW& w) __attribute__((always_inline)) {
        if constexpr (STEM) {
            int lane_ofs = l15 * 64 + lc * 16;     // opaque: keeps hipcc from forming the 64-bit
            asm volatile("" : "+v"(lane_ofs));     // per-lane pointer before the K loop and spilling it
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                w.wh[h] = ldg16((const char*)a.stem_w + h * 1024 + lane_ofs);
                w.wl[h] = ldg16((const char*)a.stem_w + 4096 + h * 1024 + lane_ofs);
            }
            // plane 1 (X2: plane 0 again - its lo piece is evaluated from the same weights; plane 1's follow by
            // stem_park(1) once every wave has picked these up): wave w moves [hi h0 | hi h1 | lo h0 | lo h1][w], lane-linear
            stem_park(X2 ? 0 : 1);
        }
    };
    auto stem_parked = [&](StemW& w) __attribute__((always_inline)) {
        if constexpr (STEM) {
            const char* const base = smem + Tile::STEMW_OFF + l15 * 64 + lc * 16;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                w.wh[h] = *reinterpret_cast<const uint4*>(base + h * 1024);
                w.wl[h] = *reinterpret_cast<const uint4*>(base + 2048 + h * 1024);
            }
        }
    };
    auto gather_plane_stem = [&](const StemW& w, bool lo_piece = false) __attribute__((always_inline)) {
        if constexpr (STEM) {
            // lane groups 0, 1, 2 read patch rows py, py+2, py+1 (so that the two groups served together
            // by a ds_read2_b32, lanes 0-31, sit 2 * pitch = 16 banks apart); lane group 3 (k = 24..31, only
            // k = 24 has a weight: the BatchNorm shift) reads the {1.0, 0} dwords / their zero lo part at one
            // fixed address (a broadcast: at most one bank shared with lane group 2)
            constexpr int TAIL_IDX = (Tile::PATCH_TAIL_OFF - Tile::PATCH_OFF) / 4;  // dword index relative to the patch
            const unsigned* const ph32 = reinterpret_cast<const unsigned*>(patch);
            const unsigned* const pl32 = ph32 + PW;  // the lo part of a patch row follows its hi part
            struct Frag { uint4 bh, bl; int dst; };  // dst = in-tile row of the lane's pixel, >= THP*TWP: lane beyond the in-tile
            // A wave's fragments are 64 in-tile pixels apart: the lane's patch index `e` and in-tile row `dst`
            // advance by constants, plus a constant more when the pixel column wraps into the next row.
            // Every VALU instruction here is paid at the partner wave's MFMA cadence (~16 cycles while the
            // co-resident workgroup streams MFMAs on this SIMD), so the loop carries nothing it can do
            // without: no multiplies, no coordinates besides the column, and in-tile pixels outside the
            // image are NOT tested here - the few tiles on the image border re-zero them in a separate pass.
            constexpr int DY = 64 / (TW + 2), DX = 64 % (TW + 2);
            const int py0 = (wave * 16 + l15) / (TW + 2);
            int px = wave * 16 + l15 - py0 * (TW + 2);
            // lane group 3's address does not move (all its increments are zero)
            int e = lc < 3 ? py0 * PP + px + (lc == 0 ? 0 : (lc == 1 ? 2 * PP : PP)) : TAIL_IDX;
            const int e_step = lc < 3 ? DY * PP + DX : 0, e_wrap = lc < 3 ? PP - (TW + 2) : 0;
            int dst = py0 * TWP + px;
            auto fetch = [&](Frag& f) __attribute__((always_inline)) {
                // lanes past the last in-tile pixel (last fragment only) read one row beyond the patch - the
                // parked stem weights, finite values - and store nothing
                f.bh = make_uint4(ph32[e], ph32[e + 1], ph32[e + 2], ph32[e + 3]);
                f.bl = make_uint4(pl32[e], pl32[e + 1], pl32[e + 2], pl32[e + 3]);
                f.dst = dst;
                px += DX;
                const bool wrap = px >= TW + 2;
                px -= wrap ? TW + 2 : 0;
                e += e_step + (wrap ? e_wrap : 0);
                dst += DY * TWP + DX + (wrap ? TWP - (TW + 2) : 0);
            };
            auto compute = [&](const Frag& f) __attribute__((always_inline)) {
                f32x4 s4[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) s4[h] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int h = 0; h < 2; ++h) mma_chunk<T>(s4[h], w.wl[h], f.bh);
#pragma unroll
                for (int h = 0; h < 2; ++h) mma_chunk<T>(s4[h], w.wh[h], f.bl);
#pragma unroll
                for (int h = 0; h < 2; ++h) mma_chunk<T>(s4[h], w.wh[h], f.bh);
                // packed row lc*4 + j of tile h is channel lc*8 + h*4 + j of the plane (host:
                // bf16_row_to_cout): the lane's 8 values are ONE 16-B chunk of the pixel's record - one
                // ds_write_b128 (2-way bank conflicts) instead of two ds_write_b64 (4-way)
                const int row = f.dst;
                uint4 pk = make_uint4(relu_pk_bf16(pack_bf16x2_pk(s4[0][0], s4[0][1])),
                                      relu_pk_bf16(pack_bf16x2_pk(s4[0][2], s4[0][3])),
                                      relu_pk_bf16(pack_bf16x2_pk(s4[1][0], s4[1][1])),
                                      relu_pk_bf16(pack_bf16x2_pk(s4[1][2], s4[1][3])));
                if constexpr (X2) {
                    if (lo_piece) {   // (wave-uniform) lo = RNE bf16 of relu(v) - hi; a negative v has hi = lo = +0
                        auto lo2 = [](float v0, float v1, unsigned hi) __attribute__((always_inline)) {
                            return pack_bf16x2_pk(fmaxf(v0, 0.f) - __uint_as_float(hi << 16),
                                                  fmaxf(v1, 0.f) - __uint_as_float(hi & 0xffff0000u));
                        };
                        pk = make_uint4(lo2(s4[0][0], s4[0][1], pk.x), lo2(s4[0][2], s4[0][3], pk.y),
                                        lo2(s4[1][0], s4[1][1], pk.z), lo2(s4[1][2], s4[1][3], pk.w));
                    }
                }
                // address = row * 64 + ((lc ^ swz(row)) << 4), with the swizzle as one xor of bit 5
                if (row < THP * TWP) *reinterpret_cast<uint4*>(lds_in + ((row * 64 + lc * 16) ^ ((row & 4) << 3))) = pk;
            };
            // two fragments per trip, operands of the next one in flight during the MFMAs of this one
            constexpr int NQ = (NIN + 15) / 16;
            Frag fa, fb;
            fetch(fa);
#pragma unroll 1
            for (int q = wave; q < NQ; q += 8) {
                fetch(fb);
                compute(fa);
                fetch(fa);
                if (q + 4 < NQ) compute(fb);
            }
            // Tiles on the image border (6 % of them at 1080p): the in-tile pixels outside the image are
            // the NEXT conv's zero padding, not stem outputs - overwrite them with zeros once every wave's
            // stores are done (workgroup-uniform branch; the caller's barrier publishes the result).
            if ((y0 == 0) | (x0 == 0) | (y0 + TH >= aH) | (x0 + TW >= aW)) {
                __syncthreads();
                for (int i = tid; i < THP * (TW + 2) * 4; i += 256) {
                    const int q = i >> 2, ch = i & 3;
                    const int qy = q / (TW + 2), qx = q - qy * (TW + 2);
                    const bool in = ((unsigned)(y0 - 1 + qy) < (unsigned)aH) & ((unsigned)(x0 - 1 + qx) < (unsigned)aW);
                    const int row = qy * TWP + qx;
                    if (!in) *reinterpret_cast<uint4*>(lds_in + row * 64 + ((ch ^ swz(row)) << 4)) = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
    };
    auto gather_plane = [&](int plane, int idle_slot, bool first) __attribute__((always_inline)) {
#ifdef FIUNET_DIAG_NO_GATHER  // timing diagnostic: compute side alone (stale LDS, results are garbage)
        return;
#endif
        if constexpr (STEM && X2) {   // virtual planes 0 / 2 / 3 / 5 = (plane 0 hi, plane 0 lo, plane 1 hi, plane 1 lo)
            if (first) {
                gather_plane_stem(stem_w0, false);
            } else {
                StemW w;
                stem_parked(w);   // plane 0's weights at virtual plane 2, plane 1's at 3 and 5 (see the K loop)
                gather_plane_stem(w, plane == 2 || plane == 5);
            }
        } else if constexpr (STEM) {  // exactly two planes (64 stem channels)
            if (first) {
                gather_plane_stem(stem_w0);
            } else {
                StemW w1;
                stem_parked(w1);
                gather_plane_stem(w1);
            }
        }
        else if (DIRECT || plane < p0) gather_plane_dma(plane);
        else gather_plane_up(plane, idle_slot);
    };

    PSTAMP(11);
    if constexpr (STEM) {
        stem_load(stem_w0);
        stage_patch();
        __syncthreads();
        PSTAMP(12);
    }
    // timing diagnostics (results are garbage, only the clock matters; round 6, review item 1): -DFIUNET_DIAG_FREE_PROLOGUE drops W(0) and
    // the first in-tile gather of every kernel that fetches it by DMA - the kernel as it would run if a cross-tile prefetch
    // had delivered both for free (the upper bound of lever (a)); -DFIUNET_DIAG_NO_WSTREAM drops the whole weight stream - the
    // upper bound of weights held in LDS for the kernel's lifetime (lever (b))
    // (staging area of an upsampled first plane - a K slice may start in the upsampled half: slot 1, W(0) being in slot 0)
#ifdef FIUNET_DIAG_FREE_PROLOGUE
    if constexpr (STEM || MODE == SRC_CONCAT_UP) gather_plane(pbeg, 1, true);
#else
    gather_plane(pbeg, 1, true);
#endif
    PSTAMP(13);
    lds_dma_wait_all();
    PSTAMP(14);
    __syncthreads();
    PSTAMP(15);
    STAMP(1);
#ifdef FIUNET_CLOCK
    // diagnostic build (-DFIUNET_CLOCK, no other stamp executes): shader cycles (s_memtime) and constant 100 MHz ticks
    // (s_memrealtime) ONCE around the K loop -> the clock this wave ran at (MI355X_MICROARCH.md, DVFS give-back item 6)
    const unsigned long long ck_t0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    int step = 0;
    int vj = X2 ? pbeg % 3 : 0;   // X2: piece combination of the current virtual plane
    for (int plane = pbeg; plane < pend; ++plane) {
        // X2: the plane after an (xh, wh) plane multiplies the same in-tile by wl - no gather, no boundary
        const bool gather_next = plane + 1 < pend && !(X2 && vj == 0);
        if constexpr (X2) vj = vj == 2 ? 0 : vj + 1;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++step) {
            // W(step) and the in-tile of `plane` are resident.  Stream W(step+1) into the other
            // ring slot (its last readers passed the barrier that ended step-1).
#ifndef FIUNET_DIAG_NO_WSTREAM
            if (step + 1 < nsteps) issue_w(step + 1);
#endif
            const char* wcur = lds_w + (step & 1) * Tile::W_STRIDE + a_off;
            // this wave's in-tile rows 0 .. ROWS_W+1 at column offset kx: row i serves tap ky for the
            // output row i - ky, so every fragment is read once and used by up to three taps
            // Last step of a plane whose successor arrives by plain LDS-DMA: after the load of the last
            // in-tile row (tap ky = 1) the rolling window holds everything the rest of the step needs,
            // so a barrier there declares the in-tile dead and the next plane's DMA is issued BEFORE
            // the remaining 64 MFMAs, which cover its round trip (no second in-tile buffer needed).
#ifndef FIUNET_NO_EARLY_GATHER
            constexpr bool EARLY_OK = ROLL && DIRECT;  // (the concat kernels would spill 6-8 registers)
#else
            constexpr bool EARLY_OK = false;
#endif
            bool early = false;
            if constexpr (EARLY_OK)
                early = kx == 2 && gather_next && (DIRECT || plane + 1 < p0);
            if constexpr (ROLL) {
                uint4 xb[ROWS_W + 2][FR];
                auto load_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                    for (int f = 0; f < FR; ++f)
                        xb[i][f] = *reinterpret_cast<const uint4*>(lds_in + b_off[kx] + (i * TWP + f * 16) * 64);
                };
#pragma unroll
                for (int i = 0; i < ROWS_W; ++i) load_row(i);   // rows of tap ky = 0
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    uint4 wa[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        wa[m] = *reinterpret_cast<const uint4*>(wcur + (ky * BN + m * 16) * 64);
                    if (ky < 2) load_row(ROWS_W + ky);  // the one new row of the next tap (row ky dies after this one)
                    if (ky == 1 && early) {
                        // that was this wave's last read of the plane's in-tile (rows 1 .. ROWS_W+1 are
                        // in registers): once every wave is here the in-tile is dead and the next
                        // plane's DMA goes out, 64 MFMAs ahead of the boundary
                        __syncthreads();
                        gather_plane(plane + 1, step & 1, false);
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < NF; ++n) {
#ifndef FIUNET_DIAG_NO_MFMA  // timing diagnostic: memory side alone (results are garbage)
                            mma_chunk<T>(acc[m][n], wa[m], xb[n / FR + ky][n % FR]);
#endif
                        }
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // compiler-only barrier: without it hipcc merges the fragment reads that
                    // consecutive taps share and keeps them in registers, i.e. builds the rolling
                    // window after all (and spills)
                    asm volatile("" ::: "memory");
                    uint4 wa[4], xb[NF];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        wa[m] = *reinterpret_cast<const uint4*>(wcur + (ky * BN + m * 16) * 64);
#pragma unroll
                    for (int n = 0; n < NF; ++n)
                        xb[n] = *reinterpret_cast<const uint4*>(
                            lds_in + b_off[kx] + (((n / FR) + ky) * TWP + (n % FR) * 16) * 64);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < NF; ++n) {
#ifndef FIUNET_DIAG_NO_MFMA
                            mma_chunk<T>(acc[m][n], wa[m], xb[n]);
#endif
                        }
                }
            }
#ifdef FIUNET_STAMP
            asm volatile("" :: "v"(acc[3][NF - 1][3]));  // keep the stamp behind the last MFMA
            __builtin_amdgcn_sched_barrier(0);
#endif
            STAMP(2);
            if (kx == 2 && gather_next && !early) {
                __syncthreads();  // every wave is done with this plane's in-tile and with W(step)
                gather_plane(plane + 1, step & 1, false);
                lds_dma_wait_all();
                __syncthreads();
                // fused x2 stem: every wave has read plane 0's parked weights for the last time (lo piece): plane 1's
                // take their place, landing under the next three steps (each step end waits for this wave's DMAs)
                if constexpr (STEM && X2) { if (plane + 1 == 2) stem_park(1); }
                STAMP(4);
            } else {
                lds_dma_wait_all();   // this wave's pieces of W(step+1) landed
                __syncthreads();      // ... and so have everyone else's
                STAMP(3);
            }
        }
    }

#ifdef FIUNET_CLOCK
    {
        asm volatile("" :: "v"(acc[3][NF - 1][3]));  // keep the stamp behind the last MFMA
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long ck_t1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
        const unsigned st_idx = blockIdx.x * 4u + (unsigned)wave;
        if (a.stamp && lane == 0 && st_idx < a.stamp_cap) {  // one private 128-B record per wave, read by nothing else in the kernel
            unsigned long long* rec = a.stamp + (size_t)st_idx * 16;
            rec[0] = ck_t1 - ck_t0;
            rec[1] = ck_r1 - ck_r0;
            rec[8] = 1ull;
        }
    }
#endif
    conv_epilogue<T, BN, TH, TW, EPI, X2>(a, acc, b, y0, x0, ct, split, wc, wp, l15, lc);
#ifdef FIUNET_STAMP
    STAMP(5);
    st_sum[0] = st_prev - st_t0;
    const unsigned st_idx = blockIdx.x * 4u + (unsigned)wave;
    if (a.stamp && lane == 0 && st_idx < a.stamp_cap) {  // one private 128-B record per wave: no atomics, no contention
        unsigned long long* rec = a.stamp + (size_t)st_idx * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) rec[k] = k == 8 ? 1ull : st_sum[k];
    }
#endif
}

// Second pass of a K-split conv (small problems: fewer workgroups than the chip has CUs, e.g. the deep levels of the ONE
// 256x256 pair the reference runs, /root/reference/model/inference.py:29): one wave per (tile, cout tile, conv wave) adds the
// `ksplit` fp32 partial-sum slices in index order - deterministic, no atomics - on top of the BatchNorm shift, and then
// runs the conv kernel's OWN epilogue on the sums: ReLU, the blocked store, the fused MaxPool2d(2) copy (EPI_POOL) and the
// two-piece split of precision "bf16x2", all from one code path (rounds 1-5: a pixel-major slab, an element-wise
// finalize kernel and a separate max-pool launch).  The slab is read in the fragment order the conv wrote it in
// (conv_epilogue, EPI_SPLITK): 1 KiB contiguous per load instruction.
// Tried instead and NOT shipped (round 6): the reduction inside the conv kernel by the last workgroup of a tile to
// arrive (arrival counter, no waiting, one dispatch less per cut stage).  With agent-scope fences around the slab
// accesses (`buffer_wbl2` + `buffer_inv`: each workgroup writes back / drops its XCD's whole L2) it is correct - all
// GPU tests, 5 400 forwards bit-identical - and 1.2-1.6x SLOWER end to end (ONE 256x256 pair: bf16 0.295 -> 0.486 ms,
// fp32 0.913 -> 1.134); with write-through (sc0 sc1) stores instead of the release fence it is fast and WRONG (stale
// slices across XCDs: golden 135x240 fails).  The dispatch boundary is the cheapest coherence point this chip offers.
// One 64-thread workgroup per (tile, conv wave): the pass is bound by how many CUs pull slab bytes at once (a deep fp32
// level is 16 tiles x 16 slices x 64 KiB), not by arithmetic, and nothing in the epilogue crosses a wave.
template <typename T, int BN, int TH, int TW, int EPI, bool X2>
__global__ __launch_bounds__(64) void splitk_finalize_tile_kernel(const ConvArgs a)
{
    static_assert(EPI == EPI_PLAIN || EPI == EPI_POOL, "a fused head is never K-split");
    constexpr int WAVES_C = BN / 64;
    constexpr int NF = conv_wave_frags(BN, TH, TW);
    int t = blockIdx.x >> 2;                  // = the conv's logical tile index (its slab record)
    const int tile = t;
    const int ct = t % a.nct; t /= a.nct;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int lane = threadIdx.x & 63, wave = blockIdx.x & 3;   // the conv wave whose fragments this workgroup reduces
    const int l15 = lane & 15, lc = lane >> 4;
    const int wc = wave % WAVES_C, wp = wave / WAVES_C;
    const int ntile = a.B * a.tilesY * a.tilesX * a.nct;
    f32x4 acc[4][NF];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* p = reinterpret_cast<const float4*>(a.kslab) + (((size_t)tile * 4 + wave) * (4 * NF)) * 64 + lane;
    const size_t slice = (size_t)ntile * 4 * (4 * NF) * 64;
    // slices in chunks: the CH x NF loads of a cout tile are independent and issued together (the pass is a chain of
    // memory round trips, not bandwidth: one slice per trip took ~0.5 us per slice), the additions stay in slice order
    const int ks = a.ksplit;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4* pm = p + (size_t)m * NF * 64;
        constexpr int CH = NF == 8 ? 2 : 4;   // (128 accumulator registers leave room for two slices in flight, 64 for four)
        for (int s0 = 0; s0 < ks; s0 += CH, pm += CH * slice) {
            float4 q[CH][NF];
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (s0 + j < ks) {
#pragma unroll
                    for (int n = 0; n < NF; ++n) q[j][n] = pm[j * slice + n * 64];
                }
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (s0 + j < ks) {
#pragma unroll
                    for (int n = 0; n < NF; ++n) {
                        acc[m][n][0] += q[j][n].x; acc[m][n][1] += q[j][n].y; acc[m][n][2] += q[j][n].z; acc[m][n][3] += q[j][n].w;
                    }
                }
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {   // BatchNorm shift (the scale is folded into the weights)
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + ct * BN + wc * 64 + conv_cout_ofs<T>(m, lc));
#pragma unroll
        for (int n = 0; n < NF; ++n) {
            acc[m][n][0] += sh.x; acc[m][n][1] += sh.y; acc[m][n][2] += sh.z; acc[m][n][3] += sh.w;
        }
    }
    conv_epilogue<T, BN, TH, TW, EPI, X2>(a, acc, b, ty * TH, tx * TW, ct, 0, wc, wp, l15, lc);
}

}  // namespace fiunet
