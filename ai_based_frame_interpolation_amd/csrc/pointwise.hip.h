// pointwise.hip.h -- the HBM-bound kernels of the UNet forward, gfx950 only.
//
//   conv3x3_first_kernel   : the Cin = 2*cf (2 or 6) stem conv + BN + ReLU on the exact-fp32 MFMA
//                            (K = 18/54 padded to 20/56; the layer is write-bound)
//                            -- /root/reference/model/unet.py:72 (inc.double_conv.0..2)
//   maxpool2_kernel        : MaxPool2d(2), NHWC (ablation path; normally fused into the conv gather)
//                            -- unet.py:28
//   upcat_kernel           : Upsample(x2, bilinear, align_corners=True) + F.pad + cat([skip, up])
//                            (ablation path; normally fused) -- unet.py:46-54
//   head1x1_kernel         : OutConv 1x1 + bias, NHWC in -> fp32 NCHW out (ablation path; normally
//                            fused into the epilogue of up4.conv.double_conv.3) -- unet.py:60
//   nhwc_to_nchw_f32_kernel: parity-test readback of an intermediate (blocked) activation as NCHW
//   pre/postprocess kernels: model/inference.py:31-35 and :54-61 on device
#pragma once
#include "conv3x3_mfma.hip.h"

namespace fiunet {

// ---- "bf16x2" precision: helpers around the two-piece activation layout [hi planes | lo planes] of one image
//      (conv3x3_mfma.hip.h, SRC_DIRECT_X2); block_bytes = bytes from a hi record to its lo record = C/32 planes
__device__ __forceinline__ void x2_split_store(char* o, size_t block_bytes, const float (&v)[8])
{
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack_bf16x2_pk(v[2 * i], v[2 * i + 1]);
        l[i] = pack_bf16x2_pk(v[2 * i] - __uint_as_float(h[i] << 16), v[2 * i + 1] - __uint_as_float(h[i] & 0xffff0000u));
    }
    *reinterpret_cast<uint4*>(o) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(o + block_bytes) = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void x2_load(const char* p, size_t block_bytes, float (&v)[8])
{
    float hi[8], lo[8];
    chunk_unpack<__bf16>(ldg16(p), hi);
    chunk_unpack<__bf16>(ldg16(p + block_bytes), lo);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = hi[i] + lo[i];   // exact: the two pieces do not overlap
}

// Stem conv on the fp32 matrix cores (exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf
// chain).  One wave = one 16-pixel row segment x 64 couts per iteration:
//   A (weights)  [16 couts][4 k]  kept in registers for the whole kernel (4 cout tiles x KG k-groups)
//   B (patches)  [4 k][16 pixels] one global load per lane per k-group, k = (tap, input channel)
// The A rows are permuted so that lane (pixel, q) ends up with couts 16q..16q+15: its epilogue
// is 32 (bf16) / 64 (fp32) contiguous bytes and the 4 lanes of a pixel cover its whole 64-channel
// NHWC record.  K = 18 (gray) or 54 (RGB) is zero-padded to a multiple of 4.
// X2 (precision bf16x2): the exact-fp32 result is split into two bf16 pieces right here and written as the [hi | lo]
// tensor of 2 * 64 channels (T = bf16 is then only the element type of `dst`; no dither: this is the fp32-contract path).
template <typename T, int CF, bool X2 = false>
__global__ __launch_bounds__(256) void conv3x3_first_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ w,
    const float* __restrict__ scale, const float* __restrict__ shift, T* __restrict__ dst, int B,
    int H, int W, float dither)
{
    // dither: amplitude of the ordered input dither of the bf16 path (stem_dither in
    // conv3x3_mfma.hip.h: +d on frame 1, -d on frame 2, same values as the fused stem); 0 = off (fp32)
    constexpr int K = 9 * 2 * CF, KG = (K + 3) / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lc = lane >> 4;
    float a[4][KG];
    int koff[KG];  // per k-group: this lane's tap/channel, packed as (dy+1) | (dx+1)<<2 | ch<<4, or -1
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        const int k = 4 * g + lc;
        const int tap = k / (2 * CF), ci = k - tap * (2 * CF);
        koff[g] = k < K ? ((tap / 3) | ((tap % 3) << 2) | (ci << 4)) : -1;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int cout = (l15 >> 2) * 16 + ct * 4 + (l15 & 3);
            a[ct][g] = k < K ? w[k * 64 + cout] : 0.f;
        }
    }
    float sc[16], sh[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { sc[c] = scale[lc * 16 + c]; sh[c] = shift[lc * 16 + c]; }

    // work unit = a run of RUN consecutive 16-pixel tiles of one image row: (b, y) and the row
    // validity of every tap are fixed for the run, only x advances, so the per-tile address work is
    // one add per k-group; the loads of tile i+1 are issued before the MFMAs of tile i.
    constexpr int RUN = 8;
    const int tilesX = (W + 15) / 16, runsX = (tilesX + RUN - 1) / RUN;
    const long long nruns = (long long)B * H * runsX;
    const size_t plane = (size_t)H * W;
    for (long long rix = (long long)blockIdx.x * 4 + wave; rix < nruns; rix += (long long)gridDim.x * 4) {
        const int rx = (int)(rix % runsX);
        const long long r = rix / runsX;
        const int y = (int)(r % H), b = (int)(r / H);
        const int xt0 = rx * RUN, xt1 = min(xt0 + RUN, tilesX);
        const float* p[KG];   // address of this lane's tap/channel for pixel x = xt0*16 + l15
        int dxs[KG];          // its dx - 1, or a value that fails every x check when the row/k is invalid
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int ko = koff[g];
            const int yy = y + (ko & 3) - 1, ci = ko >> 4;
            const bool rowok = ko >= 0 && yy >= 0 && yy < H;
            // channel order of cat([frame1, frame2], dim=1)  (unet.py:109)
            const float* src = ci < CF ? f1 : f2;
            const int cc = ci < CF ? ci : ci - CF;
            dxs[g] = rowok ? ((ko >> 2) & 3) - 1 : -(1 << 28);
            p[g] = src + ((size_t)b * CF + (rowok ? cc : 0)) * plane + (size_t)(rowok ? yy : 0) * W +
                   (xt0 * 16 + l15);
        }
        auto load_tile = [&](int xt, float* v) __attribute__((always_inline)) {
            const int x = xt * 16 + l15;
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                const int xx = x + dxs[g];
                v[g] = (xx >= 0 && xx < W) ? p[g][(xt - xt0) * 16 + dxs[g]] : 0.f;
                if constexpr (sizeof(T) == 2 && !X2) {  // the conv's zero padding stays exactly zero
                    // (row and frame of this k-slot are re-derived from koff through an opaque copy of y, so
                    // that hipcc does not hoist one more register per k-group out of the tile loop: the RGB
                    // instantiation would drop to one wave per SIMD)
                    int yo = y;
                    asm volatile("" : "+v"(yo));
                    if (xx >= 0 && xx < W)
                        v[g] = v[g] + ((koff[g] >> 4) >= CF ? -dither : dither) * stem_dither(yo + (koff[g] & 3) - 1, xx);
                }
            }
        };
        float vn[KG];
        load_tile(xt0, vn);
        for (int xt = xt0; xt < xt1; ++xt) {
            float v[KG];
#pragma unroll
            for (int g = 0; g < KG; ++g) v[g] = vn[g];
            if (xt + 1 < xt1) load_tile(xt + 1, vn);
            const int x = xt * 16 + l15;
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct][g], v[g], acc[ct], 0, 0, 0);
            float o[16];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    o[ct * 4 + j] = fmaxf(fmaf(acc[ct][j], sc[ct * 4 + j], sh[ct * 4 + j]), 0.f);
            if constexpr (X2) {
                if (x < W) {   // couts lc*16 .. +15 = two 16-B chunks of plane lc / 2; blocks of 2 planes per image
                    const size_t blk = (size_t)2 * H * W * 64;
                    char* op = (char*)dst + (size_t)b * 2 * blk + blk_off(lc >> 1, y, x, H, W) + (size_t)(lc & 1) * 32;
                    float c0[8], c1[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) { c0[c] = o[c]; c1[c] = o[8 + c]; }
                    x2_split_store(op, blk, c0);
                    x2_split_store(op + 16, blk, c1);
                }
            } else if (x < W) {
                // couts lc*16 .. lc*16+15 of this pixel: half a plane record (bf16) / one plane (fp32)
                constexpr int NE = Elem<T>::NE, PL = Elem<T>::PL;
                char* op = (char*)dst + (size_t)b * H * W * 64 * sizeof(T) +
                           blk_off((lc * 16) / PL, y, x, H, W) + (size_t)((lc * 16) % PL) * sizeof(T);
#pragma unroll
                for (int c = 0; c < 16; c += NE)
                    *reinterpret_cast<uint4*>(op + c * sizeof(T)) = chunk_pack<T>(o + c);
            }
        }
    }
}

// The RGB (6 -> 64) stem of the bf16 network on the bf16 matrix cores with hi + lo split operands (round 4).
// conv3x3_first_kernel<bf16, 3> above evaluates it on the exact-fp32 MFMA: K = 54 -> 56 fp32 MFMAs of 32 cycles per
// 16 pixels x 64 couts, with one global load per lane and k-group in front of them - 2.97 ms per B=8 1080p forward,
// 7x the time its 2.1 GB of output take to write, and the longest stage of the RGB network.  Here the raw patch of
// both frames is staged once per 16x32 tile in LDS as bf16 pairs {frame1, frame2} per colour channel, hi and lo part
// (x = xh + xl, |xl| <= 2^-9 |x|), the weights (BatchNorm scale folded in, shift riding in a bias k-slot whose data
// operand is 1.0) are split the same way and kept in registers, and a 16-pixel fragment is
//     acc += wh*xh + wh*xl + wl*xh        (fp32 accumulate; the dropped wl*xl term is 2^-16 relative)
// over three 32-slot k-chunks: chunk = the tap column dx, lane group = the tap row dy, its 8 slots = 3 colours x 2 frames
// (+ 2 empty) of patch pixel (r + dy, x + dx); lane group 3 carries the bias in chunk 0 and is empty otherwise - 9 bf16
// MFMAs of 16 cycles per cout tile instead of 14 fp32 ones of 32.  Same scheme, and the same
// ~2^-16 relative accuracy before the bf16 rounding of the output, as the fused gray stem (conv3x3_mfma.hip.h,
// SRC_STEM).  /root/reference/model/unet.py:72 with n_channels = 6 (unet.py:66); channel order of torch.cat([f1, f2]).
#ifndef FIUNET_RGB_STEM_OCC
#define FIUNET_RGB_STEM_OCC 2
#endif
// X2 (precision bf16x2): `dst` is the two-piece tensor [hi planes | lo planes] of 2 x 64 channels - relu(acc) split in fp32
// (the split-bf16 MFMAs' ~2^-16 relative accuracy is that precision's own class; no dither: the caller passes 0)
template <bool X2 = false>
__global__ __launch_bounds__(256, FIUNET_RGB_STEM_OCC) void stem_rgb_split_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, const float* __restrict__ w,  // w: [9 taps][6][64] fp32
    const float* __restrict__ scale, const float* __restrict__ shift, __bf16* __restrict__ dst, int B, int H, int W,
    float dither, const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2)
{
    // u1 / u2 != nullptr (fiunet_forward_u8): the uint8 frames [B][3][H][W] are read and normalised right here
    // (preprocess_u8_value: the same fp32 values fiunet_preprocess_u8 would have written), f1 / f2 are unused
    constexpr int CF = 3, TH = 16, TW = 32, PH = TH + 2, PW = TW + 4;
    // Patch image (round 5): PIXEL-major, one 16-B record per patch pixel = {c0, c1, c2, 0} dwords, each dword the bf16 pair
    // {frame1, frame2} of that colour; a hi image and a lo image per buffer.  MFMA k-chunk = the tap COLUMN dx, lane group
    // lc = the tap row dy (k-slot lc*8 + 2*colour + frame; slots 6, 7 carry zero weights; lane group 3 = the bias slot): the
    // lane of output pixel (r, x) reads ONE aligned 16-B record at patch (r + lc, x + dx) per operand and chunk - 6
    // ds_read_b128 per fragment.  (Round 4's colour-major images made a lane's 8 k-slots 4 consecutive columns of one colour:
    // a window sliding by one dword per lane, i.e. 24 unaligned ds_read_b32 / ds_read2_b32 per fragment and 6 ds_write_b32 per
    // staged pixel instead of 2 ds_write_b128.)
    constexpr int PIX = PH * PW;                 // records of one (hi | lo) patch image
    constexpr int BUF = 2 * PIX;                 // one patch buffer in records: hi image, lo image; there are two (double-buffered)
    constexpr int BIAS_REC = 2 * BUF;            // record {1.0 | 0, 0, 0, 0}: hi operand of the bias k-slot (lane group 3, chunk 0)
    constexpr int ZERO_REC = BIAS_REC + 1;       // zero record: its lo operand, and both operands of lane group 3 in chunks 1, 2
    __shared__ uint4 pd[ZERO_REC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lc = lane >> 4;

    // ---- A operands: this lane's row of every cout tile, the 8 k-slots of its lane group (= tap row dy), per chunk dx.
    //      Packed row lc'*4 + j of tile ct <-> cout (ct >> 1)*32 + lc'*8 + (ct & 1)*4 + j: an output lane (pixel, lc) then holds
    //      8 consecutive couts of plane 0 (tiles 0, 1) and 8 of plane 1 (tiles 2, 3) = the 16-B quarter lc of both 64-B plane
    //      records, so that ONE store instruction writes whole records (1 KiB contiguous per plane), as the conv kernels do
    //      (round 4 gave a lane 16 consecutive couts: two 16-B stores per lane, each covering half of every record it touched).
    uint4 ah[3][4], al[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int cout = (ct >> 1) * 32 + (l15 >> 2) * 8 + (ct & 1) * 4 + (l15 & 3);
            const float sc = scale[cout];
            unsigned h[4], l[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {        // slots 2c (frame 1), 2c + 1 (frame 2) of colour c
                float v0 = 0.f, v1 = 0.f;
                if (lc < 3 && c < 3) {
                    const int tap = lc * 3 + dx;
                    v0 = w[((tap * 2 * CF) + c) * 64 + cout] * sc;
                    v1 = w[((tap * 2 * CF) + CF + c) * 64 + cout] * sc;
                } else if (lc == 3 && dx == 0 && c == 0) {
                    v0 = shift[cout];
                }
                h[c] = pack_bf16x2_pk(v0, v1);
                l[c] = pack_bf16x2_pk(v0 - __uint_as_float(h[c] << 16), v1 - __uint_as_float(h[c] & 0xffff0000u));
            }
            ah[dx][ct] = make_uint4(h[0], h[1], h[2], h[3]);
            al[dx][ct] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    }
    if (tid == 0) { pd[BIAS_REC] = make_uint4(0x00003f80u, 0u, 0u, 0u); pd[ZERO_REC] = make_uint4(0u, 0u, 0u, 0u); }
    // this lane's record index within a patch image for fragment (r = 0, half 0), chunk dx = 0; lane group 3 does not move
    const int lane_rec = lc < 3 ? lc * PW + l15 : 0;
    const int rec_mul = lc < 3 ? 1 : 0;

    const int tilesX = (W + TW - 1) / TW, tilesY = (H + TH - 1) / TH;
    const long long ntiles = (long long)B * tilesX * tilesY;
    const size_t plane = (size_t)H * W;
    // The raw patch of the NEXT tile is requested (into registers) before this tile's fragments run and converted /
    // split / stored into the other of two LDS patch buffers after them: the global round trip of a tile's 18 loads per
    // thread hides under the previous tile's MFMAs, and a tile costs one barrier (round 4 staged and multiplied in turn).
    constexpr int NI = (PIX + 255) / 256;        // patch pixels per thread
    float r0[NI][CF], r1[NI][CF], dth[NI];
    unsigned okm = 0;
    auto fetch = [&](long long t) __attribute__((always_inline)) {
        const int tx = (int)(t % tilesX);
        const long long q = t / tilesX;
        const int ty = (int)(q % tilesY), b = (int)(q / tilesY);
        const int y0 = ty * TH, x0 = tx * TW;
        okm = 0;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = tid + k * 256;
            const int py = i / PW, px = i - py * PW;
            const int y = y0 - 1 + py, x = x0 - 1 + px;
            const bool ok = (i < PIX) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
            okm |= (ok ? 1u : 0u) << k;
            const size_t at = (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
            dth[k] = dither * stem_dither(y, x);
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                const size_t idx = ((size_t)b * CF + c) * plane + at;
                if (u1) { r0[k][c] = (float)u1[idx]; r1[k][c] = (float)u2[idx]; }   // wave-uniform; converted in stash()
                else { r0[k][c] = f1[idx]; r1[k][c] = f2[idx]; }
            }
        }
    };
    auto stash = [&](uint4* dstb) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int i = tid + k * 256;
            const bool ok = (okm >> k) & 1u;
            unsigned h[CF], l[CF];
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                float v0 = r0[k][c], v1 = r1[k][c];
                if (u1) { v0 = preprocess_u8_value((unsigned char)v0); v1 = preprocess_u8_value((unsigned char)v1); }
                v0 = ok ? v0 + dth[k] : 0.f;   // +d on frame 1, -d on frame 2; the conv's zero padding stays exactly zero
                v1 = ok ? v1 - dth[k] : 0.f;
                h[c] = pack_bf16x2_pk(v0, v1);
                l[c] = pack_bf16x2_pk(v0 - __uint_as_float(h[c] << 16), v1 - __uint_as_float(h[c] & 0xffff0000u));
            }
            if (i < PIX) {
                dstb[i] = make_uint4(h[0], h[1], h[2], 0u);
                dstb[PIX + i] = make_uint4(l[0], l[1], l[2], 0u);
            }
        }
    };
    long long t = blockIdx.x;
    if (t < ntiles) fetch(t);
    int buf = 0;
    for (; t < ntiles; t += gridDim.x, buf ^= 1) {
        const int tx = (int)(t % tilesX);
        const long long q = t / tilesX;
        const int ty = (int)(q % tilesY), b = (int)(q / tilesY);
        const int y0 = ty * TH, x0 = tx * TW;
        uint4* const pb = pd + buf * BUF;
        stash(pb);
        __syncthreads();   // this tile's patch is complete; every wave is past the fragments of the tile before last (same buffer)
        if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
        // lane group 3: hi operand = the bias record in chunk 0, zeros otherwise; lo operand = zeros
        const uint4* const ph0 = lc < 3 ? pb + lane_rec : pd + BIAS_REC;
        const uint4* const pl0 = lc < 3 ? pb + PIX + lane_rec : pd + ZERO_REC;
        const uint4* const phz = lc < 3 ? pb + lane_rec : pd + ZERO_REC;
        // ---- fragments: wave = 4 tile rows x 2 column halves
#pragma unroll 2
        for (int f = 0; f < 8; ++f) {
            const int r = wave * 4 + (f >> 1), xc = (f & 1) * 16 + l15;
            const int y = y0 + r, x = x0 + xc;
            const int o = rec_mul * (r * PW + (f & 1) * 16);
            f32x4 acc[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const uint4 bh = (dx == 0 ? ph0 : phz)[o + rec_mul * dx];
                const uint4 bl = pl0[o + rec_mul * dx];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) mma_chunk<__bf16>(acc[ct], al[dx][ct], bh);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) mma_chunk<__bf16>(acc[ct], ah[dx][ct], bl);
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) mma_chunk<__bf16>(acc[ct], ah[dx][ct], bh);
            }
            if (y < H && x < W) {
                // o16[0..7] = couts lc*8 .. +7 (plane 0), o16[8..15] = couts 32 + lc*8 .. +7 (plane 1): quarter lc of both records
                char* op = (char*)dst + (size_t)b * H * W * 64 * 2 + blk_off(0, y, x, H, W) + lc * 16;
                const size_t plane_bytes = (size_t)H * W * 64;
                float o16[16];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o16[ct * 4 + j] = fmaxf(acc[ct][j], 0.f);
                if constexpr (X2) {
                    const size_t blk = (size_t)2 * H * W * 64;   // bytes from a hi record to its lo record (2 planes per piece)
                    char* o2 = (char*)dst + (size_t)b * 2 * blk + blk_off(0, y, x, H, W) + lc * 16;
                    float c0[8], c1[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) { c0[c] = o16[c]; c1[c] = o16[8 + c]; }
                    x2_split_store(o2, blk, c0);
                    x2_split_store(o2 + plane_bytes, blk, c1);
                } else {
                    *reinterpret_cast<uint4*>(op) = chunk_pack<__bf16>(o16);
                    *reinterpret_cast<uint4*>(op + plane_bytes) = chunk_pack<__bf16>(o16 + 8);
                }
            }
        }
    }
}

// ConvTranspose2d(Cin, Cout = Cin / 2, kernel 2, stride 2) + bias of the bilinear=False decoder
// (/root/reference/model/unet.py:42-44,47) followed by F.pad to the skip tensor's size (unet.py:49-53), written as the
// full-resolution blocked tensor that the next conv gathers as its second source (like the materialised bilinear
// upsample).  With stride = kernel the four taps do not overlap: out[co][2y + dy][2x + dx] = bias[co] +
// sum_ci x[ci][y][x] * W[ci][co][dy][dx], i.e. four pointwise GEMMs sharing their B operand.  One wave = 32 consecutive
// low-res pixels of a row x 64 couts x 4 taps (32 accumulator tiles); both operands come straight from global memory
// in MFMA fragment order (the blocked layouts already are: a lane's k-slots are one 16-B chunk of a plane record), no
// LDS.  Not a tuned kernel: the variant is not on any benchmark line (no reference caller constructs it).
struct ConvTArgs {
    const void* low;     // [B][Cin/PL][lowH][lowW][PL]
    const void* wgt;     // [4 taps = dy*2+dx][Cin/PL][Cout][PL]
    const float* bias;   // [Cout]
    void* dst;           // [B][Cout/PL][H][W][PL]; rows / columns outside the 2x extent are zeroed by the caller
    int B, H, W, lowH, lowW, Cin, Cout;
    int padT, padL;      // F.pad top / left, whole-image values
    int upOffY, lowOffY, lowHg;   // row band of a taller image (fiunet_forward_strip); un-tiled: 0, 0, lowH
};

// X2 (precision bf16x2): `low` and `dst` are two-piece tensors [hi planes | lo planes], the weights [wh | wl] (both
// pieces [4][Cin/32][Cout][32]); a product is wh*xh + wl*xh + wh*xl, the bias is added in fp32 and the sum split again.
template <typename T, bool X2 = false>
__global__ __launch_bounds__(256) void convt2x2_kernel(const ConvTArgs a)
{
    static_assert(!X2 || sizeof(T) == 2, "two-piece operands are bf16");
    constexpr int PL = Elem<T>::PL;
    constexpr int NPX = 2;   // 16-pixel fragments per wave: every weight fragment is used for 32 pixels
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lc = lane >> 4;
    const int xt = (a.lowW + 16 * NPX - 1) / (16 * NPX), cgs = a.Cout / 64, nplanes = a.Cin / PL;
    const long long units = (long long)a.B * a.lowH * xt * cgs;
    const size_t low_plane = (size_t)a.lowH * a.lowW * 64, out_plane = (size_t)a.H * a.W * 64;
    for (long long u = (long long)blockIdx.x * 4 + wave; u < units; u += (long long)gridDim.x * 4) {
        const int cg = (int)(u % cgs);
        long long r = u / cgs;
        const int tx = (int)(r % xt); r /= xt;
        const int y = (int)(r % a.lowH), b = (int)(r / a.lowH);
        const char* bsrc[NPX];
#pragma unroll
        for (int f = 0; f < NPX; ++f) {
            const int xl = min(tx * 16 * NPX + f * 16 + l15, a.lowW - 1);
            bsrc[f] = (const char*)a.low + (size_t)b * nplanes * low_plane * (X2 ? 2 : 1) + ((size_t)y * a.lowW + xl) * 64 + lc * 16;
        }
        const char* wsrc = (const char*)a.wgt + ((size_t)(cg * 64 + l15)) * 64 + lc * 16;
        f32x4 acc[4][4][NPX];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int f = 0; f < NPX; ++f) acc[t][ct][f] = f32x4{0.f, 0.f, 0.f, 0.f};
        const size_t wpiece = (size_t)4 * nplanes * a.Cout * 64;   // X2: bytes from wh to wl
        for (int p = 0; p < nplanes; ++p) {
            uint4 xb[NPX], xl[X2 ? NPX : 1];
#pragma unroll
            for (int f = 0; f < NPX; ++f) {
                xb[f] = *reinterpret_cast<const uint4*>(bsrc[f] + (size_t)p * low_plane);
                if constexpr (X2) xl[f] = *reinterpret_cast<const uint4*>(bsrc[f] + (size_t)(nplanes + p) * low_plane);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const char* wp = wsrc + (((size_t)t * nplanes + p) * a.Cout + ct * 16) * 64;
                    const uint4 wa = *reinterpret_cast<const uint4*>(wp);
#pragma unroll
                    for (int f = 0; f < NPX; ++f) mma_chunk<T>(acc[t][ct][f], wa, xb[f]);
                    if constexpr (X2) {
                        const uint4 wl = *reinterpret_cast<const uint4*>(wp + wpiece);
#pragma unroll
                        for (int f = 0; f < NPX; ++f) {
                            mma_chunk<T>(acc[t][ct][f], wl, xb[f]);
                            mma_chunk<T>(acc[t][ct][f], wa, xl[f]);
                        }
                    }
                }
        }
        // global row of this low-res row, and where its two output rows land in this band
        const int yg = y + a.lowOffY;
#pragma unroll
        for (int f = 0; f < NPX; ++f) {
            const int x = tx * 16 * NPX + f * 16 + l15;
            if (x >= a.lowW || yg >= a.lowHg) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int Y = 2 * yg + (t >> 1) + a.padT - a.upOffY, X = 2 * x + (t & 1) + a.padL;
                if (Y < 0 || Y >= a.H || X < 0 || X >= a.W) continue;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    const int co = cg * 64 + ct * 16 + lc * 4;   // MFMA row lc*4 + j of tile ct, natural cout order
                    const float4 bi = *reinterpret_cast<const float4*>(a.bias + co);
                    const f32x4 q = acc[t][ct][f];
                    const float v[4] = {q[0] + bi.x, q[1] + bi.y, q[2] + bi.z, q[3] + bi.w};
                    char* op = (char*)a.dst + (size_t)b * (a.Cout / PL) * out_plane * (X2 ? 2 : 1) + blk_off(co / PL, Y, X, a.H, a.W) +
                               (size_t)(co % PL) * sizeof(T);
                    if constexpr (X2) {
                        const unsigned h0 = pack_bf16x2_pk(v[0], v[1]), h1 = pack_bf16x2_pk(v[2], v[3]);
                        *reinterpret_cast<uint2*>(op) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2*>(op + (size_t)(a.Cout / PL) * out_plane) = make_uint2(
                            pack_bf16x2_pk(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u)),
                            pack_bf16x2_pk(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u)));
                    } else if constexpr (sizeof(T) == 2)
                        *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    else
                        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
    }
}

// thread = (plane, output pixel, 16-byte chunk); blocked layout in and out
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ src,
                                                       T* __restrict__ dst, int B, int H, int W,
                                                       int C)  // H,W = input size
{
    const int Ho = H / 2, Wo = W / 2, P = C / Elem<T>::PL;
    const size_t total = (size_t)B * P * Ho * Wo * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i & 3);
        size_t p = i >> 2;
        const int x = (int)(p % Wo); p /= Wo;
        const int y = (int)(p % Ho); p /= Ho;  // p = b * P + plane
        const char* s = (const char*)src + (p * H + 2 * y) * (size_t)W * 64 + (size_t)(2 * x) * 64 + ch * 16;
        const size_t rowb = (size_t)W * 64;
        const uint4 r = chunk_max4<T>(ldg16(s), ldg16(s + 64), ldg16(s + rowb), ldg16(s + rowb + 64));
        *reinterpret_cast<uint4*>((char*)dst + i * 16) = r;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void upcat_kernel(const ConvArgs a, T* __restrict__ dst)
{
    const int P = (a.C0 + a.C1) / Elem<T>::PL;
    const size_t total = (size_t)a.B * P * a.H * a.W * 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ch = (int)(i & 3);
        size_t p = i >> 2;
        const int x = (int)(p % a.W); p /= a.W;
        const int y = (int)(p % a.H); p /= a.H;
        const int plane = (int)(p % P);
        const int b = (int)(p / P);
        const uint4 v = gather_chunk<T, SRC_CONCAT_UP>(a, b, y, x, plane, ch);
        *reinterpret_cast<uint4*>((char*)dst + i * 16) = v;
    }
}

// The upsampled + padded half of a concat input alone, [B][C1/PL][H][W][PL] (same values as the
// fused gather: chunk_bilerp).  Used when several cout tiles would each interpolate the same tile.
constexpr int UPS_ROWS = 8;  // output rows per thread of upsample_kernel
template <typename T>
__global__ __launch_bounds__(256) void upsample_kernel(const ConvArgs a, T* __restrict__ dst)
{
    // grid = (chunks of a row / 256, H / UPS_ROWS, B * planes).  A thread owns one (pixel column,
    // 16-B chunk) and walks UPS_ROWS output rows down it; the row is the same in the whole block,
    // so the vertical mapping is wave-uniform and the horizontally interpolated source rows are
    // carried from one output row to the next, exactly like the fused gather's column walk
    // (chunk_hlerp + chunk_vlerp: the same bits as chunk_bilerp).
    constexpr int NE = Elem<T>::NE;
    const int P1 = a.C1 / Elem<T>::PL;
    const int i = blockIdx.x * 256 + threadIdx.x;  // (x, 16-B chunk) within the row
    if (i >= a.W * 4) return;
    const int b = blockIdx.z / P1, plane = blockIdx.z - b * P1;
    const UpAxis ux = up_axis_x(a, i >> 2);
    const size_t low_row = (size_t)a.lowW * 64;
    const char* const col = (const char*)a.src1 + (size_t)b * a.lowH * a.lowW * a.C1 * sizeof(T) +
                            (size_t)plane * a.lowH * low_row + (i & 3) * 16;
    const char* const s0 = col + (size_t)ux.i0 * 64;
    const char* const s1 = col + (size_t)ux.i1 * 64;
    char* const out = (char*)dst + ((size_t)blockIdx.z * a.H * a.W * 4 + i) * 16;
    float h0[NE], h1[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) h0[k] = h1[k] = 0.f;
    int c0 = -1, c1 = -1;  // source rows currently held in h0 / h1 (wave-uniform)
    const int ybeg = blockIdx.y * UPS_ROWS, yend = min(a.H, ybeg + UPS_ROWS);
    for (int y = ybeg; y < yend; ++y) {
        const UpAxis uy = up_axis_y(a, y);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (uy.ok) {
            if (uy.i0 != c0) {
                if (uy.i0 == c1) {
#pragma unroll
                    for (int k = 0; k < NE; ++k) h0[k] = h1[k];
                } else {
                    chunk_hlerp<T>(ldg16(s0 + uy.i0 * low_row), ldg16(s1 + uy.i0 * low_row), ux.h, ux.l, h0);
                }
                c0 = uy.i0;
            }
            if (uy.i1 != c1) {
                if (uy.i1 == c0) {
#pragma unroll
                    for (int k = 0; k < NE; ++k) h1[k] = h0[k];
                } else {
                    chunk_hlerp<T>(ldg16(s0 + uy.i1 * low_row), ldg16(s1 + uy.i1 * low_row), ux.h, ux.l, h1);
                }
                c1 = uy.i1;
            }
            v = chunk_vlerp<T>(h0, h1, uy.h, uy.l);
            if (!ux.ok) v = make_uint4(0u, 0u, 0u, 0u);
        }
        *reinterpret_cast<uint4*>(out + (size_t)y * a.W * 64) = v;
    }
}

// Upsample(x2, bilinear, align_corners=True) + F.pad of a two-piece tensor: the fp32 value hi + lo of the four
// neighbours, aten's association in fp32 (chunk_bilerp's), split again.  a.src1 = low-res [B][2 * C1/32][lowH][lowW][32],
// dst = [B][2 * C1/32][H][W][32]; a.C1 = the REAL channel count.  Same work split as upsample_kernel: grid =
// (chunks of a row / 256, H / UPS_ROWS, B * real planes), a thread owns one (pixel column, 16-B chunk) and walks
// UPS_ROWS output rows down it, carrying the horizontally interpolated source rows from one output row to the next.
__global__ __launch_bounds__(256) void x2_upsample_kernel(const ConvArgs a, char* __restrict__ dst)
{
    const int np = a.C1 / 32;
    const int i = blockIdx.x * 256 + threadIdx.x;  // (x, 16-B chunk) within the row
    if (i >= a.W * 4) return;
    const int b = blockIdx.z / np, plane = blockIdx.z - b * np;
    const UpAxis ux = up_axis_x(a, i >> 2);
    const size_t low_row = (size_t)a.lowW * 64;
    const size_t HW = (size_t)a.H * a.W, lHW = (size_t)a.lowH * a.lowW;
    const size_t blk = (size_t)np * HW * 64, lblk = (size_t)np * lHW * 64;
    const char* const col = (const char*)a.src1 + (size_t)b * 2 * lblk + (size_t)plane * lHW * 64 + (i & 3) * 16;
    const char* const s0 = col + (size_t)ux.i0 * 64;
    const char* const s1 = col + (size_t)ux.i1 * 64;
    char* const out = dst + (size_t)b * 2 * blk + ((size_t)plane * HW * 4 + i) * 16;
    auto hl = [&](int row, float (&h)[8]) __attribute__((always_inline)) {
        float va[8], vb[8];
        x2_load(s0 + (size_t)row * low_row, lblk, va);
        x2_load(s1 + (size_t)row * low_row, lblk, vb);
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = fmaf(ux.l, vb[k], __fmul_rn(ux.h, va[k]));
    };
    float h0[8], h1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h0[k] = h1[k] = 0.f;
    int c0 = -1, c1 = -1;  // source rows currently held in h0 / h1 (wave-uniform)
    const int ybeg = blockIdx.y * UPS_ROWS, yend = min(a.H, ybeg + UPS_ROWS);
    for (int y = ybeg; y < yend; ++y) {
        const UpAxis uy = up_axis_y(a, y);
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = 0.f;
        if (uy.ok) {
            if (uy.i0 != c0) {
                if (uy.i0 == c1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) h0[k] = h1[k];
                } else {
                    hl(uy.i0, h0);
                }
                c0 = uy.i0;
            }
            if (uy.i1 != c1) {
                if (uy.i1 == c0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) h1[k] = h0[k];
                } else {
                    hl(uy.i1, h1);
                }
                c1 = uy.i1;
            }
            if (ux.ok) {
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = fmaf(uy.l, h1[k], __fmul_rn(uy.h, h0[k]));
            }
        }
        x2_split_store(out + (size_t)y * a.W * 64, blk, o);
    }
}

// thread = pixel; reads 64 channels, writes nc fp32 planes
template <typename T>
__global__ __launch_bounds__(256) void head1x1_kernel(const T* __restrict__ src,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ out, int B, int H, int W,
                                                      int nc)
{
    const size_t HW = (size_t)H * W, total = (size_t)B * HW;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    constexpr int NE = Elem<T>::NE;
    float acc[3] = {0.f, 0.f, 0.f};
    constexpr int PL = Elem<T>::PL;
    const size_t bb = i / HW, pp = i - bb * HW;
    const char* s = (const char*)src + bb * HW * 64 * sizeof(T) + pp * 64;  // plane 0 record
#pragma unroll
    for (int c = 0; c < 64; c += NE) {
        float f[NE];
        chunk_unpack<T>(ldg16(s + (size_t)(c / PL) * HW * 64 + (c % PL) * sizeof(T)), f);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < nc) {
#pragma unroll
                for (int j = 0; j < NE; ++j) acc[k] = fmaf(f[j], w[k * 64 + c + j], acc[k]);
            }
    }
    const size_t b = i / HW, p = i - b * HW;
    for (int k = 0; k < nc; ++k) out[(b * nc + k) * HW + p] = acc[k] + bias[k];
}

template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const T* __restrict__ src,
                                                               float* __restrict__ dst, int B,
                                                               int C, int H, int W)
{
    const size_t total = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t p = i;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H); p /= H;
        const int c = (int)(p % C);
        const int b = (int)(p / C);
        constexpr int PL = Elem<T>::PL;
        dst[i] = (float)src[((((size_t)b * (C / PL) + c / PL) * H + y) * W + x) * PL + c % PL];
    }
}

// read-back of a two-piece activation (precision bf16x2): hi + lo of [B][2 * C/32][H][W][32] as fp32 NCHW
__global__ __launch_bounds__(256) void x2_to_nchw_f32_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, int B,
                                                             int C, int H, int W)
{
    const size_t total = (size_t)B * C * H * W;
    const size_t piece = (size_t)(C / 32) * H * W * 32;   // elements from a hi value to its lo value
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t p = i;
        const int x = (int)(p % W); p /= W;
        const int y = (int)(p % H); p /= H;
        const int c = (int)(p % C);
        const int b = (int)(p / C);
        const size_t at = (size_t)b * 2 * piece + ((((size_t)(c / 32)) * H + y) * W + x) * 32 + c % 32;
        dst[i] = (float)src[at] + (float)src[at + piece];
    }
}

// Weights of precision bf16x2, built on the device from the packed fp32 copy (BatchNorm scale already folded in):
// w32 [cin/16][9 slots][cout][16] (rows in natural cout order) -> out [2 pieces][cin/32][9][cout][32] bf16 with the bf16
// kernels' row permutation (row R holds cout (R & ~31) + ((R & 15) >> 2) * 8 + ((R >> 4) & 1) * 4 + (R & 3), fiunet.hip
// bf16_row_to_cout); piece 0 = RNE bf16 of w, piece 1 = RNE bf16 of the remainder (w = wh + wl to 2^-17 relative).
// taps = 9 (3x3 convs) or 4 (ConvTranspose2d 2x2, w32 [4][cin/16][cout][16] -> out [2][4][cin/32][cout][32], natural rows)
__global__ __launch_bounds__(256) void x2_pack_weights_kernel(const float* __restrict__ w32, unsigned short* __restrict__ out,
                                                              int cin, int cout, int convt)
{
    const int taps = convt ? 4 : 9;
    const size_t half = (size_t)cin * taps * cout, total = 2 * half;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int piece = i >= half;
        size_t r = piece ? i - half : i;
        const int k = (int)(r % 32); r /= 32;
        const int R = (int)(r % cout); r /= cout;
        int plane, slot;
        if (convt) { plane = (int)(r % (cin / 32)); slot = (int)(r / (cin / 32)); }
        else { slot = (int)(r % 9); plane = (int)(r / 9); }
        const int ci = plane * 32 + k;
        const int co = convt ? R : (R & ~31) + ((R & 15) >> 2) * 8 + ((R >> 4) & 1) * 4 + (R & 3);
        const size_t src = convt ? (((size_t)slot * (cin / 16) + ci / 16) * cout + co) * 16 + ci % 16
                                 : (((size_t)(ci / 16) * 9 + slot) * cout + co) * 16 + ci % 16;
        const float w = w32[src];
        const __bf16 hi = (__bf16)w;
        const __bf16 v = piece ? (__bf16)(w - (float)hi) : hi;
        out[i] = __builtin_bit_cast(unsigned short, v);
    }
}

// model/inference.py:31-35: image.astype(float32) / 255.0 ; 2.0 * image - 1.0
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t* __restrict__ in,
                                                            float* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        out[i] = preprocess_u8_value(in[i]);
    }
}

// model/inference.py:54-61: (x + 1) / 2 ; clamp(0, 1) ; (x * 255).astype(uint8) -- truncation
__global__ __launch_bounds__(256) void postprocess_u8_kernel(const float* __restrict__ in,
                                                             uint8_t* __restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        out[i] = postprocess_u8_value(in[i]);
    }
}

}  // namespace fiunet
