// conv3x3_pers.hip.h -- persistent, fully double-buffered variant of the fused 3x3 conv
// (conv3x3_mfma.hip.h explains the GEMM mapping, LDS image and gather modes; this file reuses
// its helpers).  Replaces the same reference ops: /root/reference/model/unet.py:11-18, :28, :46-54.
//
// What is different from conv3x3_mfma_kernel:
//   * ONE workgroup of 8 waves (512 threads) per CU owns the whole 160 KiB of LDS:
//       in-tile   2 x (18 x 40 pixels x 64 B)  = 92 160 B   (16x32 output pixels + halo, 2-deep)
//       weights   2 x (3 taps x BN x 64 B)     <= 49 152 B  (2-deep ring, one (plane,ky) step each)
//     so BOTH operand streams run one stage ahead of the MFMAs and a step ends with a single
//     barrier: no exposed gather at plane boundaries.
//   * The workgroup is persistent: it walks its list of (spatial tile, cout tile) items as one
//     continuous stream of planes; the first in-tile and weight slice of the next item are
//     prefetched during the last plane of the current one, and the epilogue stores of an item
//     drain under the next item's MFMAs.
//   * Pooled / bilinearly-upsampled planes (register path) are gathered for plane q+1 while
//     plane q is being multiplied: each of the 3 steps of a plane issues the loads of a batch
//     of chunks before its MFMAs and commits them to the idle in-tile buffer afterwards.
//   * Items are dealt so that the 32 CUs of one XCD work on consecutive items at the same time:
//     the cout tiles of one spatial tile (same input tile, different weights) and neighbouring
//     tiles (shared halo) hit the same L2.
//   * Wave tiling: BN=128 -> 2 cout halves x 4 pixel groups, 64 couts x 128 pixels per wave;
//     BN=64  -> 8 pixel groups, 64 couts x 64 pixels per wave.
#pragma once
#include "conv3x3_mfma.hip.h"

namespace fiunet {

template <int BN> struct PersTile {
    static constexpr int TH = 16, TW = 32;
    static constexpr int TWP = 40, THP = TH + 2;
    static constexpr int IN_BYTES = THP * TWP * 64;
    static constexpr int W_BYTES = 3 * BN * 64;
    static constexpr int LDS_BYTES = 2 * IN_BYTES + 2 * W_BYTES;
};

__device__ __forceinline__ void wait_all_and_barrier()
{
    // LDS-DMA pieces (asm-issued, invisible to hipcc) + ds_writes of this wave, then barrier
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

struct ItemPos {
    int b, y0, x0, ct;
};

template <typename T, int BN, int MODE, bool HEAD>
__global__ __launch_bounds__(512, 2) void conv3x3_pers_kernel(const ConvArgs a)
{
    using Tile = PersTile<BN>;
    constexpr int PL = Elem<T>::PL;
    constexpr int TH = Tile::TH, TW = Tile::TW, TWP = Tile::TWP, THP = Tile::THP;
    constexpr int WAVES_C = BN / 64, WAVES_P = 8 / WAVES_C;
    constexpr int ROWS_W = TH / WAVES_P;  // tile rows per wave: 4 (BN=128) or 2 (BN=64)
    constexpr int NT = ROWS_W * 2;        // 16-pixel fragments per wave
    static_assert(!HEAD || BN == 64, "fused head needs all 64 couts in one wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_in = smem;                       // 2 buffers
    char* const lds_w = smem + 2 * Tile::IN_BYTES;   // 2 slots
    const unsigned lds_in_addr = lds_addr_of(lds_in);
    const unsigned lds_w_addr = lds_addr_of(lds_w);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lc = lane >> 4;
    const int wc = wave % WAVES_C, wp = wave / WAVES_C;

    // ---- this workgroup's item list (XCD-contiguous dealing) -----------------------------------
    const int per_img = a.tilesX * a.tilesY;
    const int total = a.B * per_img * a.nct;
    const int cpx = gridDim.x >> 3;                   // workgroups per XCD group (grid % 8 == 0)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ipx = (total + 7) >> 3;                 // items per XCD group
    const int xbeg = xcd * ipx, xend = min(xbeg + ipx, total);
    const int first = xbeg + slot;
    const int nmine = first < xend ? (xend - first + cpx - 1) / cpx : 0;
    if (nmine == 0) return;
    auto decode = [&](int k) __attribute__((always_inline)) {
        const int item = first + k * cpx;
        ItemPos p;
        p.ct = item % a.nct;
        int t = item / a.nct;
        const int tx = t % a.tilesX; t /= a.tilesX;
        const int ty = t % a.tilesY;
        p.b = t / a.tilesY;
        p.y0 = ty * TH; p.x0 = tx * TW;
        return p;
    };

    const int nplanes = (a.C0 + a.C1) / PL;
    const int p0 = a.C0 / PL;  // planes [0,p0) come from src0, the rest from src1 (CONCAT_UP)
    const int aH = a.H, aW = a.W;
    const unsigned dma_px_bytes = a.C0 * sizeof(T);
    const char* const zero_page = (const char*)a.zero_page;

    // ---- weight stream ----------------------------------------------------------------------------
    constexpr int NWP = Tile::W_BYTES / 1024;  // 1-KiB pieces per step: 24 (BN=128) / 12 (BN=64)
    auto issue_w = [&](int ct, int plane, int ky, int slot_i) __attribute__((always_inline)) {
        const char* wsrc = (const char*)a.wgt + (size_t)ct * BN * 64 +
                           ((size_t)(plane * 9 + ky * 3) * a.Cout) * 64;
#pragma unroll
        for (int jj = 0; jj < (NWP + 7) / 8; ++jj) {
            const int j = jj * 8 + wave;
            if (j < NWP) {
                const int lrow = j * 16 + (lane >> 2);
                const int kx = lrow / BN, row = lrow - kx * BN;
                const int off = (kx * a.Cout + row) * 64 + (((lane & 3) ^ swz(lrow)) << 4);
                glds16(wsrc + off, __builtin_amdgcn_readfirstlane(
                                       lds_w_addr + (unsigned)(slot_i * Tile::W_BYTES + j * 1024)));
            }
        }
    };

    // ---- in-tile gather, DMA flavour (planes stored as-is in an NHWC tensor) --------------------
    constexpr int NPIECE = THP * TWP / 16;  // 45
    auto gather_dma = [&](const ItemPos& ip, int plane, int buf) __attribute__((always_inline)) {
        const char* const base = (const char*)a.src0 + (size_t)ip.b * aH * aW * dma_px_bytes;
#pragma unroll 1
        for (int j = wave; j < NPIECE; j += 8) {
            const int row = j * 16 + (lane >> 2);
            const int py = row / TWP, px = row - py * TWP;
            const int y = ip.y0 - 1 + py, x = ip.x0 - 1 + px;
            const bool ok = (px < TW + 2) & (y >= 0) & (y < aH) & (x >= 0) & (x < aW);
            const unsigned off = (unsigned)(y * aW + x) * dma_px_bytes + plane * 64 +
                                 (((lane & 3) ^ swz(row)) << 4);
            const char* src = ok ? base + off : zero_page + ((lane & 3) << 4);
            glds16(src, __builtin_amdgcn_readfirstlane(
                            lds_in_addr + (unsigned)(buf * Tile::IN_BYTES + j * 1024)));
        }
    };
    // ---- in-tile gather, register flavour (pool / bilinear): 5 chunks per thread per plane, in
    //      3 batches (2,2,1) that ride along the 3 steps of the previous plane ----------------------
    constexpr int NCH = THP * (TW + 2) * 4;  // 2448
    constexpr int NG = (NCH + 511) / 512;    // 5
    constexpr int GB = 2;
    static_assert(NG <= 3 * GB, "a plane's register gather must fit in 3 batches");
    auto reg_load = [&](const ItemPos& ip, int plane, int batch, uint4* g) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const int i = tid + (batch * GB + k) * 512;
            const int pix = i >> 2, ch = i & 3;
            const int py = pix / (TW + 2), px = pix - py * (TW + 2);
            if (batch * GB + k < NG)
                g[k] = gather_chunk<T, MODE>(a, ip.b, ip.y0 - 1 + py, ip.x0 - 1 + px, plane, ch);
        }
    };
    auto reg_commit = [&](int batch, int buf, const uint4* g) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const int i = tid + (batch * GB + k) * 512;
            const int pix = i >> 2, ch = i & 3;
            const int py = pix / (TW + 2), px = pix - py * (TW + 2);
            const int row = py * TWP + px;
            if (batch * GB + k < NG && i < NCH)
                *reinterpret_cast<uint4*>(lds_in + buf * Tile::IN_BYTES + row * 64 +
                                          ((ch ^ swz(row)) << 4)) = g[k];
        }
    };
    auto plane_is_dma = [&](int plane) __attribute__((always_inline)) {
        return MODE == SRC_DIRECT || (MODE == SRC_CONCAT_UP && plane < p0);
    };

    // ---- per-lane LDS read offsets -----------------------------------------------------------------
    const int a_off = (wc * 64 + l15) * 64 + ((lc ^ swz(l15)) << 4);
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        b_off[kx] = (wp * ROWS_W * TWP + kx + l15) * 64 + ((lc ^ swz(kx + l15)) << 4);

    f32x4 acc[4][NT];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- epilogue of one item: y = relu(acc*scale+shift) -> NHWC (+ optional fused 1x1 head) ----
    auto epilogue = [&](const ItemPos& ip) __attribute__((always_inline)) {
        const int cbase = ip.ct * BN + wc * 64 + lc * 4;
        float4 sc[4], sh[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            sc[m] = *reinterpret_cast<const float4*>(a.scale + cbase + m * 16);
            sh[m] = *reinterpret_cast<const float4*>(a.shift + cbase + m * 16);
        }
        float hw[3][4][4];
        if (HEAD) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const float4 v = c < a.head_nc
                        ? *reinterpret_cast<const float4*>(a.head_w + c * 64 + lc * 4 + m * 16)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
                    hw[c][m][0] = v.x; hw[c][m][1] = v.y; hw[c][m][2] = v.z; hw[c][m][3] = v.w;
                }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int y = ip.y0 + wp * ROWS_W + n / 2;
            const int x = ip.x0 + (n % 2) * 16 + l15;
            const bool ok = (y < aH) && (x < aW);
            const size_t pix = ((size_t)ip.b * aH + y) * aW + x;
            float hsum[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                float v[4];
                v[0] = fmaf(acc[m][n][0], sc[m].x, sh[m].x);
                v[1] = fmaf(acc[m][n][1], sc[m].y, sh[m].y);
                v[2] = fmaf(acc[m][n][2], sc[m].z, sh[m].z);
                v[3] = fmaf(acc[m][n][3], sc[m].w, sh[m].w);
                if (a.relu) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
                }
                if (HEAD) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) hsum[c] = fmaf(v[j], hw[c][m][j], hsum[c]);
                }
                if (ok && a.dst) {
                    T* o = (T*)a.dst + pix * a.Cout + cbase + m * 16;
                    if constexpr (sizeof(T) == 4) {
                        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        *reinterpret_cast<uint2*>(o) =
                            make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                    }
                }
            }
            if (HEAD) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float s = hsum[c];
                    s += __shfl_xor(s, 16);
                    s += __shfl_xor(s, 32);
                    if (c < a.head_nc && ok && lc == 0)
                        a.head_out[(((size_t)ip.b * a.head_nc + c) * aH + y) * aW + x] = s + a.head_b[c];
                }
            }
        }
    };

    // ---- prologue: first weight slice + first in-tile -----------------------------------------------
    ItemPos cur = decode(0);
    issue_w(cur.ct, 0, 0, 0);
    if (plane_is_dma(0)) {
        gather_dma(cur, 0, 0);
    } else {
#pragma unroll
        for (int bt = 0; bt < 3; ++bt) {
            uint4 g[GB];
            reg_load(cur, 0, bt, g);
            reg_commit(bt, 0, g);
        }
    }
    zero_acc();
    wait_all_and_barrier();

    // ---- the plane stream --------------------------------------------------------------------------
    int k_item = 0, plane = 0;
    bool pending_epilogue = false;
    ItemPos done = cur;
    const int nstream = nmine * nplanes;
    for (int q = 0; q < nstream; ++q) {
        // position q+1 of the stream
        const bool has_next = q + 1 < nstream;
        const bool next_item = plane + 1 == nplanes;
        const int nplane = next_item ? 0 : plane + 1;
        ItemPos nxt = cur;
        if (has_next && next_item) nxt = decode(k_item + 1);
        const bool next_dma = plane_is_dma(nplane);
        const int ibuf = q & 1, nbuf = ibuf ^ 1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int s = q * 3 + ky;
            // -- issue: weights of the next step, and (part of) the next plane's in-tile
            if (ky < 2) issue_w(cur.ct, plane, ky + 1, (s + 1) & 1);
            else if (has_next) issue_w(nxt.ct, nplane, 0, (s + 1) & 1);
            uint4 g[GB];
            if (has_next) {
                if (next_dma) { if (ky == 0) gather_dma(nxt, nplane, nbuf); }
                else reg_load(nxt, nplane, ky, g);
            }
            // -- the previous item's accumulators leave before this item's first MFMA
            if (ky == 0 && pending_epilogue) {
                epilogue(done);
                zero_acc();
                pending_epilogue = false;
            }
            // -- 3 taps x (4 x NT) MFMA tiles
            const char* wcur = lds_w + (s & 1) * Tile::W_BYTES + a_off;
            const char* icur = lds_in + ibuf * Tile::IN_BYTES;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                uint4 wa[4], xb[NT];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    wa[m] = *reinterpret_cast<const uint4*>(wcur + (kx * BN + m * 16) * 64);
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    xb[n] = *reinterpret_cast<const uint4*>(
                        icur + b_off[kx] + (((n / 2) + ky) * TWP + (n % 2) * 16) * 64);
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) mma_chunk<T>(acc[m][n], wa[m], xb[n]);
            }
            // -- commit the register-path batch into the idle in-tile buffer
            if (has_next && !next_dma) reg_commit(ky, nbuf, g);
            wait_all_and_barrier();
        }
        if (next_item) {
            pending_epilogue = true;
            done = cur;
            cur = nxt;
            plane = 0;
            ++k_item;
        } else {
            plane = nplane;
        }
    }
    epilogue(done);
}

}  // namespace fiunet
