// conv3x3_pair.hip.h -- the fused 3x3 conv of conv3x3_mfma.hip.h as ONE 8-wave workgroup per CU
// that works on TWO pixel tiles at once (waves 0-3: tile A, waves 4-7: tile B) against ONE shared
// weight stream, for gfx950 only.
//
// Why: conv3x3_mfma_kernel runs two independent 4-wave workgroups per CU, each with its own
// 2 x 24 KiB weight ring, so 96 of the CU's 160 KiB of LDS hold two copies of the same weights and
// there is no room for a second input-tile buffer: at every plane boundary a workgroup stops its
// MFMAs for two barriers and one memory round trip (10-15 % of its time).
// Sharing the ring between the two tiles frees 48 KiB, which here buys a DOUBLE-BUFFERED input tile
// per group:
//     LDS = 2 groups x 2 buffers x in-tile (25.6 KiB at 8x32) + 2 x 24 KiB ring = 148 KiB.
// The gather of plane p+1 is issued during the first (kx = 0) step of plane p into the idle buffer
// and has two full steps (~6k cycles) to land; weights are streamed one step ahead as before.  The K
// loop then has exactly one barrier per step and no gather bubble, the weight bytes a CU pulls from
// L2 halve, and the LDS-DMA queue is drained with COUNTED waits (the in-tile pieces stay in flight
// across the kx = 0 barrier).
//
// Same arithmetic, same summation order (planes, kx, ky) and the same epilogue code as
// conv3x3_mfma_kernel: outputs are bit-identical to it.
//
// Replaces the same reference ops as conv3x3_mfma.hip.h (DoubleConv: /root/reference/model/unet.py:11-18;
// MaxPool2d via EPI_POOL: unet.py:28; OutConv via EPI_HEAD: unet.py:60).
#pragma once
#include "conv3x3_mfma.hip.h"

namespace fiunet {

template <int BN, int TH, int TW> struct PairTile {
    using Base = ConvTile<BN, TH, TW, SRC_DIRECT>;
    static constexpr int IN_BYTES = Base::IN_BYTES;
    static constexpr int W_BYTES = Base::W_BYTES;
    static constexpr int W_OFF = 4 * IN_BYTES;            // [A0 | A1 | B0 | B1 | W slot 0 | W slot 1]
    static constexpr int LDS_BYTES = W_OFF + 2 * W_BYTES;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU: 160 KiB of LDS");
};

template <int N> __device__ __forceinline__ void lds_dma_wait_le()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <typename T, int BN, int TH, int TW, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_pair_kernel(const ConvArgs a)
{
    using Tile = PairTile<BN, TH, TW>;
    using Base = typename Tile::Base;
    constexpr int PL = Elem<T>::PL;
    constexpr int TWP = Base::TWP, THP = Base::THP;
    constexpr int WAVES_C = BN / 64, WAVES_P = 4 / WAVES_C;
    constexpr int FR = TW / 16;
    constexpr int NF = conv_wave_frags(BN, TH, TW);
    constexpr int ROWS_W = NF / FR;
    static_assert(NF == 8 || NF == 4, "wave tile must be 64 couts x 128 or 64 pixels");
    static_assert(TH == ROWS_W * WAVES_P && ROWS_W * FR == NF, "tile does not split over the waves");
    static_assert(EPI != EPI_POOL || ROWS_W % 2 == 0, "pooled epilogue: a wave owns whole row pairs");
    static_assert(!epi_is_splitk(EPI), "small problems stay on conv3x3_mfma_kernel");
    constexpr int HNC = EPI == EPI_HEAD ? 1 : (EPI == EPI_HEAD3 ? 3 : 0);
    static_assert(HNC == 0 || BN == 64, "fused head needs all 64 couts in one wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    char* const lds_in0 = smem + grp * 2 * Tile::IN_BYTES;  // this group's buffer 0; buffer 1 follows
    char* const lds_w = smem + Tile::W_OFF;
    const unsigned lds_in_addr = lds_addr_of(lds_in0);
    const unsigned lds_w_addr = lds_addr_of(lds_w);

    // XCD-aware, bijective block remap (as conv3x3_mfma_kernel): cout tiles of one tile pair and
    // neighbouring pairs get consecutive logical ids on one XCD
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ct = lid % a.nct;
    const int ntiles = a.B * a.tilesY * a.tilesX;
    int t = (lid / a.nct) * 2 + grp;
    const bool live = t < ntiles;  // an odd tile count leaves the last workgroup's group B idle:
    t = min(t, ntiles - 1);        // it recomputes the last tile and stores nothing
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty * TH, x0 = tx * TW;

    const int l15 = lane & 15, lc = lane >> 4;
    const int wc = wave % WAVES_C, wp = wave / WAVES_C;

    f32x4 acc[4][NF];
    conv_acc_init<T, BN, TH, TW, EPI>(a, acc, ct, wc, lc);

    const int a_off = (wc * 64 + l15) * 64 + ((lc ^ swz(l15)) << 4);
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        b_off[kx] = (wp * ROWS_W * TWP + kx + l15) * 64 + ((lc ^ swz(kx + l15)) << 4);

    const int nplanes = a.C0 / PL;
    const int nsteps = nplanes * 3;
    const char* wbase = (const char*)a.wgt + (size_t)ct * BN * 64;

    // ---- weight stream: 8 waves x NW 1-KiB pieces per (plane, ky) step ------------------------
    constexpr int NW = Tile::W_BYTES / 1024 / 8;
    static_assert(NW * 8 * 1024 == Tile::W_BYTES, "weight step must split over 8 waves");
    int w_src_off[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int lrow = (wave8 * NW + j) * 16 + (lane >> 2);
        const int tap = lrow / BN, row = lrow - tap * BN;  // tap = ky within the (plane, kx) step
        w_src_off[j] = (tap * a.Cout + row) * 64 + (((lane & 3) ^ swz(lrow)) << 4);
    }
    auto issue_w = [&](int step) __attribute__((always_inline)) {
        const int pl = step / 3, kx = step - pl * 3;
        const char* wsrc = wbase + ((size_t)(pl * 9 + kx * 3) * a.Cout) * 64;  // packed [plane][kx][ky][cout]
        const unsigned dst = __builtin_amdgcn_readfirstlane(
            lds_w_addr + (unsigned)((step & 1) * Tile::W_BYTES + wave8 * NW * 1024));
#pragma unroll
        for (int j = 0; j < NW; ++j) glds16(wsrc + w_src_off[j], dst + j * 1024);
    };

    // ---- in-tile gather of this group's tile: NPW pieces per wave, per-lane offsets hoisted ------
    constexpr int NPIECE = THP * TWP / 16;
    static_assert(THP * TWP % 16 == 0, "in-tile must be a whole number of 1-KiB pieces");
    constexpr int NPW = (NPIECE + 3) / 4;   // pieces of the wave with the most (wave 0)
    constexpr int NPW_MIN = NPIECE / 4;     // ... and with the fewest: the counted wait below
    const int aH = a.H, aW = a.W;
    const char* const dma_src = (const char*)a.src0 + (size_t)b * a.H * a.W * a.C0 * sizeof(T);
    const char* const zero_src = (const char*)a.zero_page + ((lane & 3) << 4);
    unsigned in_off[NPW];
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) {
        const int row = (wave + 4 * jj) * 16 + (lane >> 2);
        const int py = row / TWP, px = row - py * TWP;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = (px < TW + 2) & (py < THP) & (y >= 0) & (y < aH) & (x >= 0) & (x < aW);
        in_off[jj] = ok ? (unsigned)(y * aW + x) * 64u + (((lane & 3) ^ swz(row)) << 4) : ~0u;
    }
    const unsigned plane_bytes = (unsigned)(aH * aW) * 64u;
    auto gather_plane = [&](int plane, int buf) __attribute__((always_inline)) {
        const char* const base = dma_src + (size_t)plane * plane_bytes;
        const unsigned dst0 = lds_in_addr + (unsigned)(buf * Tile::IN_BYTES);
#pragma unroll
        for (int jj = 0; jj < NPW; ++jj) {
            const int j = wave + 4 * jj;
            if (j < NPIECE) {
                const char* src = in_off[jj] != ~0u ? base + in_off[jj] : zero_src;
                glds16(src, __builtin_amdgcn_readfirstlane(dst0 + (unsigned)j * 1024u));
            }
        }
    };

    issue_w(0);
    gather_plane(0, 0);
    lds_dma_wait_all();
    __builtin_amdgcn_s_barrier();

    int step = 0;
    for (int plane = 0; plane < nplanes; ++plane) {
        const int par = plane & 1;
        const char* const in_cur = lds_in0 + par * Tile::IN_BYTES;
        const bool more = plane + 1 < nplanes;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++step) {
            // resident: W(step) and this plane's in-tile.  In flight after these issues: W(step+1)
            // (oldest), then -- on kx = 0 -- the next plane's in-tile pieces (youngest).
            if (step + 1 < nsteps) issue_w(step + 1);
            if (kx == 0 && more) gather_plane(plane + 1, par ^ 1);
            const char* wcur = lds_w + (step & 1) * Tile::W_BYTES + a_off;
            constexpr bool ROLL = sizeof(T) == 2 && FR == 2 && EPI != EPI_POOL;  // rolling row window: see conv3x3_mfma_kernel (the pooled variant has no registers for it here)
            if constexpr (ROLL) {
                uint4 xb[ROWS_W + 2][FR];
                auto load_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                    for (int f = 0; f < FR; ++f)
                        xb[i][f] = *reinterpret_cast<const uint4*>(in_cur + b_off[kx] + (i * TWP + f * 16) * 64);
                };
#pragma unroll
                for (int i = 0; i < ROWS_W; ++i) load_row(i);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    uint4 wa[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        wa[m] = *reinterpret_cast<const uint4*>(wcur + (ky * BN + m * 16) * 64);
                    if (ky < 2) load_row(ROWS_W + ky);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < NF; ++n) mma_chunk<T>(acc[m][n], wa[m], xb[n / FR + ky][n % FR]);
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
                            in_cur + b_off[kx] + (((n / FR) + ky) * TWP + (n % FR) * 16) * 64);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < NF; ++n) mma_chunk<T>(acc[m][n], wa[m], xb[n]);
                }
            }
            // W(step+1) must have landed before the barrier that publishes it.  LDS-DMAs of a wave
            // retire in issue order, so on kx = 0 it is enough to wait until at most the in-tile
            // pieces issued after it are outstanding (NPW_MIN: a wave that issued one piece more
            // just waits for that piece too); they are needed two barriers later and are waited
            // for by the vmcnt(0) of the next step.
            if (kx == 0 && more) lds_dma_wait_le<NPW_MIN>();
            else lds_dma_wait_all();
            __builtin_amdgcn_s_barrier();
        }
    }

    conv_epilogue<T, BN, TH, TW, EPI>(a, acc, b, live ? y0 : ((aH + 1) & ~1), x0, ct, 0, wc, wp, l15, lc);
}

}  // namespace fiunet
