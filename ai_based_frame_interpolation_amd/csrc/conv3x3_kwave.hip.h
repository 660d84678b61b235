// conv3x3_kwave.hip.h -- the fused 3x3 conv of conv3x3_mfma.hip.h for SMALL problems with a long K loop, with the K
// loop cut INSIDE the workgroup: the four waves of a workgroup work on the SAME 64 couts x 2x32 pixels, each on its own
// quarter of the input planes, against its own private in-tile and weight ring in LDS, and meet once, through LDS, at
// the end.  gfx950 only; all three precisions.
//
// Why (round 6): the reference itself only ever forwards ONE 256x256 pair
// (/root/reference/model/inference.py:29,101-122).  At that size a deep layer has 4-64 tiles and 8-32 planes of K; a lone
// workgroup takes ~0.6 us per step (one plane x kx) whatever is done about the weight stream, so the layer's time is
// the LENGTH OF THE SERIAL STEP CHAIN.  conv3x3_mfma_kernel shortens it by cutting K over workgroups - which costs an fp32
// slab (written, then read back through the Infinity Cache) and a second dispatch for the reduction (~4.5 us floor +
// 3-8 us; an in-kernel reduction by the last workgroup to arrive needs agent-scope fences that cost more than the
// dispatch, measured).  Cutting K over the WAVES of a workgroup shortens the chain four times with no slab, no second
// dispatch and no barrier inside the K loop at all: a wave's LDS traffic is private, so its only synchronisation is its
// own `s_waitcnt`.
//
// Same arithmetic per product as conv3x3_mfma_kernel (the element type's MFMA, fp32 accumulation, BatchNorm scale in the weights and
// shift in the accumulators' start value, the same epilogue code incl. the fused MaxPool2d(2) copy); the fp32 SUMMATION
// ORDER of an output element is that of a 4-way K cut: ((q0 + q1) + q2) + q3 over the plane quarters, planes / kx / ky in
// order inside a quarter.  Deterministic; independent of the batch and of the position in the batch.
//
// Replaces the same reference ops as conv3x3_mfma.hip.h (DoubleConv: /root/reference/model/unet.py:11-18;
// MaxPool2d via EPI_POOL: unet.py:28).  Direct sources only (one or two full-resolution tensors).
#pragma once
#include "conv3x3_mfma.hip.h"

namespace fiunet {

struct KWaveTile {
    static constexpr int BN = 64, TH = 2, TW = 32;
    static constexpr int TWP = 40, THP = TH + 2;                 // in-tile: 4 rows x 40 pixels (pitch multiple of 8)
    static constexpr int IN_BYTES = THP * TWP * 64;              // 10 240
    static constexpr int W_BYTES = 3 * BN * 64;                  // one (plane, kx) step: 3 taps x 64 rows x 64 B
    static constexpr int W_SLOTS = 2;
    static constexpr int WAVE_BYTES = IN_BYTES + W_SLOTS * W_BYTES;   // 34 816 per wave
    static constexpr int LDS_BYTES = 4 * WAVE_BYTES;                  // 139 264: one workgroup per CU
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU: 160 KiB of LDS");
    static_assert(16 * 1024 <= WAVE_BYTES, "a wave's partial sums (16 KiB) are exchanged through its own region");
};

// EPI: EPI_PLAIN or EPI_POOL (conv_epilogue of conv3x3_mfma.hip.h, called by wave 0 on the reduced sums).
// X2: precision "bf16x2" (SRC_DIRECT_X2 of conv3x3_mfma.hip.h): two-piece activations [hi planes | lo planes] and weights
// [wh | wl]; a wave's quarter is a range of REAL planes, each run as three virtual planes - (xh, wh), (xh, wl) on the same
// in-tile, (xl, wh) - and the epilogue writes the two pieces of the output.
// T: __bf16 (precisions "bf16" / "bf16x2") or float (the exact-fp32 path: same 64-B plane records and LDS images, 16
// channels per plane, four fp32 MFMAs per fragment pair - there a step is MFMA time, so the kernel is chosen where it keeps
// every SIMD of the chip busy, i.e. from 256 workgroups x 4 waves on, and what it saves is the slab and the reduce dispatch).
template <int EPI, bool X2 = false, typename T = __bf16>
__global__ __launch_bounds__(256, 1) void conv3x3_kwave_kernel(const ConvArgs a)
{
    using Tile = KWaveTile;
    constexpr int PL = Elem<T>::PL;
    static_assert(!X2 || sizeof(T) == 2, "two-piece operands are bf16");
    constexpr int TWP = Tile::TWP, THP = Tile::THP, TW = Tile::TW, TH = Tile::TH, BN = Tile::BN;
    constexpr int FR = 2, NF = 4, ROWS_W = 2;
    static_assert(EPI == EPI_PLAIN || EPI == EPI_POOL, "plain or pooled epilogue");
    static_assert(conv_wave_frags(64, 8, 32) == NF, "the epilogue is instantiated for the 64 x 64 wave tile");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lc = lane >> 4;
    char* const my = smem + wave * Tile::WAVE_BYTES;     // [in-tile | W slot 0 | W slot 1], private to this wave
    char* const lds_in = my;
    char* const lds_w = my + Tile::IN_BYTES;
    const unsigned lds_in_addr = lds_addr_of(lds_in);
    const unsigned lds_w_addr = lds_addr_of(lds_w);

    // XCD-aware, bijective block remap (as conv3x3_mfma_kernel)
    int lid;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int ct = lid % a.nct;
    int t = lid / a.nct;
    const int tx = t % a.tilesX; t /= a.tilesX;
    const int ty = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    const int aH = a.H, aW = a.W;

    // this wave's quarter of the K loop (planes); the BatchNorm shift starts wave 0's accumulators, zero the others'
    const int nplanes = (a.C0 + a.C1) / PL;      // REAL planes
    const int p0 = a.C0 / PL;
    const int pbeg = wave * nplanes / 4, pend = (wave + 1) * nplanes / 4;
    constexpr int VP = X2 ? 3 : 1;                // virtual planes per real plane
    const int nsteps = (pend - pbeg) * 3 * VP;
    f32x4 acc[4][NF];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave == 0) {
            const float4 sh = *reinterpret_cast<const float4*>(a.shift + ct * BN + conv_cout_ofs<T>(m, lc));
            v = f32x4{sh.x, sh.y, sh.z, sh.w};
        }
#pragma unroll
        for (int n = 0; n < NF; ++n) acc[m][n] = v;
    }

    // per-lane LDS read offsets, as in conv3x3_mfma_kernel with wc = wp = 0
    const int a_off = l15 * 64 + ((lc ^ swz(l15)) << 4);
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) b_off[kx] = (kx + l15) * 64 + ((lc ^ swz(kx + l15)) << 4);

    // ---- weight stream: the wave moves all 12 1-KiB pieces of a step itself (packed rows r0 .. r0+15, r0 = 16 j) ----
    const char* const wbase = (const char*)a.wgt + (size_t)ct * BN * 64;
    const unsigned w_lane_off = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ swz(lane >> 2)) << 4));
    constexpr int NWP = Tile::W_BYTES / 1024;   // 12
    auto issue_w = [&](int step) __attribute__((always_inline)) {
        const int lp = step / 3, kx = step - lp * 3;
        int pl = pbeg + lp;
        if constexpr (X2) {   // virtual plane lp = 3 * (real - pbeg) + j; weight planes: [wh of every real plane | wl of every real plane]
            const int real = pbeg + lp / 3;
            pl = lp % 3 == 1 ? nplanes + real : real;
        }
        const char* wsrc = wbase + ((size_t)(pl * 9 + kx * 3) * a.Cout) * 64;   // packed [plane][kx][ky][cout]
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_w_addr + (unsigned)((step & 1) * Tile::W_BYTES));
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int r0 = j * 16, tap = r0 / BN;
            glds16s(wsrc + (size_t)(tap * (a.Cout - BN) + r0) * 64, w_lane_off, dst + j * 1024);
        }
    };
    if (nsteps > 0) issue_w(0);

    // ---- in-tile gather: 10 pieces of 16 in-tile pixels, the per-lane source offsets hoisted (plane-invariant) ----
    constexpr int NPIECE = THP * TWP / 16;   // 10
    static_assert(THP * TWP % 16 == 0, "in-tile must be a whole number of 1-KiB pieces");
    const char* const dma_src = (const char*)a.src0 + (size_t)b * aH * aW * a.C0 * sizeof(T) * (X2 ? 2 : 1);
    const char* const dma_src1 = a.C1 > 0 ? (const char*)a.src1 + (size_t)b * aH * aW * a.C1 * sizeof(T) * (X2 ? 2 : 1) : nullptr;
    const unsigned plane_bytes = (unsigned)(aH * aW) * 64u;
    unsigned in_off[NPIECE];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
        const int row = j * 16 + (lane >> 2);
        const int py = row / TWP, px = row - py * TWP;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = (px < TW + 2) & ((unsigned)y < (unsigned)aH) & ((unsigned)x < (unsigned)aW);
        in_off[j] = ok ? (unsigned)(y * aW + x) * 64u + (((lane & 3) ^ swz(row)) << 4) : ~0u;
        // padding slots (outside the image, row-pitch filler) are the same for every plane: zeroed once
        if (!ok) *reinterpret_cast<uint4*>(lds_in + j * 1024 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    // plane: real plane; lo: its lo piece (X2: the lo planes follow the hi planes of their tensor)
    auto gather_plane = [&](int plane, bool lo) __attribute__((always_inline)) {
        const char* base = plane >= p0 ? dma_src1 + (size_t)(plane - p0 + (lo ? nplanes - p0 : 0)) * plane_bytes
                                       : dma_src + (size_t)(plane + (lo ? p0 : 0)) * plane_bytes;
#pragma unroll
        for (int j = 0; j < NPIECE; ++j)
            if (in_off[j] != ~0u) glds16s(base, in_off[j], __builtin_amdgcn_readfirstlane(lds_in_addr + (unsigned)j * 1024u));
    };
    if (nsteps > 0) gather_plane(pbeg, false);
    lds_dma_wait_all();   // W(0), the first in-tile (and the zero fill, in program order) - no barrier: the region is this wave's own

    int step = 0;
    const int nvp = (pend - pbeg) * VP;           // virtual planes of this wave
    for (int vp = 0; vp < nvp; ++vp) {
        // the next virtual plane's in-tile: X2 - after (xh, wh) comes (xh, wl) on the SAME in-tile (no gather), then the lo piece
        const int nreal = pbeg + (vp + 1) / VP, nj = (vp + 1) % VP;
        const bool gather_next = vp + 1 < nvp && !(X2 && nj == 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx, ++step) {
            // W(step+1) into the other slot: this wave's reads of it (step - 1) completed before that step's MFMAs issued
            if (step + 1 < nsteps) issue_w(step + 1);
            const char* wcur = lds_w + (step & 1) * Tile::W_BYTES + a_off;
            const bool next_tile = kx == 2 && gather_next;
            uint4 xb[ROWS_W + 2][FR];
            auto load_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                for (int f = 0; f < FR; ++f)
                    xb[i][f] = *reinterpret_cast<const uint4*>(lds_in + b_off[kx] + (i * TWP + f * 16) * 64);
            };
#pragma unroll
            for (int i = 0; i < ROWS_W; ++i) load_row(i);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                uint4 wa[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) wa[m] = *reinterpret_cast<const uint4*>(wcur + (ky * BN + m * 16) * 64);
                if (ky < 2) load_row(ROWS_W + ky);
                if (ky == 1 && next_tile) {
                    // that was this wave's last read of the plane's in-tile: once the reads have RETURNED (nobody else
                    // touches the region) the next plane's DMA goes out, under the step's remaining 32 MFMAs
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    gather_plane(nreal, X2 && nj == 2);
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < NF; ++n) mma_chunk<T>(acc[m][n], wa[m], xb[n / FR + ky][n % FR]);
            }
            lds_dma_wait_all();   // W(step+1) and, at a plane's end, the next in-tile have landed (wave-private: no barrier)
        }
    }

    // ---- the four quarters meet: waves 1..3 park their sums in their own region, wave 0 adds them in wave order ----
    if (wave != 0) {
        float4* o = reinterpret_cast<float4*>(my) + lane;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n)
                o[(m * NF + n) * 64] = make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]);
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float4* p = reinterpret_cast<const float4*>(smem + w * Tile::WAVE_BYTES) + lane;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NF; ++n) {
                const float4 q = p[(m * NF + n) * 64];
                acc[m][n][0] += q.x; acc[m][n][1] += q.y; acc[m][n][2] += q.z; acc[m][n][3] += q.w;
            }
    }
    // the epilogue of the 64 x 64 wave tile (rows y0, y0 + 1 of the image: wp = 0): ReLU, the blocked store, the pooled copy
    conv_epilogue<T, 64, 8, 32, EPI, X2>(a, acc, b, y0, x0, ct, 0, 0, 0, l15, lc);
}

}  // namespace fiunet
