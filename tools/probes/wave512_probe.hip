// Probe for the round-3 review's item 4: ONE wave per SIMD with a 512-register budget - wave tile 64 couts x 256
// pixels, the 64 accumulator tiles (256 registers) pinned in AGPRs by inline-asm MFMAs, fragment reads
// software-pipelined by hand ACROSS the step barrier (weight taps in a 7-slot ring, double-buffered in-tile) -
// on the up1.0 problem of the bench (B=8, 1024 -> 512 channels at 135x240, bf16), as a real (spot-checked)
// direct 3x3 convolution, to be compared with the shipped two-workgroups-per-CU kernel on the same layer
// (1.57 ms, 1490 TFLOP/s in profiles/r03_*).  Not product code: plain epilogue (relu, 8-B stores), no BatchNorm.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o wave512_probe wave512_probe.hip ; run: ./wave512_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int TH = 16, TW = 32, TWP = 40, THP = TH + 2, BN = 128;
constexpr int IN_BYTES = THP * TWP * 64;     // 46080: one 32-channel plane of the input tile (+ halo)
constexpr int TAP_BYTES = BN * 64;           // 8192: one (plane, kx, ky) tap of 128 couts
constexpr int NSLOT = 7;                     // live taps: 3 being read + 1 prefetched + 3 in flight
constexpr int LDS_BYTES = 2 * IN_BYTES + NSLOT * TAP_BYTES;   // 149504
constexpr int NPIECE = IN_BYTES / 1024;      // 45
constexpr int NPW = (NPIECE + 3) / 4;        // 12 pieces per wave

struct Args {
    const char* src;   // [B][Cin/32][H][W][32] bf16
    const char* wgt;   // [Cin/32][kx][ky][Cout][32] bf16
    char* dst;         // [B][Cout/32][H][W][32] bf16
    int B, H, W, Cin, Cout, tilesX, tilesY, nct;
};

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, unsigned lds_wave_base)
{
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, %0" :: "s"(sbase), "v"(voff), "s"(lds_wave_base) : "memory");
}
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))

__device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__global__ __launch_bounds__(256, 1) void conv_w512(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds_base = (unsigned)(size_t)smem;
    char* const lds_w = smem + 2 * IN_BYTES;

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

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lc = lane >> 4;
    const int wc = wave & 1, wp = wave >> 1;

    f32x4 acc[4][16];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nplanes = a.Cin / 32;
    const int a_off = (wc * 64 + l15) * 64 + ((lc ^ swz(l15)) << 4);
    int b_off[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) b_off[kx] = (wp * 8 * TWP + kx + l15) * 64 + ((lc ^ swz(kx + l15)) << 4);

    // ---- weight taps: tap i = plane * 9 + kx * 3 + ky lives in slot i % 7; a tap is 8 pieces of 1 KiB, two per wave
    const char* const wbase = a.wgt + (size_t)ct * BN * 64;
    const unsigned w_lane_off = (unsigned)((lane >> 2) * 64 + (((lane & 3) ^ swz(lane >> 2)) << 4));
    auto issue_tap = [&](int tap, int slot) __attribute__((always_inline)) {
        const char* wsrc = wbase + (size_t)tap * a.Cout * 64 + (size_t)wave * 2048;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 2 * IN_BYTES + slot * TAP_BYTES + wave * 2048);
        glds16s(wsrc, w_lane_off, dst);
        glds16s(wsrc + 1024, w_lane_off, dst + 1024);
    };

    // ---- in-tile: per-lane source offsets of this wave's pieces (plane-invariant), ~0u = outside the image
    unsigned in_off[NPW];
    const char* const src_b = a.src + (size_t)b * nplanes * a.H * a.W * 64;
    const unsigned plane_bytes = (unsigned)(a.H * a.W) * 64u;
#pragma unroll
    for (int jj = 0; jj < NPW; ++jj) {
        const int j = wave + 4 * jj;
        const int row = j * 16 + (lane >> 2);
        const int py = row / TWP, px = row - py * TWP;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = j < NPIECE && px < TW + 2 && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
        in_off[jj] = ok ? (unsigned)(y * a.W + x) * 64u + (((lane & 3) ^ swz(row)) << 4) : ~0u;
        if (j < NPIECE && !ok) {  // padding slots: zero once, in both buffers
            *reinterpret_cast<uint4*>(smem + j * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(smem + IN_BYTES + j * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
        }
    }
    // pieces [jlo, jhi) of this wave for `plane` into buffer `buf`
    auto issue_in = [&](int plane, int buf, int jlo, int jhi) __attribute__((always_inline)) {
        const char* const base = src_b + (size_t)plane * plane_bytes;
#pragma unroll
        for (int jj = 0; jj < NPW; ++jj) {
            if (jj < jlo || jj >= jhi) continue;
            const int j = wave + 4 * jj;
            if (in_off[jj] != ~0u)
                glds16s(base, in_off[jj], __builtin_amdgcn_readfirstlane(lds_base + buf * IN_BYTES + j * 1024));
        }
    };

    // ---- prologue: plane 0 (buffer 0), taps 0..3
    issue_in(0, 0, 0, NPW);
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_tap(i, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Two operand register sets, used alternately (no register moves): a step works on set P (rows 0..7 and
    // its ky = 0 weights arrived during the previous step) and prefetches the next step's into set Q.
    u32x4 xA[10][2], xB[10][2], w0A[4], w0B[4], w1[4], w2[4];
    auto ld = [&](const char* p) __attribute__((always_inline)) { return *reinterpret_cast<const u32x4*>(p); };
    auto load_row = [&](u32x4 (&dst)[2], int buf, int kx, int i) __attribute__((always_inline)) {
        dst[0] = ld(smem + buf * IN_BYTES + b_off[kx] + (i * TWP) * 64);
        dst[1] = ld(smem + buf * IN_BYTES + b_off[kx] + (i * TWP + 16) * 64);
    };
    auto load_w = [&](u32x4 (&dst)[4], int slot) __attribute__((always_inline)) {
        const char* p = lds_w + slot * TAP_BYTES + a_off;
#pragma unroll
        for (int m = 0; m < 4; ++m) dst[m] = ld(p + m * 16 * 64);
    };
#pragma unroll
    for (int i = 0; i < 8; ++i) load_row(xA[i], 0, 0, i);
    load_w(w0A, 0);

    int slot = 0;   // slot of tap 3 * step (this step's ky = 0)
    auto nxt = [](int s, int k) { s += k; return s >= NSLOT ? s - NSLOT : s; };
    const int nsteps = nplanes * 3;
    int step = 0;
    // One DMA item of a step: k = 0..5 the two pieces of taps 3s+4, 3s+5, 3s+6; k = 6..11 this step's share of the
    // next plane's in-tile (steps kx = 0, 1 only).  All of it is waited for at the END of the step.
    auto dma_item = [&](int k, int plane, int buf, int kx) __attribute__((always_inline)) {
        if (k < 6) {
            const int t = k >> 1;
            {   // unconditional (no control flow in the MFMA stream): past the end the last tap is fetched again
                // into a slot nobody reads any more
                const int tap = min(3 * step + 4 + t, nplanes * 9 - 1);
                const char* wsrc = wbase + (size_t)tap * a.Cout * 64 + (size_t)wave * 2048 + (k & 1) * 1024;
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 2 * IN_BYTES + nxt(slot, 4 + t) * TAP_BYTES +
                                                                    wave * 2048 + (k & 1) * 1024);
                glds16s(wsrc, w_lane_off, dst);
            }
        } else if (kx < 2) {   // (last plane: itself again, into the buffer nobody reads any more)
            issue_in(min(plane + 1, nplanes - 1), buf ^ 1, kx * (NPW / 2) + (k - 6), kx * (NPW / 2) + (k - 6) + 1);
        }
    };
    // 64 MFMAs of one tap with a hook after every MFMA (q = 0..63, a compile-time constant after unrolling): the
    // other instructions of a step are spread one or two at a time into the MFMAs' 16-cycle shadows
    auto tap_mfmas = [&](u32x4 (&w)[4], u32x4 (&x)[10][2], int ky, auto&& hook) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                MFMA(acc[m][n], w[m], x[n / 2 + ky][n % 2]);
                hook(m * 16 + n);
            }
    };
    auto do_step = [&](u32x4 (&xP)[10][2], u32x4 (&w0P)[4], u32x4 (&xQ)[10][2], u32x4 (&w0Q)[4], int plane, int buf,
                       int kx) __attribute__((always_inline)) {
        const char* const w1p = lds_w + nxt(slot, 1) * TAP_BYTES + a_off;
        const char* const w2p = lds_w + nxt(slot, 2) * TAP_BYTES + a_off;
        const char* const w3p = lds_w + nxt(slot, 3) * TAP_BYTES + a_off;
        const int nbuf = kx == 2 ? buf ^ 1 : buf, nkx = kx == 2 ? 0 : kx + 1;
        // tap ky = 0 (operands prefetched during the previous step): the step's DMAs go out under it, one per 4
        // MFMAs, then the 6 fragment reads of tap 1
        tap_mfmas(w0P, xP, 0, [&](int q) __attribute__((always_inline)) {
            if (q % 4 == 1 && q / 4 < 12) dma_item(q / 4, plane, buf, kx);
            if (q == 50) xP[8][0] = ld(smem + buf * IN_BYTES + b_off[kx] + (8 * TWP) * 64);
            if (q == 52) xP[8][1] = ld(smem + buf * IN_BYTES + b_off[kx] + (8 * TWP + 16) * 64);
            if (q >= 54 && q < 62 && q % 2 == 0) w1[(q - 54) / 2] = ld(w1p + ((q - 54) / 2) * 16 * 64);
        });
        tap_mfmas(w1, xP, 1, [&](int q) __attribute__((always_inline)) {
            if (q == 40) xP[9][0] = ld(smem + buf * IN_BYTES + b_off[kx] + (9 * TWP) * 64);
            if (q == 42) xP[9][1] = ld(smem + buf * IN_BYTES + b_off[kx] + (9 * TWP + 16) * 64);
            if (q >= 44 && q < 52 && q % 2 == 0) w2[(q - 44) / 2] = ld(w2p + ((q - 44) / 2) * 16 * 64);
        });
        // next step's first operands (tap 3s+3 and rows 0..7: published by the barrier BEFORE this step), one
        // read per 3 MFMAs
        tap_mfmas(w2, xP, 2, [&](int q) __attribute__((always_inline)) {
            if (q % 3 == 0 && q / 3 < 16)
                xQ[q / 6][(q / 3) % 2] = ld(smem + nbuf * IN_BYTES + b_off[nkx] + ((q / 6) * TWP + ((q / 3) % 2) * 16) * 64);
            if (q % 3 == 0 && q / 3 >= 16 && q / 3 < 20) w0Q[q / 3 - 16] = ld(w3p + (q / 3 - 16) * 16 * 64);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = nxt(slot, 3);
        ++step;
    };
#pragma unroll 1
    for (int plane = 0; plane < nplanes; plane += 2) {   // nplanes is even
        do_step(xA, w0A, xB, w0B, plane, 0, 0);
        do_step(xB, w0B, xA, w0A, plane, 0, 1);
        do_step(xA, w0A, xB, w0B, plane, 0, 2);
        do_step(xB, w0B, xA, w0A, plane + 1, 1, 0);
        do_step(xA, w0A, xB, w0B, plane + 1, 1, 1);
        do_step(xB, w0B, xA, w0A, plane + 1, 1, 2);
    }

    // ---- epilogue: relu, bf16, 8-B stores (tile m = couts wc*64 + m*16 + lc*4 .. +3 of pixel (row, l15 + 16 f))
    const size_t out_b = (size_t)b * (a.Cout / 32) * a.H * a.W * 64;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        const int y = y0 + wp * 8 + n / 2, x = x0 + (n % 2) * 16 + l15;
        if (y < a.H && x < a.W) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int cout = ct * BN + wc * 64 + m * 16 + lc * 4;
                const f32x4 v = acc[m][n];
                uint2 pk;
                pk.x = pack2(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
                pk.y = pack2(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
                *reinterpret_cast<uint2*>(a.dst + out_b + ((size_t)(cout / 32) * a.H * a.W + (size_t)y * a.W + x) * 64 +
                                          (cout % 32) * 2) = pk;
            }
        }
    }
}

static unsigned short f2bf(float f)
{
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main()
{
    const int B = 8, H = 135, W = 240, Cin = 1024, Cout = 512;
    const size_t n_src = (size_t)B * Cin * H * W, n_w = (size_t)Cin * 9 * Cout, n_dst = (size_t)B * Cout * H * W;
    std::vector<unsigned short> hsrc(n_src), hw(n_w);
    unsigned long long rng = 88172645463325252ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 11) & 0xffffff) / 16777216.f; };
    for (auto& v : hsrc) { const float r = rnd() * 2.f - 1.f; v = f2bf(r > 0 ? r : 0.f); }   // relu-like activations
    for (auto& v : hw) v = f2bf((rnd() * 2.f - 1.f) * 0.02f);
    char *dsrc, *dw, *ddst;
    hipMalloc(&dsrc, n_src * 2); hipMalloc(&dw, n_w * 2); hipMalloc(&ddst, n_dst * 2);
    hipMemcpy(dsrc, hsrc.data(), n_src * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), n_w * 2, hipMemcpyHostToDevice);
    hipMemset(ddst, 0xff, n_dst * 2);
    Args a{dsrc, dw, ddst, B, H, W, Cin, Cout, (W + TW - 1) / TW, (H + TH - 1) / TH, Cout / BN};
    const int grid = B * a.tilesX * a.tilesY * a.nct;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_w512), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) conv_w512<<<grid, 256, LDS_BYTES>>>(a);
    hipError_t err = hipDeviceSynchronize();
    printf("launch: grid %d, LDS %d B, status %s\n", grid, LDS_BYTES, hipGetErrorString(err));
    if (err != hipSuccess) return 1;
    float best = 1e9f, sum = 0;
    const int R = 10;
    for (int r = 0; r < R; ++r) {
        hipEventRecord(e0);
        conv_w512<<<grid, 256, LDS_BYTES>>>(a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = 2.0 * B * H * W * 9.0 * Cin * Cout;
    printf("wave512 16x32x128 tile, 1 workgroup/CU: best %.4f ms  mean %.4f ms  %.0f TFLOP/s algorithmic (mean %.0f)\n",
           best, sum / R, flop / best / 1e9, flop / (sum / R) / 1e9);
    // spot check against a double-precision sum on the host
    std::vector<unsigned short> hdst(n_dst);
    hipMemcpy(hdst.data(), ddst, n_dst * 2, hipMemcpyDeviceToHost);
    double worst = 0; int bad = 0;
    for (int k = 0; k < 200; ++k) {
        const int b = (int)(rnd() * B), co = (int)(rnd() * Cout);
        int y = (int)(rnd() * H), x = (int)(rnd() * W);
        if (k < 40) { y = (k & 1) ? H - 1 - (k >> 3) % 3 : (k >> 3) % 3; x = (k & 2) ? W - 1 - (k >> 4) % 3 : (k >> 4) % 3; }  // borders
        double s = 0;
        for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = x + kx - 1;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const float xv = bf2f(hsrc[(((size_t)b * (Cin / 32) + ci / 32) * H * W + (size_t)yy * W + xx) * 32 + ci % 32]);
                    const float wv = bf2f(hw[(((size_t)(ci / 32) * 9 + kx * 3 + ky) * Cout + co) * 32 + ci % 32]);
                    s += (double)xv * wv;
                }
        const double ref = s > 0 ? s : 0;
        const double got = bf2f(hdst[(((size_t)b * (Cout / 32) + co / 32) * H * W + (size_t)y * W + x) * 32 + co % 32]);
        const double d = fabs(got - ref);
        if (d > worst) worst = d;
        if (d > 0.01 + 0.01 * fabs(ref)) ++bad;
    }
    printf("spot check: 200 outputs (40 on the borders), worst |d| %.4g, %d outside tolerance\n", worst, bad);
    return bad != 0;
}
