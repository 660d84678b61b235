// metrics.hip.h -- image-quality metrics of the evaluation path on device, gfx950 only.
//
// The reference scores every interpolated frame on the HOST with scikit-image
// (/root/reference/model/evaluation.py:194-218, evaluation_simple.py:134-156:
// peak_signal_noise_ratio(target, pred, data_range=255), structural_similarity(target, pred,
// data_range=255)) after postprocess_image has produced the uint8 frame.  With the forward on the
// GPU those two host passes (plus the device->host copy of every frame) become the bottleneck of an
// evaluation run, so both are computed here from the uint8 frames fiunet_forward_u8 leaves in HBM.
//
//   sqdiff_u8_kernel / psnr_finalize_kernel : sum (a-b)^2 as an exact 64-bit integer, then
//                                             10*log10(255^2 / (sum/n)) in fp64
//   ssim_u8_kernel / ssim_finalize_kernel   : skimage's default SSIM: 7x7 uniform window, sample
//                                             covariance (49/48), K1 = 0.01, K2 = 0.03, mean over the
//                                             image minus a 3-pixel border.  Window sums of a, b, a^2,
//                                             b^2, ab are exact int32; the per-pixel map and its mean
//                                             are fp64 in a fixed order (deterministic).
// Both are HBM-bound (2 bytes read per pixel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fiunet {

// grid = (blocks per image, images); `sums` zeroed by the caller.  Integer atomics: exact in any order.
__global__ __launch_bounds__(256) void sqdiff_u8_kernel(const uint8_t* __restrict__ a,
                                                        const uint8_t* __restrict__ b, size_t n,
                                                        unsigned long long* __restrict__ sums)
{
    const size_t img = blockIdx.y;
    const uint8_t* pa = a + img * n;
    const uint8_t* pb = b + img * n;
    unsigned long long s = 0;
    // 16 bytes per lane per step where the image base allows it, bytes otherwise
    const bool vec = (((uintptr_t)pa | (uintptr_t)pb) & 15) == 0;
    const size_t nv = vec ? n / 16 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)gridDim.x * 256) {
        const uint4 va = reinterpret_cast<const uint4*>(pa)[i], vb = reinterpret_cast<const uint4*>(pb)[i];
        const unsigned wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w};
        unsigned acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = (int)((wa[k] >> (8 * j)) & 255u) - (int)((wb[k] >> (8 * j)) & 255u);
                acc += (unsigned)(d * d);
            }
        s += acc;
    }
    for (size_t i = nv * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int d = (int)pa[i] - (int)pb[i];
        s += (unsigned)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    // one atomic per workgroup (the 8 per-image counters are a serialisation point in L2)
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t = part[0] + part[1] + part[2] + part[3];
        if (t) atomicAdd(sums + img, t);
    }
}

__global__ void psnr_finalize_kernel(const unsigned long long* __restrict__ sums, size_t n,
                                     double* __restrict__ out, int images)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= images) return;
    const double mse = (double)sums[i] / (double)n;
    out[i] = sums[i] == 0 ? __longlong_as_double(0x7ff0000000000000LL)  // +inf, as skimage
                          : 10.0 * log10((255.0 * 255.0) / mse);
}

constexpr int SSIM_WIN = 7, SSIM_PAD = 3;
constexpr int SSIM_TX = 64, SSIM_TY = 16;  // output pixels per workgroup

// grid = (tiles_x * tiles_y, images): S over the tile's valid-window pixels, summed in a fixed
// order into partial[img][tile].
__global__ __launch_bounds__(256) void ssim_u8_kernel(const uint8_t* __restrict__ a,
                                                      const uint8_t* __restrict__ b, int H, int W,
                                                      int tiles_x, double* __restrict__ partial)
{
    constexpr int IW = SSIM_TX + SSIM_WIN - 1, IH = SSIM_TY + SSIM_WIN - 1, IWP = IW + 2;
    __shared__ uint8_t ta[IH * IWP], tb[IH * IWP];
    __shared__ int hs[5][IH][SSIM_TX];  // horizontal 7-sums of a, b, a^2, b^2, ab
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const size_t img = blockIdx.y;
    const uint8_t* pa = a + img * (size_t)H * W;
    const uint8_t* pb = b + img * (size_t)H * W;
    // output pixel (oy, ox) of the cropped map <-> window rows oy..oy+6, cols ox..ox+6 of the image
    const int oy0 = ty * SSIM_TY, ox0 = tx * SSIM_TX;
    const int OH = H - 2 * SSIM_PAD, OW = W - 2 * SSIM_PAD;
    for (int i = tid; i < IH * IW; i += 256) {
        const int r = i / IW, c = i - r * IW;
        const int y = min(oy0 + r, H - 1), x = min(ox0 + c, W - 1);
        ta[r * IWP + c] = pa[(size_t)y * W + x];
        tb[r * IWP + c] = pb[(size_t)y * W + x];
    }
    __syncthreads();
    for (int i = tid; i < IH * SSIM_TX; i += 256) {
        const int r = i / SSIM_TX, c = i - r * SSIM_TX;
        int sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
#pragma unroll
        for (int k = 0; k < SSIM_WIN; ++k) {
            const int va = ta[r * IWP + c + k], vb = tb[r * IWP + c + k];
            sa += va; sb += vb; saa += va * va; sbb += vb * vb; sab += va * vb;
        }
        hs[0][r][c] = sa; hs[1][r][c] = sb; hs[2][r][c] = saa; hs[3][r][c] = sbb; hs[4][r][c] = sab;
    }
    __syncthreads();
    const double NP = 49.0, cov_norm = NP / (NP - 1.0);
    const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
    double acc = 0.0;
    for (int i = tid; i < SSIM_TY * SSIM_TX; i += 256) {
        const int r = i / SSIM_TX, c = i - r * SSIM_TX;
        if (oy0 + r >= OH || ox0 + c >= OW) continue;
        int s[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < SSIM_WIN; ++k)
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q] += hs[q][r + k][c];
        const double ux = (double)s[0] / NP, uy = (double)s[1] / NP;
        const double uxx = (double)s[2] / NP, uyy = (double)s[3] / NP, uxy = (double)s[4] / NP;
        const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy);
        const double vxy = cov_norm * (uxy - ux * uy);
        const double A1 = 2.0 * ux * uy + C1, A2 = 2.0 * vxy + C2;
        const double B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        acc += (A1 * A2) / (B1 * B2);
    }
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) partial[img * gridDim.x + tile] = red[0];
}

// one workgroup per image: fixed-order sum of the tile partials, then the mean
__global__ __launch_bounds__(256) void ssim_finalize_kernel(const double* __restrict__ partial, int tiles,
                                                            double count, double* __restrict__ out)
{
    __shared__ double red[256];
    const int tid = threadIdx.x;
    const double* p = partial + (size_t)blockIdx.x * tiles;
    double acc = 0.0;
    for (int i = tid; i < tiles; i += 256) acc += p[i];
    red[tid] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = red[0] / count;
}

}  // namespace fiunet
