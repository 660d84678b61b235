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


// ---------------------------------------------------------------------------------------------------
// Gaussian-window SSIM of the training loss (/root/reference/model/train.py:18-73, SSIMLoss._ssim):
// depth-wise conv2d of img1, img2, img1*img1, img2*img2, img1*img2 with the normalised
// window_size x window_size Gaussian (sigma 1.5, :27-35), ZERO padding window_size//2 (:38-46),
// C1 = 0.01^2, C2 = 0.03^2 (:48-49), map = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2))
// (:51), then a mean (:53-56).  fp32 planes in the caller's value range (the loss is applied to [0,1]
// tensors, train.py:142,192).  The reference's window is the outer product of the normalised 1-D
// Gaussian, so the filter is evaluated separably: horizontal 11-tap sums into LDS, vertical 11-tap sums
// in registers.  The three products are rounded to fp32 first, as the reference's `img1*img1` tensors
// are; all sums, the map and its mean are fp64 in a fixed order (deterministic; the reference's own fp32
// conv carries ~1e-7 of rounding that this does not reproduce, far below the 1e-5 the tests ask for).
// The same pass also returns sum (img1 - img2)^2 per plane for CombinedLoss' MSE term (train.py:75-87).
// Roofline: HBM (8 bytes read per pixel).
// Tile of 16 x 32 output pixels: 36 KiB of LDS per workgroup, four workgroups per CU.  The first version
// (64 x 16 tiles, 82 KiB, ONE workgroup per CU) was latency-bound at one wave per SIMD: 0.69 ms for 8 x 1080p
// frames against 0.30 ms now (-DFIUNET_GSSIM_TX / _TY sweep: 32x16 0.36, 32x32 0.35, 64x8 0.51, 16x48 0.31,
// 16x64 0.34, 8x64 0.31, 8x32 0.33 ms).
#ifndef FIUNET_GSSIM_TX
#define FIUNET_GSSIM_TX 16
#define FIUNET_GSSIM_TY 32
#endif
constexpr int GSSIM_TX = FIUNET_GSSIM_TX, GSSIM_TY = FIUNET_GSSIM_TY, GSSIM_MAXR = 15;
struct GaussWindow { float g[2 * GSSIM_MAXR + 1]; };  // by value in the kernel arguments

// grid = (tiles_x * tiles_y, planes); dynamic LDS: 2 fp32 input tiles + 5 fp64 row-sum planes.
// RT > 0: window radius known at compile time (the reference only ever uses 5); RT == 0: runtime r.
template <int RT>
__global__ __launch_bounds__(256) void ssim_gauss_f32_kernel(const float* __restrict__ a,
                                                             const float* __restrict__ b, int H, int W,
                                                             int tiles_x, int r_arg, GaussWindow win,
                                                             double* __restrict__ partial)
{
    const int R = RT > 0 ? RT : r_arg;
    const int IW = GSSIM_TX + 2 * R, IH = GSSIM_TY + 2 * R, IWP = IW + 1;
    extern __shared__ double gssim_lds[];
    double* hs = gssim_lds;                                    // [5][IH][TX]
    float* ta = reinterpret_cast<float*>(hs + 5 * IH * GSSIM_TX);  // [IH][IWP]
    float* tb = ta + IH * IWP;
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const size_t img = blockIdx.y;
    const float* pa = a + img * (size_t)H * W;
    const float* pb = b + img * (size_t)H * W;
    const int oy0 = ty * GSSIM_TY, ox0 = tx * GSSIM_TX;
    double sq = 0.0;
    for (int i = tid; i < IH * IW; i += 256) {
        const int rr = i / IW, c = i - rr * IW;
        const int y = oy0 + rr - R, x = ox0 + c - R;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;  // F.conv2d padding: zeros
        const float va = in ? pa[(size_t)y * W + x] : 0.f, vb = in ? pb[(size_t)y * W + x] : 0.f;
        ta[rr * IWP + c] = va;
        tb[rr * IWP + c] = vb;
        // every image pixel belongs to exactly one tile's interior
        if (in && rr >= R && rr < R + GSSIM_TY && c >= R && c < R + GSSIM_TX) {
            const float d = va - vb;  // nn.MSELoss: fp32 difference
            sq += (double)d * (double)d;
        }
    }
    __syncthreads();
    for (int i = tid; i < IH * GSSIM_TX; i += 256) {
        const int rr = i / GSSIM_TX, c = i - rr * GSSIM_TX;
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) {
            const float va = ta[rr * IWP + c + k], vb = tb[rr * IWP + c + k];
            const double g = (double)win.g[k];
            s0 = fma(g, (double)va, s0);
            s1 = fma(g, (double)vb, s1);
            s2 = fma(g, (double)(va * va), s2);
            s3 = fma(g, (double)(vb * vb), s3);
            s4 = fma(g, (double)(va * vb), s4);
        }
        hs[(0 * IH + rr) * GSSIM_TX + c] = s0;
        hs[(1 * IH + rr) * GSSIM_TX + c] = s1;
        hs[(2 * IH + rr) * GSSIM_TX + c] = s2;
        hs[(3 * IH + rr) * GSSIM_TX + c] = s3;
        hs[(4 * IH + rr) * GSSIM_TX + c] = s4;
    }
    __syncthreads();
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    double acc = 0.0;
    for (int i = tid; i < GSSIM_TY * GSSIM_TX; i += 256) {
        const int rr = i / GSSIM_TX, c = i - rr * GSSIM_TX;
        if (oy0 + rr >= H || ox0 + c >= W) continue;
        double s[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) {
            const double g = (double)win.g[k];
#pragma unroll
            for (int q = 0; q < 5; ++q) s[q] = fma(g, hs[(q * IH + rr + k) * GSSIM_TX + c], s[q]);
        }
        const double mu1 = s[0], mu2 = s[1];
        const double mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu1_mu2 = mu1 * mu2;
        const double s1 = s[2] - mu1_sq, s2 = s[3] - mu2_sq, s12 = s[4] - mu1_mu2;
        acc += ((2.0 * mu1_mu2 + C1) * (2.0 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
    red[0][tid] = acc;
    red[1][tid] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        partial[(img * gridDim.x + tile) * 2 + 0] = red[0][0];
        partial[(img * gridDim.x + tile) * 2 + 1] = red[1][0];
    }
}

// one workgroup per plane: fixed-order sums of the tile partials -> mean SSIM map, sum of squared error
__global__ __launch_bounds__(256) void ssim_gauss_finalize_kernel(const double* __restrict__ partial, int tiles,
                                                                  double count, double* __restrict__ out_ssim,
                                                                  double* __restrict__ out_sqerr)
{
    __shared__ double red[2][256];
    const int tid = threadIdx.x;
    const double* p = partial + (size_t)blockIdx.x * tiles * 2;
    double acc = 0.0, sq = 0.0;
    for (int i = tid; i < tiles; i += 256) {
        acc += p[2 * i];
        sq += p[2 * i + 1];
    }
    red[0][tid] = acc;
    red[1][tid] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            red[0][tid] += red[0][tid + o];
            red[1][tid] += red[1][tid + o];
        }
        __syncthreads();
    }
    if (tid == 0) {
        out_ssim[blockIdx.x] = red[0][0] / count;
        if (out_sqerr) out_sqerr[blockIdx.x] = red[1][0];
    }
}

}  // namespace fiunet
