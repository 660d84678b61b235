/*
 * Plain-C CPU oracle for the UNet frame-pair forward -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker.  The shipped HIP path never links or calls it.
 *
 * Independent (no PyTorch, no BLAS) restatement of /root/reference/model/unet.py in NCHW
 * fp32 with double-precision accumulation inside each convolution, so that it can arbitrate
 * between the PyTorch-CPU oracle (oracle/unet_oracle.py), the golden vectors recorded from
 * the real reference (tests/golden/) and the HIP kernels.  Pinned in tests/test_oracle.py.
 *
 *   conv3x3 pad=1 no bias -> BatchNorm2d(eval, eps=1e-5) -> ReLU     unet.py:11-18
 *   MaxPool2d(2) (stride 2, floor)                                   unet.py:28
 *   Upsample x2 bilinear align_corners=True; F.pad; cat([skip, up])  unet.py:40,46-54
 *   conv1x1 + bias                                                   unet.py:60
 *   wiring                                                           unet.py:72-95, 105-112
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FIO_BN_EPS 1e-5f

/* y[b,co,y,x] = relu(bn(sum_{ci,ky,kx} in[b,ci,y+ky-1,x+kx-1] * w[co,ci,ky,kx])) */
void fio_conv3x3_bn_relu(const float* in, const float* w, const float* gamma, const float* beta,
                         const float* mean, const float* var, float* out, int B, int Cin,
                         int Cout, int H, int W, int relu)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            double* acc = (double*)malloc(sizeof(double) * (size_t)H * W);
            memset(acc, 0, sizeof(double) * (size_t)H * W);
            for (int ci = 0; ci < Cin; ++ci) {
                const float* ip = in + ((size_t)b * Cin + ci) * H * W;
                const float* wp = w + ((size_t)co * Cin + ci) * 9;
                for (int ky = 0; ky < 3; ++ky)
                    for (int kx = 0; kx < 3; ++kx) {
                        const double wv = wp[ky * 3 + kx];
                        const int y0 = ky == 0 ? 1 : 0, y1 = ky == 2 ? H - 1 : H;
                        const int x0 = kx == 0 ? 1 : 0, x1 = kx == 2 ? W - 1 : W;
                        for (int y = y0; y < y1; ++y) {
                            const float* row = ip + (size_t)(y + ky - 1) * W + (kx - 1);
                            double* arow = acc + (size_t)y * W;
                            for (int x = x0; x < x1; ++x) arow[x] += wv * row[x];
                        }
                    }
            }
            float* op = out + ((size_t)b * Cout + co) * H * W;
            if (gamma) {
                const float invstd = 1.0f / sqrtf(var[co] + FIO_BN_EPS);
                for (size_t i = 0; i < (size_t)H * W; ++i) {
                    float v = ((float)acc[i] - mean[co]) * invstd * gamma[co] + beta[co];
                    op[i] = (relu && v < 0.0f) ? 0.0f : v;
                }
            } else {
                for (size_t i = 0; i < (size_t)H * W; ++i) {
                    float v = (float)acc[i];
                    op[i] = (relu && v < 0.0f) ? 0.0f : v;
                }
            }
            free(acc);
        }
}

void fio_maxpool2(const float* in, float* out, int B, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        const float* ip = in + (size_t)bc * H * W;
        float* op = out + (size_t)bc * Ho * Wo;
        for (int y = 0; y < Ho; ++y)
            for (int x = 0; x < Wo; ++x) {
                float a = ip[(size_t)(2 * y) * W + 2 * x], b = ip[(size_t)(2 * y) * W + 2 * x + 1];
                float c = ip[(size_t)(2 * y + 1) * W + 2 * x],
                      d = ip[(size_t)(2 * y + 1) * W + 2 * x + 1];
                float m = a > b ? a : b, n = c > d ? c : d;
                op[(size_t)y * Wo + x] = m > n ? m : n;
            }
    }
}

/* out[b, 0:Cs] = skip ; out[b, Cs:Cs+Cl] = pad(upsample2x(low)) ; skip is [B,Cs,H,W], low is
 * [B,Cl,h,w]; pad left/top = diff/2, right/bottom the remainder (unet.py:49-53). */
void fio_upsample_pad_concat(const float* low, const float* skip, float* out, int B, int Cl,
                             int h, int w, int Cs, int H, int W)
{
    const int Hu = 2 * h, Wu = 2 * w;
    const int padT = (H - Hu) / 2, padL = (W - Wu) / 2;
    const float sy = Hu > 1 ? (float)(h - 1) / (float)(Hu - 1) : 0.0f;
    const float sx = Wu > 1 ? (float)(w - 1) / (float)(Wu - 1) : 0.0f;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        float* ob = out + (size_t)b * (Cs + Cl) * H * W;
        memcpy(ob, skip + (size_t)b * Cs * H * W, sizeof(float) * (size_t)Cs * H * W);
        for (int c = 0; c < Cl; ++c) {
            const float* lp = low + ((size_t)b * Cl + c) * h * w;
            float* op = ob + (size_t)(Cs + c) * H * W;
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    const int yu = y - padT, xu = x - padL;
                    float v = 0.0f;
                    if (yu >= 0 && yu < Hu && xu >= 0 && xu < Wu) {
                        const float fy = sy * (float)yu, fx = sx * (float)xu;
                        const int y0 = (int)fy, x0 = (int)fx;
                        const int y1 = y0 < h - 1 ? y0 + 1 : y0, x1 = x0 < w - 1 ? x0 + 1 : x0;
                        const float ly = fy - (float)y0, lx = fx - (float)x0;
                        const float hy = 1.0f - ly, hx = 1.0f - lx;
                        v = hy * (hx * lp[(size_t)y0 * w + x0] + lx * lp[(size_t)y0 * w + x1]) +
                            ly * (hx * lp[(size_t)y1 * w + x0] + lx * lp[(size_t)y1 * w + x1]);
                    }
                    op[(size_t)y * W + x] = v;
                }
        }
    }
}

void fio_conv1x1_bias(const float* in, const float* w, const float* bias, float* out, int B,
                      int Cin, int Cout, int H, int W)
{
    const size_t HW = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            float* op = out + ((size_t)b * Cout + co) * HW;
            for (size_t i = 0; i < HW; ++i) {
                double acc = 0.0;
                for (int ci = 0; ci < Cin; ++ci)
                    acc += (double)w[(size_t)co * Cin + ci] * in[((size_t)b * Cin + ci) * HW + i];
                op[i] = (float)acc + bias[co];
            }
        }
}

/* Weight table order: for each of the 9 DoubleConv blocks (inc, down1..4, up1..4), for each of
 * its two convs: {conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var}; then
 * outc.conv.weight, outc.conv.bias  => 9*2*5 + 2 = 92 pointers (state-dict order without the
 * num_batches_tracked counters). */
#define FIO_NPTR 92

static void fio_double_conv(const float* const* p, const float* in, float* mid, float* out, int B,
                            int Cin, int Cmid, int Cout, int H, int W)
{
    fio_conv3x3_bn_relu(in, p[0], p[1], p[2], p[3], p[4], mid, B, Cin, Cmid, H, W, 1);
    fio_conv3x3_bn_relu(mid, p[5], p[6], p[7], p[8], p[9], out, B, Cmid, Cout, H, W, 1);
}

/* frame1/frame2: [B,cf,H,W]; out: [B,n_classes,H,W]; returns 0, or -1 on bad size / OOM. */
int fio_unet_forward(const float* const* wt, const float* frame1, const float* frame2, float* out,
                     int B, int cf, int n_classes, int H, int W)
{
    if (H < 16 || W < 16 || B < 1) return -1;
    static const int cm[9] = {64, 128, 256, 512, 512, 512, 256, 128, 64};
    static const int co[9] = {64, 128, 256, 512, 512, 256, 128, 64, 64};
    int hs[5], ws[5];
    hs[0] = H; ws[0] = W;
    for (int k = 1; k < 5; ++k) { hs[k] = hs[k - 1] / 2; ws[k] = ws[k - 1] / 2; }
    const size_t HW = (size_t)H * W;
    const size_t big = (size_t)B * 128 * HW; /* largest tensor: up4 concat, 128 ch at full res */
    float* x0 = (float*)malloc(sizeof(float) * (size_t)B * 2 * cf * HW);
    float* skip[5];
    float* t0 = (float*)malloc(sizeof(float) * big);
    float* t1 = (float*)malloc(sizeof(float) * big);
    float* t2 = (float*)malloc(sizeof(float) * big);
    if (!x0 || !t0 || !t1 || !t2) return -1;
    for (int b = 0; b < B; ++b) { /* torch.cat([frame1, frame2], dim=1)  (unet.py:109) */
        memcpy(x0 + (size_t)b * 2 * cf * HW, frame1 + (size_t)b * cf * HW, sizeof(float) * cf * HW);
        memcpy(x0 + ((size_t)b * 2 + 1) * cf * HW, frame2 + (size_t)b * cf * HW,
               sizeof(float) * cf * HW);
    }
    for (int k = 0; k < 5; ++k) {
        skip[k] = (float*)malloc(sizeof(float) * (size_t)B * co[k] * hs[k] * ws[k]);
        if (!skip[k]) return -1;
    }
    fio_double_conv(wt, x0, t0, skip[0], B, 2 * cf, cm[0], co[0], H, W);
    for (int k = 1; k < 5; ++k) {
        fio_maxpool2(skip[k - 1], t0, B, co[k - 1], hs[k - 1], ws[k - 1]);
        fio_double_conv(wt + 10 * k, t0, t1, skip[k], B, co[k - 1], cm[k], co[k], hs[k], ws[k]);
    }
    const float* cur = skip[4];
    float* dec[2] = {t2, (float*)malloc(sizeof(float) * big)};
    if (!dec[1]) return -1;
    int ccur = co[4], lv = 4;
    for (int k = 5; k < 9; ++k) {
        const int sl = 8 - k; /* skip level 3,2,1,0 */
        float* dst = dec[k & 1];
        fio_upsample_pad_concat(cur, skip[sl], t0, B, ccur, hs[lv], ws[lv], co[sl], hs[sl], ws[sl]);
        fio_double_conv(wt + 10 * k, t0, t1, dst, B, co[sl] + ccur, cm[k], co[k], hs[sl], ws[sl]);
        cur = dst; ccur = co[k]; lv = sl;
    }
    fio_conv1x1_bias(cur, wt[90], wt[91], out, B, 64, n_classes, H, W);
    free(x0); free(t0); free(t1); free(dec[0]); free(dec[1]);
    for (int k = 0; k < 5; ++k) free(skip[k]);
    return 0;
}
