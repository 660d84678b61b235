"""TEST INFRASTRUCTURE ONLY (see oracle/unet_oracle.py's header): CPU restatement of the two
image-quality metrics the reference's evaluators call on the uint8 frames
(/root/reference/model/evaluation.py:194-218, evaluation_simple.py:134-156):

    psnr(target, pred, data_range=255)   # skimage.metrics.peak_signal_noise_ratio
    ssim(target, pred, data_range=255)   # skimage.metrics.structural_similarity

scikit-image is a third-party dependency of the reference (`requirements.txt:8`, UNPINNED) and is
not installed in this image, and the reference holds no golden PSNR/SSIM values, so this restatement
follows skimage's published algorithm (structural_similarity with its defaults: win_size 7,
uniform filter = scipy.ndimage.uniform_filter, use_sample_covariance=True, K1 0.01, K2 0.03, mean of
the map cropped by (win_size-1)//2) and is pinned only by `ssim_bruteforce` below, an independent
pure-Python evaluation of the SSIM definition on small images: **parity unpinned against skimage
itself**.  Only tests/ may import this module.

`ssim_gauss` / `combined_loss` restate the reference's OTHER SSIM, the pure-torch Gaussian-window one of
its training loss (/root/reference/model/train.py:18-87).  That one IS pinned: oracle/gen_golden.py
(`--ssim-only`) imports the real `train.SSIMLoss` / `train.CombinedLoss` in the build container and records
inputs + values in tests/golden/ssim_gauss_*.npz; tests/test_oracle.py checks this restatement against
them.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage


def gauss_window(window_size: int = 11, sigma: float = 1.5, dtype=None):
    """train.py:27-35: normalised fp32 1-D Gaussian -> [1, 1, ws, ws] outer product (fp32 product, as
    the reference's `.mm`; with dtype=float64 the outer product of the SAME fp32 1-D values is taken in
    double, which is what a separable evaluation with fp64 sums computes)."""
    import torch
    g = torch.tensor([float(np.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)))
                      for x in range(window_size)], dtype=torch.float32)
    g = (g / g.sum()).unsqueeze(1)
    if dtype is not None and dtype != torch.float32:
        g = g.to(dtype)
        return g.mm(g.t()).unsqueeze(0).unsqueeze(0)
    return g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)


def ssim_gauss(img1, img2, window_size: int = 11, size_average: bool = True, dtype=None):
    """SSIMLoss._ssim (train.py:37-56) on [B, C, H, W] torch CPU tensors: depth-wise conv2d with zero
    padding window_size//2.  dtype=torch.float64 evaluates the same formula in double (the value the
    device kernel, which sums in fp64, should sit next to)."""
    import torch
    import torch.nn.functional as F
    c = img1.shape[1]
    dt = dtype or img1.dtype
    a, b = img1.to(dt), img2.to(dt)
    win = gauss_window(window_size, dtype=dt).expand(c, 1, window_size, window_size).contiguous()
    conv = lambda t: F.conv2d(t, win, padding=window_size // 2, groups=c)
    mu1, mu2 = conv(a), conv(b)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    # the products are fp32 tensors in the reference whatever the accumulation type
    s1 = conv((img1 * img1).to(dt)) - mu1_sq
    s2 = conv((img2 * img2).to(dt)) - mu2_sq
    s12 = conv((img1 * img2).to(dt)) - mu1_mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def combined_loss(pred, target, mse_weight: float = 0.5, ssim_weight: float = 0.5, dtype=None):
    """CombinedLoss.forward (train.py:75-87)."""
    dt = dtype or pred.dtype
    mse = ((pred - target).to(dt) ** 2).mean()
    return mse_weight * mse + ssim_weight * (1 - ssim_gauss(pred, target, dtype=dtype))


def psnr_u8(pred: np.ndarray, target: np.ndarray) -> float:
    err = np.mean((target.astype(np.float64) - pred.astype(np.float64)) ** 2, dtype=np.float64)
    if err == 0:
        return float("inf")
    return float(10.0 * np.log10((255.0 ** 2) / err))


def ssim_u8(pred: np.ndarray, target: np.ndarray, win: int = 7) -> float:
    """structural_similarity(target, pred, data_range=255) for 2-D uint8 images."""
    if min(pred.shape) < win:
        raise ValueError("win_size exceeds image extent")
    x = target.astype(np.float64)
    y = pred.astype(np.float64)
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    f = lambda im: ndimage.uniform_filter(im, size=win)
    ux, uy = f(x), f(y)
    uxx, uyy, uxy = f(x * x), f(y * y), f(x * y)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:s.shape[0] - pad, pad:s.shape[1] - pad].mean(dtype=np.float64))


def ssim_bruteforce(pred: np.ndarray, target: np.ndarray, win: int = 7) -> float:
    """The definition, window by window, exact rational window statistics (small images only)."""
    h, w = pred.shape
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    n = win * win
    tot, cnt = 0.0, 0
    for i in range(h - win + 1):
        for j in range(w - win + 1):
            a = target[i:i + win, j:j + win].astype(np.int64).ravel()
            b = pred[i:i + win, j:j + win].astype(np.int64).ravel()
            ma, mb = a.sum() / n, b.sum() / n
            va = ((a * a).sum() - a.sum() ** 2 / n) / (n - 1)
            vb = ((b * b).sum() - b.sum() ** 2 / n) / (n - 1)
            vab = ((a * b).sum() - a.sum() * b.sum() / n) / (n - 1)
            tot += ((2 * ma * mb + c1) * (2 * vab + c2)) / ((ma * ma + mb * mb + c1) * (va + vb + c2))
            cnt += 1
    return tot / cnt
