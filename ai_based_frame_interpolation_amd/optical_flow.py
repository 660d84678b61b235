"""The reference evaluators' third method, the optical-flow baseline (model/evaluation_simple.py:76-103,
model/evaluation.py:134-192), WITHOUT OpenCV.

The reference calls `cv2.calcOpticalFlowFarneback(f0, f1, None, pyr_scale=0.5, levels=3, winsize=15,
iterations=3, poly_n=5, poly_sigma=1.1, flags=0)`, halves the flow and samples frame 0 at (x, y) + flow / 2
with `cv2.remap(..., INTER_LINEAR, BORDER_REPLICATE)`.  OpenCV is a third-party dependency of the reference
(`requirements.txt`, unpinned) that no image this code has run on contains, and the reference holds no flow
or frame fixtures, so this module RESTATES the published algorithm - G. Farneback, "Two-Frame Motion Estimation
Based on Polynomial Expansion" (SCIA 2003), in the form OpenCV's `optflowgf.cpp` implements it: Gaussian
pyramid, per-level polynomial expansion (separable Gaussian-weighted least squares), iterative displacement
update over a box-filtered normal-equation field, border attenuation - and remap's 5-bit fixed-point bilinear
sampling.  **Parity unpinned**: nothing here has been compared with OpenCV's output; what the tests pin is the
behaviour the algorithm must have (a translated texture yields that translation; remap with zero flow is the
identity) and the reference's own quirk, which is kept: it samples frame 0 at grid + flow / 2, i.e. AGAINST the
motion, so on translating content this baseline scores BELOW the linear blend (evaluation.py's docstring and
tests/test_host.py::test_optical_flow_baseline_is_the_references_formula).  When `cv2` IS importable the evaluator calls
OpenCV itself (evaluation.py) and this module is not used.

Everything is plain torch on whatever device the frames are on (host plumbing, not a hot path: the reference
runs it frame by frame on the CPU).
"""
from __future__ import annotations

import math
from typing import Tuple

import torch
import torch.nn.functional as F

_BORDER = (0.14, 0.14, 0.4472, 0.4472, 0.4472)   # attenuation of the 5 outermost rows / columns


def _gauss_kernel_cv(ksize: int, sigma: float, dtype, device) -> torch.Tensor:
    """cv2.getGaussianKernel: fixed small kernels for sigma <= 0 and ksize <= 7, else exp(-x^2 / 2 sigma^2)
    normalised to 1 (sigma <= 0: 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8)."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:
        return torch.tensor(small[ksize], dtype=dtype, device=device)
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = torch.arange(ksize, dtype=torch.float64, device=device) - (ksize - 1) * 0.5
    k = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).to(dtype)


def _gaussian_blur(img: torch.Tensor, ksize: int, sigma: float) -> torch.Tensor:
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) on an [H, W] float image, BORDER_REFLECT_101."""
    k = _gauss_kernel_cv(ksize, sigma, img.dtype, img.device)
    r = ksize // 2
    x = img[None, None]
    if r > 0:
        # reflect-101 needs the image to be larger than the radius; tiny pyramid levels fall back to replicate
        mode = "reflect" if min(img.shape) > r else "replicate"
        x = F.pad(x, (r, r, r, r), mode=mode)
    x = F.conv2d(x, k.view(1, 1, 1, -1))
    x = F.conv2d(x, k.view(1, 1, -1, 1))
    return x[0, 0]


def _resize_linear(img: torch.Tensor, width: int, height: int) -> torch.Tensor:
    """cv2.resize(img, (width, height), interpolation=INTER_LINEAR) for float images (any trailing channel
    dim): source coordinate (dst + 0.5) * scale - 0.5, indices clamped to the image."""
    h, w = img.shape[0], img.shape[1]
    if (h, w) == (height, width):
        return img.clone()

    def axis(n_dst, n_src):
        scale = n_src / n_dst
        f = (torch.arange(n_dst, dtype=torch.float64, device=img.device) + 0.5) * scale - 0.5
        i0 = torch.floor(f)
        t = (f - i0).to(img.dtype)
        i0 = i0.long()
        # OpenCV: below 0 -> index 0 with weight 0; past the end -> last index with weight 0
        t = torch.where(i0 < 0, torch.zeros_like(t), t)
        t = torch.where(i0 >= n_src - 1, torch.zeros_like(t), t)
        i0c = i0.clamp(0, n_src - 1)
        i1c = (i0 + 1).clamp(0, n_src - 1)
        return i0c, i1c, t

    y0, y1, ty = axis(height, h)
    x0, x1, tx = axis(width, w)
    shape_y = (-1, 1) + (1,) * (img.dim() - 2)
    shape_x = (1, -1) + (1,) * (img.dim() - 2)
    ty, tx = ty.view(shape_y), tx.view(shape_x)
    top = img[y0][:, x0] * (1 - tx) + img[y0][:, x1] * tx
    bot = img[y1][:, x0] * (1 - tx) + img[y1][:, x1] * tx
    return top * (1 - ty) + bot * ty


def _poly_exp_setup(n: int, sigma: float, device):
    """FarnebackPrepareGaussian: the 1-D weights g, x g, x^2 g and the four entries of the inverse of the
    6x6 Gram matrix of the basis {1, x, y, x^2, y^2, xy} under the weight g(x) g(y)."""
    x = torch.arange(-n, n + 1, dtype=torch.float64)
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    g = g / g.sum()
    xg, xxg = x * g, x * x * g
    gy, gx = torch.meshgrid(g, g, indexing="ij")
    yy, xx = torch.meshgrid(x, x, indexing="ij")
    wgt = gy * gx
    basis = [torch.ones_like(xx), xx, yy, xx * xx, yy * yy, xx * yy]
    G = torch.tensor([[float((wgt * a * b).sum()) for b in basis] for a in basis], dtype=torch.float64)
    invG = torch.linalg.inv(G)
    ig11, ig03, ig33, ig55 = float(invG[1, 1]), float(invG[0, 3]), float(invG[3, 3]), float(invG[5, 5])
    f32 = lambda t: t.to(torch.float32).to(device)
    return f32(g), f32(xg), f32(xxg), ig11, ig03, ig33, ig55


def _conv_rows(img: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """correlation along y with replicated borders: out[y] = sum_j k[j] img[y + j - n]"""
    n = k.numel() // 2
    x = F.pad(img[None, None], (0, 0, n, n), mode="replicate")
    return F.conv2d(x, k.view(1, 1, -1, 1))[0, 0]


def _conv_cols(img: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    n = k.numel() // 2
    x = F.pad(img[None, None], (n, n, 0, 0), mode="replicate")
    return F.conv2d(x, k.view(1, 1, 1, -1))[0, 0]


def poly_exp(img: torch.Tensor, n: int = 5, sigma: float = 1.1) -> torch.Tensor:
    """FarnebackPolyExp: local quadratic model f(d) ~ d^T A d + b^T d + c of every pixel's neighbourhood
    (Gaussian-weighted least squares, radius n).  Returns [H, W, 5] = (b_y-like r2, b_x-like r3, A_yy r4,
    A_xx r5, A_xy r6) in OpenCV's channel order: (b3 ig11, b2 ig11, b1 ig03 + b5 ig33, b1 ig03 + b4 ig33,
    b6 ig55)."""
    g, xg, xxg, ig11, ig03, ig33, ig55 = _poly_exp_setup(n, sigma, img.device)
    # vertical pass: r0 = g * f, r1 = (y g) * f, r2 = (y^2 g) * f
    r0, r1, r2 = _conv_rows(img, g), _conv_rows(img, xg), _conv_rows(img, xxg)
    # horizontal pass
    b1 = _conv_cols(r0, g)
    b2 = _conv_cols(r0, xg)      # x moment
    b4 = _conv_cols(r0, xxg)     # x^2 moment
    b3 = _conv_cols(r1, g)       # y moment
    b6 = _conv_cols(r1, xg)      # xy moment
    b5 = _conv_cols(r2, g)       # y^2 moment
    return torch.stack([b3 * ig11, b2 * ig11, b1 * ig03 + b5 * ig33, b1 * ig03 + b4 * ig33, b6 * ig55], dim=-1)


def _border_scale(height: int, width: int, device) -> torch.Tensor:
    nb = len(_BORDER)
    sy = torch.ones(height, device=device)
    sx = torch.ones(width, device=device)
    b = torch.tensor(_BORDER, device=device)
    ky, kx = min(nb, height), min(nb, width)
    sy[:ky] *= b[:ky]
    sy[height - ky:] *= b[:ky].flip(0)
    sx[:kx] *= b[:kx]
    sx[width - kx:] *= b[:kx].flip(0)
    return sy[:, None] * sx[None, :]


def update_matrices(R0: torch.Tensor, R1: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """FarnebackUpdateMatrices: R1 sampled (bilinearly) at the displaced position, averaged with R0, and the
    per-pixel normal equations [G11, G12, G22, h1, h2] of the displacement update.  flow[..., 0] = dx, [..., 1] = dy."""
    h, w = flow.shape[0], flow.shape[1]
    ys, xs = torch.meshgrid(torch.arange(h, device=flow.device, dtype=flow.dtype),
                            torch.arange(w, device=flow.device, dtype=flow.dtype), indexing="ij")
    dx, dy = flow[..., 0], flow[..., 1]
    fx, fy = xs + dx, ys + dy
    x1, y1 = torch.floor(fx), torch.floor(fy)
    ax, ay = (fx - x1)[..., None], (fy - y1)[..., None]
    x1, y1 = x1.long(), y1.long()
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc, yc = x1.clamp(0, w - 2), y1.clamp(0, h - 2)
    s = (R1[yc, xc] * (1 - ax) * (1 - ay) + R1[yc, xc + 1] * ax * (1 - ay) +
         R1[yc + 1, xc] * (1 - ax) * ay + R1[yc + 1, xc + 1] * ax * ay)
    ins = inside[..., None]
    r2 = torch.where(inside, s[..., 0], torch.zeros_like(dx))
    r3 = torch.where(inside, s[..., 1], torch.zeros_like(dx))
    r4 = torch.where(inside, (R0[..., 2] + s[..., 2]) * 0.5, R0[..., 2])
    r5 = torch.where(inside, (R0[..., 3] + s[..., 3]) * 0.5, R0[..., 3])
    r6 = torch.where(inside, (R0[..., 4] + s[..., 4]) * 0.25, R0[..., 4] * 0.5)
    del ins
    r2 = (R0[..., 0] - r2) * 0.5
    r3 = (R0[..., 1] - r3) * 0.5
    r2 = r2 + r4 * dy + r6 * dx
    r3 = r3 + r6 * dy + r5 * dx
    sc = _border_scale(h, w, flow.device).to(flow.dtype)
    r2, r3, r4, r5, r6 = r2 * sc, r3 * sc, r4 * sc, r5 * sc, r6 * sc
    return torch.stack([r4 * r4 + r6 * r6, (r4 + r5) * r6, r5 * r5 + r6 * r6, r4 * r2 + r6 * r3, r6 * r2 + r5 * r3], dim=-1)


def _box_mean(M: torch.Tensor, size: int) -> torch.Tensor:
    """mean over a size x size window, replicated borders, on [H, W, C]"""
    m = size // 2
    x = M.permute(2, 0, 1)[None]
    x = F.pad(x, (m, m, m, m), mode="replicate")
    x = F.avg_pool2d(x, kernel_size=size, stride=1)
    return x[0].permute(1, 2, 0)


def update_flow_blur(R0, R1, flow, M, block_size: int, update_mats: bool):
    """FarnebackUpdateFlow_Blur: box-filter the normal equations, solve the 2x2 system per pixel."""
    S = _box_mean(M, block_size)
    g11, g12, g22, h1, h2 = S[..., 0], S[..., 1], S[..., 2], S[..., 3], S[..., 4]
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    new = torch.stack([(g11 * h2 - g12 * h1) * idet, (g22 * h1 - g12 * h2) * idet], dim=-1)
    if update_mats:
        M = update_matrices(R0, R1, new)
    return new, M


@torch.no_grad()
def calc_optical_flow_farneback(prev_u8: torch.Tensor, next_u8: torch.Tensor, pyr_scale: float = 0.5,
                                levels: int = 3, winsize: int = 15, iterations: int = 3, poly_n: int = 5,
                                poly_sigma: float = 1.1) -> torch.Tensor:
    """Dense flow prev -> next of two [H, W] uint8 (or float) frames, [H, W, 2] = (dx, dy) float32; the
    argument meaning of cv2.calcOpticalFlowFarneback with flags = 0 (box window, no initial flow)."""
    if prev_u8.shape != next_u8.shape or prev_u8.dim() != 2:
        raise ValueError("expected two [H, W] frames of equal shape")
    imgs = [prev_u8.to(torch.float32), next_u8.to(torch.float32)]
    H, W = prev_u8.shape
    min_size = 32
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if W * scale < min_size or H * scale < min_size:
            break
        k += 1
    levels = k
    flow = None
    for k in range(levels, -1, -1):
        scale = pyr_scale ** k
        sigma = (1.0 / scale - 1.0) * 0.5
        smooth = max(int(round(sigma * 5)) | 1, 3)
        w, h = int(round(W * scale)), int(round(H * scale))
        if flow is None:
            flow = torch.zeros(h, w, 2, dtype=torch.float32, device=prev_u8.device)
        else:
            flow = _resize_linear(flow, w, h) * (1.0 / pyr_scale)
        R = []
        for img in imgs:
            blurred = _gaussian_blur(img, smooth, sigma)
            R.append(poly_exp(_resize_linear(blurred, w, h), poly_n, poly_sigma))
        M = update_matrices(R[0], R[1], flow)
        for i in range(iterations):
            flow, M = update_flow_blur(R[0], R[1], flow, M, winsize, i < iterations - 1)
    return flow


_INTER_BITS, _REMAP_COEF_BITS = 5, 15


@torch.no_grad()
def remap_bilinear_u8(src_u8: torch.Tensor, map_x: torch.Tensor, map_y: torch.Tensor) -> torch.Tensor:
    """cv2.remap(src, map_x, map_y, INTER_LINEAR, borderMode=BORDER_REPLICATE) on an [H, W] uint8 image:
    coordinates rounded to 1/32 pixel, integer weights of 15 bits, result (sum + 2^14) >> 15."""
    h, w = src_u8.shape
    tab = 1 << _INTER_BITS
    sx = torch.round(map_x.to(torch.float64) * tab).long()
    sy = torch.round(map_y.to(torch.float64) * tab).long()
    x0, y0 = sx >> _INTER_BITS, sy >> _INTER_BITS
    fx, fy = (sx & (tab - 1)).to(torch.float64) / tab, (sy & (tab - 1)).to(torch.float64) / tab
    one = float(1 << _REMAP_COEF_BITS)
    # OpenCV builds the table in float and rounds every weight to an integer (the four weights of a cell
    # are then corrected to sum to 2^15 exactly; the correction goes to the largest weight)
    wts = torch.stack([(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy], dim=-1) * one
    iw = torch.round(wts).long()
    diff = (1 << _REMAP_COEF_BITS) - iw.sum(-1)
    big = iw.argmax(-1, keepdim=True)
    iw.scatter_add_(-1, big, diff[..., None])
    xa, xb = x0.clamp(0, w - 1), (x0 + 1).clamp(0, w - 1)
    ya, yb = y0.clamp(0, h - 1), (y0 + 1).clamp(0, h - 1)
    s = src_u8.long()
    acc = s[ya, xa] * iw[..., 0] + s[ya, xb] * iw[..., 1] + s[yb, xa] * iw[..., 2] + s[yb, xb] * iw[..., 3]
    return ((acc + (1 << (_REMAP_COEF_BITS - 1))) >> _REMAP_COEF_BITS).clamp(0, 255).to(torch.uint8)


@torch.no_grad()
def optical_flow_interpolation_baseline(frame0_u8: torch.Tensor, frame1_u8: torch.Tensor) -> torch.Tensor:
    """evaluation_simple.py:76-103 on two [H, W] uint8 frames: Farneback flow 0 -> 1 with the reference's
    parameters, frame 0 sampled at (x, y) + flow / 2 clipped to the image, bilinear, replicated border."""
    flow = calc_optical_flow_farneback(frame0_u8, frame1_u8, pyr_scale=0.5, levels=3, winsize=15, iterations=3,
                                       poly_n=5, poly_sigma=1.1)
    h, w = frame0_u8.shape
    ys, xs = torch.meshgrid(torch.arange(h, device=flow.device, dtype=torch.float32),
                            torch.arange(w, device=flow.device, dtype=torch.float32), indexing="ij")
    new_x = (xs + flow[..., 0] * 0.5).clamp(0, w - 1)
    new_y = (ys + flow[..., 1] * 0.5).clamp(0, h - 1)
    return remap_bilinear_u8(frame0_u8, new_x, new_y)
