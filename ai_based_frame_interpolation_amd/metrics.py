"""PSNR / SSIM of uint8 frames on the GPU (the reference's evaluators do this on the host with
scikit-image, model/evaluation.py:194-218, evaluation_simple.py:134-156).  Same definitions and
defaults: PSNR with data_range 255; SSIM with a 7x7 uniform window, sample covariance, K1 0.01,
K2 0.03, mean over the image minus a 3-pixel border.  Inputs stay on the device; results are
float64 tensors, one value per [H, W] plane.  No CPU fallback.

Also here: the reference's OTHER SSIM, the Gaussian-window one of its training loss
(model/train.py:18-87: `SSIMLoss`, `CombinedLoss`) - the only SSIM in the reference that is pure torch,
so the only one whose values are pinned by fixtures recorded from the reference itself
(tests/golden/ssim_gauss_*.npz).  `ssim_gauss`, `SSIMLoss` and `CombinedLoss` evaluate it with one HIP
pass over the two fp32 tensors (forward values only: this is the inference tier, nothing here is
differentiable)."""
from __future__ import annotations

import ctypes

import torch

from . import _native


def _planes(pred: torch.Tensor, target: torch.Tensor):
    if pred.shape != target.shape or pred.dim() < 2:
        raise RuntimeError(f"expected two uint8 tensors of equal shape [..., H, W], got "
                           f"{tuple(pred.shape)} and {tuple(target.shape)}")
    if pred.dtype != torch.uint8 or target.dtype != torch.uint8:
        raise RuntimeError("metrics are defined on the uint8 frames (postprocess_image output)")
    if not pred.is_cuda or not target.is_cuda:
        raise RuntimeError("device metrics need CUDA/HIP tensors; there is no CPU fallback here")
    h, w = pred.shape[-2:]
    n = pred.numel() // (h * w) if h * w else 0
    return pred.contiguous(), target.contiguous(), n, h, w


def _run(fn_name, pred, target):
    p, t, n, h, w = _planes(pred, target)
    L = _native.lib()
    nbytes = L.fiunet_metrics_workspace_bytes(n, h, w)
    if nbytes == 0:
        _native.check(1, "fiunet_metrics_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    out = torch.empty(n, dtype=torch.float64, device=p.device)
    with torch.cuda.device(p.device):
        s = torch.cuda.current_stream(p.device).cuda_stream
        _native.check(getattr(L, fn_name)(p.data_ptr(), t.data_ptr(), n, h, w, out.data_ptr(),
                                          ws.data_ptr(), ctypes.c_size_t(nbytes), s), fn_name)
    return out.view(pred.shape[:-2])


def psnr_u8(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """compute_psnr(pred, target) (evaluation.py:194-205) for every [H, W] plane."""
    return _run("fiunet_psnr_u8", pred, target)


def ssim_u8(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """compute_ssim(pred, target) (evaluation.py:207-218) for every [H, W] plane."""
    return _run("fiunet_ssim_u8", pred, target)


_WINDOWS = {}


def _gauss_1d(window_size: int, sigma: float = 1.5) -> torch.Tensor:
    """SSIMLoss._gaussian (train.py:27-29): fp32 tensor of exp(-(x - ws//2)^2 / (2 sigma^2)) divided by
    its own torch sum (host tensor; the kernel applies it separably)."""
    if window_size not in _WINDOWS:
        import math
        g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2))
                          for x in range(window_size)], dtype=torch.float32)
        _WINDOWS[window_size] = (g / g.sum()).contiguous()
    return _WINDOWS[window_size]


def _gauss_planes(img1: torch.Tensor, img2: torch.Tensor, window_size: int):
    """-> (per-plane mean of the SSIM map [B, C] float64, per-plane sum of squared error [B, C] float64)"""
    if img1.shape != img2.shape or img1.dim() != 4:
        raise RuntimeError(f"expected two [B, C, H, W] tensors of equal shape, got {tuple(img1.shape)} and "
                           f"{tuple(img2.shape)}")
    if img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise RuntimeError("the Gaussian-window SSIM is defined on the fp32 tensors of the loss (train.py:192)")
    if not img1.is_cuda or not img2.is_cuda:
        raise RuntimeError("device metrics need CUDA/HIP tensors; there is no CPU fallback here")
    b, c, h, w = img1.shape
    n = b * c
    if n == 0 or h == 0 or w == 0:
        raise RuntimeError("empty input")
    a, t = img1.contiguous(), img2.contiguous()
    L = _native.lib()
    nbytes = L.fiunet_ssim_gauss_workspace_bytes(n, h, w)
    if nbytes == 0:
        _native.check(1, "fiunet_ssim_gauss_workspace_bytes")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    out = torch.empty((2, n), dtype=torch.float64, device=a.device)
    win = _gauss_1d(int(window_size))
    with torch.cuda.device(a.device):
        s = torch.cuda.current_stream(a.device).cuda_stream
        _native.check(L.fiunet_ssim_gauss_f32(a.data_ptr(), t.data_ptr(), n, h, w, int(window_size),
                                              win.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                              ws.data_ptr(), ctypes.c_size_t(nbytes), s), "fiunet_ssim_gauss_f32")
    return out[0].view(b, c), out[1].view(b, c)


def ssim_gauss(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11, size_average: bool = True,
               dtype=None) -> torch.Tensor:
    """SSIMLoss._ssim (train.py:37-56): `ssim_map.mean()` (a 0-dim tensor) if size_average else the
    per-sample mean over (C, H, W) ([B]).  Returned in the inputs' dtype like the reference's result
    unless `dtype` says otherwise (the device value is float64)."""
    planes, _ = _gauss_planes(img1, img2, window_size)
    val = planes.mean() if size_average else planes.mean(dim=1)
    return val.to(dtype or img1.dtype)


class SSIMLoss:
    """train.py:18-73: `1 - ssim` with the 11x11 sigma-1.5 Gaussian window.  The reference rebuilds its
    window when the channel count changes (:59-70); here the window is depth-wise by construction, so
    `channel` is accepted and ignored."""

    def __init__(self, window_size: int = 11, size_average: bool = True, channel: int = 1):
        self.window_size, self.size_average, self.channel = window_size, size_average, channel

    def __call__(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        return 1 - ssim_gauss(img1, img2, self.window_size, self.size_average)

    forward = __call__


class CombinedLoss:
    """train.py:75-87: mse_weight * MSELoss(pred, target) + ssim_weight * SSIMLoss()(pred, target), both
    terms from ONE pass over the two tensors."""

    def __init__(self, mse_weight: float = 0.5, ssim_weight: float = 0.5):
        self.mse_weight, self.ssim_weight = mse_weight, ssim_weight

    def terms(self, pred: torch.Tensor, target: torch.Tensor):
        """-> (mse, ssim) as float64 0-dim device tensors"""
        planes, sq = _gauss_planes(pred, target, 11)
        return sq.sum() / float(pred.numel()), planes.mean()

    def __call__(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        mse, ssim = self.terms(pred, target)
        return (self.mse_weight * mse + self.ssim_weight * (1 - ssim)).to(pred.dtype)

    forward = __call__
