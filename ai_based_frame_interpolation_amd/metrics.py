"""PSNR / SSIM of uint8 frames on the GPU (the reference's evaluators do this on the host with
scikit-image, model/evaluation.py:194-218, evaluation_simple.py:134-156).  Same definitions and
defaults: PSNR with data_range 255; SSIM with a 7x7 uniform window, sample covariance, K1 0.01,
K2 0.03, mean over the image minus a 3-pixel border.  Inputs stay on the device; results are
float64 tensors, one value per [H, W] plane.  No CPU fallback."""
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
