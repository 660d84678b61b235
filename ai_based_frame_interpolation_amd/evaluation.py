"""The reference's triplet evaluation (model/evaluation_simple.py:134-244, model/evaluation.py:220-330)
with everything per-frame on the device: for each triplet (frame_t0, ground truth, frame_t1) produce the
middle frame with each method, turn it into the uint8 image `postprocess_image` would give
(inference.py:54-61), and score it against the ground truth with PSNR and SSIM (skimage definitions,
data_range 255: metrics.py).  The reference does the scoring on the host, one frame at a time.

Methods: "unet" (the HIP forward, `forward_u8`) and "linear" ((f0 + f1) / 2 on the [-1, 1] tensors,
evaluation_simple.py:71-74).  The reference's third method, Farneback optical flow
(evaluation_simple.py:76-103), is OpenCV code and OpenCV is not in this image: asking for it raises.
Statistics per method follow evaluation_simple.py:226-242 (numpy mean / population std / min / max).
"""
from __future__ import annotations

from typing import Dict, Iterable

import numpy as np
import torch

from . import _native, metrics

METHODS = ("unet", "linear")


def _linear_u8(f0: torch.Tensor, f1: torch.Tensor) -> torch.Tensor:
    """postprocess_image(linear_interpolation_baseline(preprocess(f0), preprocess(f1))) on device."""
    a, b = _native.preprocess_u8(f0), _native.preprocess_u8(f1)
    return _native.postprocess_u8((a + b) / 2.0)


def _frame_psnr(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    """PSNR of the whole frame (all channels pooled into one MSE, as skimage does for an ndarray)."""
    b = pred.shape[0]
    return metrics.psnr_u8(pred.reshape(b, 1, -1), gt.reshape(b, 1, -1)).reshape(b)


@torch.no_grad()
def evaluate_triplets(model, frame_t0: torch.Tensor, frame_t1: torch.Tensor, ground_truth: torch.Tensor,
                      methods: Iterable[str] = METHODS, batch: int = 8) -> Dict:
    """frame_t0, frame_t1, ground_truth: uint8 [N, C, H, W] on the model's device.  Returns the
    reference's result layout: {'total_triplets', 'methods', 'metrics_by_method': {m: {average_psnr,
    average_ssim, std_*, min_*, max_*}}, 'per_triplet': {m: {'psnr': ndarray, 'ssim': ndarray}}}."""
    methods = tuple(methods)
    for m in methods:
        if m == "optical_flow":
            raise NotImplementedError("the optical-flow baseline is OpenCV's Farneback "
                                      "(evaluation_simple.py:76-103); OpenCV is not available here")
        if m not in METHODS:
            raise ValueError(f"unknown method {m!r}; choose from {METHODS}")
    if not (frame_t0.shape == frame_t1.shape == ground_truth.shape) or frame_t0.dim() != 4:
        raise RuntimeError("expected three uint8 [N, C, H, W] tensors of equal shape")
    n = frame_t0.shape[0]
    per = {m: {"psnr": [], "ssim": []} for m in methods}
    for s in range(0, n, batch):
        e = min(s + batch, n)
        f0, f1, gt = frame_t0[s:e], frame_t1[s:e], ground_truth[s:e]
        for m in methods:
            pred = model.forward_u8(f0, f1) if m == "unet" else _linear_u8(f0, f1)
            # one value per frame: channels (RGB variant) are averaged, as skimage's channel_axis does
            per[m]["psnr"].append(_frame_psnr(pred, gt))
            per[m]["ssim"].append(metrics.ssim_u8(pred, gt).mean(dim=1))
    out = {"total_triplets": n, "successful_evaluations": n, "methods": list(methods),
           "metrics_by_method": {}, "per_triplet": {}}
    for m in methods:
        ps = torch.cat(per[m]["psnr"]).cpu().numpy() if n else np.zeros(0)
        ss = torch.cat(per[m]["ssim"]).cpu().numpy() if n else np.zeros(0)
        out["per_triplet"][m] = {"psnr": ps, "ssim": ss}
        out["metrics_by_method"][m] = {
            "average_psnr": float(np.mean(ps)) if n else 0.0, "average_ssim": float(np.mean(ss)) if n else 0.0,
            "std_psnr": float(np.std(ps)) if n else 0.0, "std_ssim": float(np.std(ss)) if n else 0.0,
            "min_psnr": float(np.min(ps)) if n else 0.0, "max_psnr": float(np.max(ps)) if n else 0.0,
            "min_ssim": float(np.min(ss)) if n else 0.0, "max_ssim": float(np.max(ss)) if n else 0.0,
        }
    return out
