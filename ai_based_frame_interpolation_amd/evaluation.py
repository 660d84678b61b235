"""The reference's triplet evaluation (model/evaluation_simple.py:134-244, model/evaluation.py:220-330)
with everything per-frame on the device: for each triplet (frame_t0, ground truth, frame_t1) produce the
middle frame with each method, turn it into the uint8 image `postprocess_image` would give
(inference.py:54-61), and score it against the ground truth with PSNR and SSIM (skimage definitions,
data_range 255: metrics.py).  The reference does the scoring on the host, one frame at a time.

Methods: "unet" (the HIP forward, `forward_u8`), "linear" ((f0 + f1) / 2 on the [-1, 1] tensors,
evaluation_simple.py:71-74) and "optical_flow" (evaluation_simple.py:76-103).  The third one IS OpenCV in the
reference: `cv2.calcOpticalFlowFarneback` + `cv2.remap`.  When `cv2` is importable it is called itself, on the
host, with the reference's parameters; otherwise (every image this code has run on) the method runs
`optical_flow.py`, a torch RESTATEMENT of Farneback's algorithm and of remap's fixed-point sampling that is
**parity-unpinned against OpenCV** (no OpenCV here to compare with, no fixtures in the reference) - the result
dict says which backend produced the numbers (`optical_flow_backend`).  Only the scoring runs in HIP kernels.
Statistics per method follow evaluation_simple.py:226-242 (numpy mean / population std / min / max).
"""
from __future__ import annotations

from typing import Dict, Iterable

import numpy as np
import torch

from . import _native, metrics, optical_flow

METHODS = ("unet", "linear")          # the default pair (no flow estimation)
ALL_METHODS = METHODS + ("optical_flow",)  # the reference's three (evaluation_simple.py:134-244)


def optical_flow_backend() -> str:
    """Which implementation the "optical_flow" method uses in this process."""
    try:
        import cv2  # type: ignore  # noqa: F401
        return "opencv"
    except ImportError:
        return "restated (ai_based_frame_interpolation_amd.optical_flow; parity unpinned against OpenCV)"


def _optical_flow_u8(f0: torch.Tensor, f1: torch.Tensor) -> torch.Tensor:
    """optical_flow_interpolation_baseline (evaluation_simple.py:76-103), frame by frame: Farneback flow
    f0 -> f1 (pyr_scale 0.5, 3 levels, winsize 15, 3 iterations, poly_n 5, poly_sigma 1.1), frame 0 sampled at
    (x, y) + flow/2 clipped to the image, bilinear, replicated border (as written in the reference: this moves
    the content AGAINST its motion, so on translating content the baseline scores below the linear blend).
    uint8 [N,1,H,W] in, uint8 [N,1,H,W] on the same device out.  OpenCV itself when importable, else the
    restatement of optical_flow.py on the frames' own device."""
    if f0.shape[1] != 1:
        raise RuntimeError("the optical-flow baseline is defined on grayscale frames (one channel)")
    try:
        import cv2  # type: ignore
    except ImportError:
        return torch.stack([optical_flow.optical_flow_interpolation_baseline(f0[i, 0], f1[i, 0])
                            for i in range(f0.shape[0])]).unsqueeze(1)
    a_all, b_all = f0[:, 0].cpu().numpy(), f1[:, 0].cpu().numpy()
    out = np.empty_like(a_all)
    for i in range(a_all.shape[0]):
        a, b = np.ascontiguousarray(a_all[i]), np.ascontiguousarray(b_all[i])
        flow = cv2.calcOpticalFlowFarneback(a, b, None, pyr_scale=0.5, levels=3, winsize=15, iterations=3,
                                            poly_n=5, poly_sigma=1.1, flags=0)
        half = flow * 0.5
        h, w = a.shape
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
        new_x = np.clip(xs + half[:, :, 0], 0, w - 1)
        new_y = np.clip(ys + half[:, :, 1], 0, h - 1)
        out[i] = cv2.remap(a, new_x, new_y, cv2.INTER_LINEAR, borderMode=cv2.BORDER_REPLICATE)
    return torch.from_numpy(out).unsqueeze(1).to(f0.device)


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
        if m not in ALL_METHODS:
            raise ValueError(f"unknown method {m!r}; choose from {ALL_METHODS}")
    if "optical_flow" in methods and frame_t0.shape[1] != 1:
        raise RuntimeError("the optical-flow baseline is defined on grayscale frames (one channel)")
    if not (frame_t0.shape == frame_t1.shape == ground_truth.shape) or frame_t0.dim() != 4:
        raise RuntimeError("expected three uint8 [N, C, H, W] tensors of equal shape")
    n = frame_t0.shape[0]
    per = {m: {"psnr": [], "ssim": []} for m in methods}
    for s in range(0, n, batch):
        e = min(s + batch, n)
        f0, f1, gt = frame_t0[s:e], frame_t1[s:e], ground_truth[s:e]
        for m in methods:
            pred = (model.forward_u8(f0, f1) if m == "unet" else
                    _linear_u8(f0, f1) if m == "linear" else _optical_flow_u8(f0, f1))
            # one value per frame: channels (RGB variant) are averaged, as skimage's channel_axis does
            per[m]["psnr"].append(_frame_psnr(pred, gt))
            per[m]["ssim"].append(metrics.ssim_u8(pred, gt).mean(dim=1))
    out = {"total_triplets": n, "successful_evaluations": n, "methods": list(methods),
           "metrics_by_method": {}, "per_triplet": {}}
    if "optical_flow" in methods:
        out["optical_flow_backend"] = optical_flow_backend()
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
