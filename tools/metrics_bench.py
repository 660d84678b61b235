"""Throughput of the device PSNR / SSIM kernels on 8 x 1080p uint8 frames (HBM roofline: 2 B/pixel read)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from ai_based_frame_interpolation_amd import metrics
dev = torch.device("cuda:0")
a = torch.randint(0, 256, (8, 1, 1080, 1920), dtype=torch.uint8, device=dev)
b = torch.randint(0, 256, (8, 1, 1080, 1920), dtype=torch.uint8, device=dev)
for name, fn in (("psnr_u8", metrics.psnr_u8), ("ssim_u8", metrics.ssim_u8)):
    for _ in range(5): fn(a, b)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
    for _ in range(n): fn(a, b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    gb = 2 * a.numel() / 1e9
    print(f"{name}: {dt*1e6:.1f} us per 8 x 1080p frames = {8/dt:.0f} frames/s, {gb/dt:.0f} GB/s algorithmic "
          f"({gb/dt/8000*100:.1f} % of 8 TB/s)")
# Gaussian-window SSIM + MSE of the training loss (train.py:18-87) on fp32 tensors: 8 B/pixel read
fa = torch.rand(8, 1, 1080, 1920, device=dev)
fb = (fa + 0.1 * (torch.rand(8, 1, 1080, 1920, device=dev) - 0.5)).clamp(0, 1)
loss = metrics.CombinedLoss()
for _ in range(5): loss(fa, fb)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 50
for _ in range(n): loss(fa, fb)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
gb = 8 * fa.numel() / 1e9
print(f"CombinedLoss (ssim_gauss_f32 + mse): {dt*1e6:.1f} us per 8 x 1080p fp32 frames = {8/dt:.0f} frames/s, "
      f"{gb/dt:.0f} GB/s algorithmic ({gb/dt/8000*100:.1f} % of 8 TB/s)")
