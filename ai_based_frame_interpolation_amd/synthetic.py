"""Seeded procedural video frames for the benchmarks and the parity tests.

The reference ships no data (`data/.gitkeep`) and there is no codec in this image, so BASELINE
config 4 ("3000-frame synthetic 1080p video") runs on frames made here: a smooth textured
background with slowly moving Gaussian blobs and discs plus a little per-frame noise, in the spirit
of the reference's demo frames (/root/reference/demo_simple.py:17-40).  The motion per frame is
small, so the true middle frame of (t, t+2) is frame t+1 and a linear blend of the two is already
~30 dB from it -- which is what makes a PSNR-vs-truth comparison of two implementations meaningful.

Plumbing only (torch ops on whatever device is asked for); nothing here is on the hot path.
"""
from __future__ import annotations

import torch


def moving_frames(t0: int, n: int, height: int, width: int, device="cpu", seed: int = 0,
                  noise: int = 4, step: float = 1.0) -> torch.Tensor:
    """uint8 `[n, height, width]`: frames t0*step, (t0+1)*step, ... of one seeded scene.  Frame t is
    a pure function of (seed, t*step), so different ranks can generate disjoint ranges of one video."""
    dev = torch.device(device)
    g = torch.Generator().manual_seed(seed)
    nb = 7
    p = torch.rand(nb, 6, generator=g)  # cx, cy, vx, vy, radius, luminance
    ys = torch.arange(height, device=dev, dtype=torch.float32)[:, None]
    xs = torch.arange(width, device=dev, dtype=torch.float32)[None, :]
    s = min(height, width) / 270.0  # scene scale: features and speeds grow with the frame
    out = torch.empty((n, height, width), dtype=torch.uint8, device=dev)
    for i in range(n):
        t = (t0 + i) * step
        img = 96.0 + 36.0 * torch.sin(xs / (23.0 * s) + 0.05 * t) * torch.cos(ys / (17.0 * s) - 0.03 * t)
        for k in range(nb):
            cx = (p[k, 0].item() * width + (p[k, 2].item() - 0.5) * 3.0 * s * t) % width
            cy = (p[k, 1].item() * height + (p[k, 3].item() - 0.5) * 2.0 * s * t) % height
            r = (6.0 + 14.0 * p[k, 4].item()) * s
            lum = 60.0 + 120.0 * p[k, 5].item()
            d2 = (xs - cx) ** 2 + (ys - cy) ** 2
            if k % 2 == 0:   # soft blob
                img = img + lum * torch.exp(-d2 / (2.0 * r * r))
            else:            # disc with a 2-pixel soft edge
                img = img + lum * torch.clamp((r - torch.sqrt(d2)) / 2.0 + 0.5, 0.0, 1.0)
        if noise > 0:
            gn = torch.Generator(device=dev).manual_seed(seed * 1000003 + int(round(t * 16)) + 17)
            img = img + torch.randint(0, noise, (height, width), generator=gn, device=dev).to(torch.float32)
        out[i] = img.clamp_(0.0, 255.0).to(torch.uint8)
    return out


def triplet(height: int, width: int, device="cpu", seed: int = 0, noise: int = 4):
    """(frame t, true middle frame t+1, frame t+2) uint8 `[height, width]` each."""
    f = moving_frames(0, 3, height, width, device, seed, noise)
    return f[0], f[1], f[2]


__all__ = ["moving_frames", "triplet"]
