"""SURVEY 8d config 4 on ONE GPU: 3000 synthetic 1080p uint8 grayscale frames (seeded procedural
pattern: moving discs + randint(0, 30) noise, in the spirit of the reference's demo_simple.py:17-40),
factor 2 -> 2999 forwards.  Timed end to end (host frames in, host frames out, PCIe inclusive) and
compute only (frames resident in HBM).  usage: python tools/video_config4.py [n_frames]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ai_based_frame_interpolation_amd as P

def synth(n, h=1080, w=1920, seed=0):
    g = torch.Generator().manual_seed(seed)
    yy = torch.arange(h).view(h, 1).float(); xx = torch.arange(w).view(1, w).float()
    frames = torch.empty((n, h, w), dtype=torch.uint8)
    discs = [(torch.rand(4, generator=g) * torch.tensor([h, w, 6.0, 6.0]), 40 + 60 * torch.rand(1, generator=g).item(),
              80 + int(150 * torch.rand(1, generator=g).item())) for _ in range(6)]
    for i in range(n):
        img = torch.full((h, w), 20.0)
        for (p, r, lum) in discs:
            cy = (p[0] + (p[2] - 3.0) * i) % h; cx = (p[1] + (p[3] - 3.0) * i) % w
            img = torch.where((yy - cy) ** 2 + (xx - cx) ** 2 < r * r, torch.tensor(float(lum)), img)
        noise = torch.randint(0, 30, (h, w), generator=g, dtype=torch.int16)
        frames[i] = (img.to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
    return frames

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = P.FrameInterpolationUNet(bilinear=True, precision="bf16").to(dev).eval()
t0 = time.perf_counter(); frames = synth(n).pin_memory(); print(f"synthesised {n} frames in {time.perf_counter() - t0:.1f} s", flush=True)
obuf = torch.empty((2 * n - 1, 1080, 1920), dtype=torch.uint8).pin_memory()
P.interpolate_sequence_host(m, frames[:17], batch=8)          # warm-up
t0 = time.perf_counter(); out = P.interpolate_sequence_host(m, frames, batch=8, out=obuf); dt = time.perf_counter() - t0
print(f"end to end (host -> host, PCIe inclusive): {n} frames -> {out.shape[0]} in {dt:.3f} s = "
      f"{(n - 1) / dt:.1f} interpolated frames/s", flush=True)
# compute only, in chunks that fit comfortably in HBM
chunk, tot, cnt = 600, 0.0, 0
ok = True
for s in range(0, n - 1, chunk):
    d = frames[s:s + chunk + 1].to(dev); torch.cuda.synchronize()
    t0 = time.perf_counter(); o2 = P.interpolate_sequence(m, d, batch=8); torch.cuda.synchronize()
    tot += time.perf_counter() - t0; cnt += d.shape[0] - 1
    ok &= bool(torch.equal(out[2 * s:2 * (s + d.shape[0] - 1) + 1], o2.cpu()))
print(f"compute only (frames resident in HBM): {cnt} forwards in {tot:.3f} s = {cnt / tot:.1f} interpolated frames/s")
print("end-to-end output == device-resident output:", ok)
