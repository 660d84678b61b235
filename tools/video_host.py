"""PCIe-inclusive rate of the factor-2 video loop: host uint8 1080p frames in, host frames out."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ai_based_frame_interpolation_amd as P
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = P.FrameInterpolationUNet(bilinear=True, precision="bf16").to(dev).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 401
frames = torch.randint(0, 256, (n, 1080, 1920), dtype=torch.uint8).pin_memory()
P.interpolate_sequence_host(m, frames[:17], batch=8)          # warm-up
obuf = torch.empty((2 * n - 1, 1080, 1920), dtype=torch.uint8).pin_memory()
t0 = time.perf_counter(); out = P.interpolate_sequence_host(m, frames, batch=8, out=obuf); dt = time.perf_counter() - t0
print(f"host->host: {n} frames -> {out.shape[0]} in {dt:.3f} s = {(n - 1) / dt:.1f} interpolated frames/s (PCIe inclusive)")
d = frames.to(dev); torch.cuda.synchronize()
t0 = time.perf_counter(); o2 = P.interpolate_sequence(m, d, batch=8); torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
print(f"device-resident: {(n - 1) / dt2:.1f} interpolated frames/s")
print("equal:", bool(torch.equal(out, o2.cpu())))
