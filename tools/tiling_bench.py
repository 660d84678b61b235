"""SURVEY 8d config 5 on one GPU: a 2160x3840 pair un-tiled vs cut into 4 strips (+112-row halo).
Per-strip time = what one of 4 GPUs would spend on its band (the strips are independent forwards).
usage: python tools/tiling_bench.py [bf16|fp32]"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import tiling
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
torch.manual_seed(0)
m = P.FrameInterpolationUNet(bilinear=True, precision=prec)
with torch.no_grad():
    for n, p in m.named_parameters():
        if p.dim() == 4 and p.shape[-1] == 3: p.normal_(0, (2.0 / (p.shape[1] * 9)) ** 0.5)
m = m.to(dev).eval()
H, W, N = 2160, 3840, 4
f1 = torch.rand(1, 1, H, W, device=dev) * 2 - 1; f2 = torch.rand(1, 1, H, W, device=dev) * 2 - 1

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

whole = m(f1, f2)
t_whole = timeit(lambda: m(f1, f2))
tiled = tiling.forward_tiled(m.forward_strip, f1, f2, N)
print(f"{prec} 1x{H}x{W}: un-tiled {t_whole:.2f} ms; tiled == un-tiled bitwise: {torch.equal(tiled, whole)}")
t_all = timeit(lambda: tiling.forward_tiled(m.forward_strip, f1, f2, N), n=10)
print(f"  4 strips back to back on ONE GPU: {t_all:.2f} ms ({t_all / t_whole:.2f}x: halo recompute + copies)")
for i, s in enumerate(tiling.strip_plan(H, N)):
    a = f1[..., s.ext0:s.ext1, :].contiguous(); b = f2[..., s.ext0:s.ext1, :].contiguous()
    t = timeit(lambda: m.forward_strip(a, b, s.ext0, H))
    mb_in = 2 * a.numel() * 4 / 1e6; mb_out = (s.core1 - s.core0) * W * 4 / 1e6
    print(f"  strip {i}: rows {s.core0}-{s.core1} (+halo {s.ext0}-{s.ext1}): {t:.2f} ms; "
          f"wire: {mb_in:.1f} MB in, {mb_out:.1f} MB out")
