"""Per-stage HIP-event times of one configuration: python tools/stage_times.py B H W precision [frame_channels] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
b, h, w, prec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cf = int(sys.argv[5]) if len(sys.argv) > 5 else 1
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
bil = (sys.argv[7] != "convt") if len(sys.argv) > 7 else True
dev = torch.device("cuda:0")
m = bench.make_bench_model(prec, frame_channels=cf, bilinear=bil).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(b, cf, h, w, device=dev, generator=g) * 2 - 1
f2 = torch.rand(b, cf, h, w, device=dev, generator=g) * 2 - 1
for _ in range(2): m(f1, f2)
m._ctx.profile_enable(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(steps): m(f1, f2)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps
_, rows = m._ctx.profile_read()
print(f"B={b} {w}x{h} {prec} cf={cf} {'bilinear' if bil else 'ConvTranspose2d'} decoder: {ms:.3f} ms/step, {b / ms * 1e3:.1f} frames/s, sum of stages {sum(r[1] for r in rows):.3f} ms")
for n, t, fl in rows:
    print(f"  {t:8.3f} ms  {fl / (t * 1e-3) / 1e12 if t > 0 else 0:7.1f} TFLOP/s  {n}")
