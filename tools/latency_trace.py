"""Kernel-level timeline of the single-pair forward (the reference's only operating point: ONE 256x256 pair,
/root/reference/model/inference.py:29,101-122).  Run under `rocprofv3 --kernel-trace`; prints nothing itself but
a wall-clock figure.  usage: python tools/latency_trace.py <precision> [B H W] [iters]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
prec = sys.argv[1]
b, h, w = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 256, 256)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
dev = torch.device("cuda:0")
m = bench.make_bench_model(prec).to(dev).eval()
g = torch.Generator(device=dev).manual_seed(1)
f1 = torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1
f2 = torch.rand(b, 1, h, w, device=dev, generator=g) * 2 - 1
for _ in range(10): m(f1, f2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): m(f1, f2)
torch.cuda.synchronize()
print(f"{prec} B={b} {h}x{w}: {(time.perf_counter() - t0) / iters * 1e3:.4f} ms/forward")
