import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ai_based_frame_interpolation_amd as P
dev = torch.device("cuda:0")
m = P.FrameInterpolationUNet(bilinear=True).to(dev).eval()
for prec in ("fp32", "bf16"):
    m.precision = prec
    for (b, h, w) in ((1, 256, 256), (1, 1080, 1920), (4, 256, 256)):
        f1 = torch.rand(b, 1, h, w, device=dev); f2 = torch.rand(b, 1, h, w, device=dev)
        for _ in range(20): m(f1, f2)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200 if h == 256 else 30
        for _ in range(n): m(f1, f2)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        # latency of a single synchronous call
        t0 = time.perf_counter()
        for _ in range(20): m(f1, f2); torch.cuda.synchronize()
        dl = (time.perf_counter() - t0) / 20
        print(f"{prec} B={b} {h}x{w}: back-to-back {dt*1e3:.3f} ms/forward ({b/dt:.0f} fps), sync latency {dl*1e3:.3f} ms")

# HIP-graph replay of the single-pair case
for prec in ("fp32", "bf16"):
    m.precision = prec
    f1 = torch.rand(1, 1, 256, 256, device=dev); f2 = torch.rand(1, 1, 256, 256, device=dev)
    ref = m(f1, f2).clone()
    g = P.GraphedForward(m, 1, 256, 256)
    out = g(f1, f2)
    torch.cuda.synchronize()
    assert torch.equal(out, ref), (out - ref).abs().max()
    t0 = time.perf_counter()
    for _ in range(200): g(f1, f2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    t0 = time.perf_counter()
    for _ in range(50): g(f1, f2); torch.cuda.synchronize()
    dl = (time.perf_counter() - t0) / 50
    print(f"{prec} B=1 256x256 HIP graph: back-to-back {dt*1e3:.3f} ms, sync latency {dl*1e3:.3f} ms (bit-identical to eager)")
