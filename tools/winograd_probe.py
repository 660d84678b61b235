"""Go / no-go probe for Winograd F(2x2, 3x3) on the exact-fp32 path (round-2 review, item 7), WITHOUT
building the kernel: measures on this box an UPPER BOUND of what the three-stage form could reach for
the layer the review names (down3.3: 512 -> 512 channels at 135x240, batch 8) and compares it with the
shipped direct conv of the same layer.

  stage 2 (the only MFMA work)   16 batched fp32 GEMMs [tiles x 512] x [512 x 512], tiles = B * 68 * 120,
                                 timed with torch.bmm (hipBLASLt / rocBLAS fp32: a vendor-tuned GEMM, i.e. the
                                 best case for a hand-written batched GEMM stage)
  stages 1 and 3 (transforms)    HBM-bound: the transformed input V is 16/4 = 4x the input, the products M 4x the
                                 output, both fp32, each written once and read once; timed as plain device
                                 copies of the same byte counts (again the best case: a real transform kernel
                                 also gathers 4x4 patches with stride 2 and applies B^T d B / A^T m A)

Diagnostic tooling only: torch's GEMM is never on the product path.
usage: python tools/winograd_probe.py
"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ai_based_frame_interpolation_amd as P
import bench

dev = torch.device("cuda:0")
B, C, H, W = 8, 512, 135, 240
th, tw = (H + 1) // 2, (W + 1) // 2
tiles = B * th * tw


def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

# ---- the shipped direct conv of that layer: stage 7 (down3.3) of a B=8 1080p fp32 forward ----------
model = bench.make_bench_model("fp32").to(dev).eval()
f1 = torch.rand(B, 1, 1080, 1920, device=dev) * 2 - 1
f2 = torch.rand(B, 1, 1080, 1920, device=dev) * 2 - 1
model(f1, f2)
model._ctx.profile_enable(True)
for _ in range(2): model(f1, f2)
_, rows = model._ctx.profile_read()
model._ctx.profile_enable(False)
name, direct_ms, flops = rows[7]
del f1, f2, model
torch.cuda.empty_cache()
print(f"direct conv down3.3 (stage 7, {name}): {direct_ms:.3f} ms = {flops / direct_ms / 1e9:.1f} TFLOP/s of 157.3")

# ---- stage 2: 16 batched GEMMs ----------------------------------------------------------------------
V = torch.randn(16, tiles, C, device=dev)
U = torch.randn(16, C, C, device=dev)
gemm_s = timeit(lambda: torch.bmm(V, U))
gemm_flops = 16 * 2.0 * tiles * C * C
print(f"stage 2, 16 x [{tiles} x {C}] x [{C} x {C}] fp32 (torch.bmm): {gemm_s * 1e3:.3f} ms = "
      f"{gemm_flops / gemm_s / 1e12:.1f} TFLOP/s executed ({flops / gemm_s / 1e12:.1f} 'direct-conv' TFLOP/s)")

# ---- stages 1 and 3 as pure traffic: read x (B*C*H*W) + write V (16 * tiles * C); read M + write y ---
x = torch.empty(B * C * H * W, device=dev)
v_flat = V.view(-1)
v_dst = torch.empty_like(v_flat)
t1 = timeit(lambda: v_dst.copy_(v_flat))   # reads 2.14 GB, writes 2.14 GB
# traffic model: stage 1 = read x + write V; stage 3 = read M (= V-sized) + write y (= x-sized)
bytes_x, bytes_v = x.numel() * 4, v_flat.numel() * 4
bw = 2 * bytes_v / t1            # achieved copy bandwidth (read + write)
tr_s = 2 * (bytes_x + bytes_v) / bw
print(f"transforms as pure traffic: x {bytes_x / 1e9:.2f} GB, V / M {bytes_v / 1e9:.2f} GB each; device copy runs at "
      f"{bw / 1e12:.2f} TB/s -> stages 1 + 3 >= {tr_s * 1e3:.3f} ms")
total = gemm_s + tr_s
print(f"three-stage Winograd lower bound: {total * 1e3:.3f} ms vs direct {direct_ms:.3f} ms -> at most "
      f"{direct_ms / (total * 1e3):.2f}x on this layer (go threshold of the review: >= 1.5x)")
