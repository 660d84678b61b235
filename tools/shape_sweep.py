"""One-off wider sweep (GPU box): random batch / frame shapes, the three precisions, HIP path vs the CPU oracle,
plus default vs gather-upsample and vs the unfused path (close, <= 1e-2 / 3e-2 relative: since round 6 the gather-upsample
option changes the K cut of a small problem's concat convs - in-workgroup cut on the materialised half against a slab cut
on the fused gather - so bit-equality of the two holds only where no layer is cut: tests/test_gpu_configs.py pins it there;
the unfused path's head reads the bf16-rounded last activation and its stem is the standalone fp32 kernel).
usage: python tools/shape_sweep.py [n_random=40]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import unet_oracle as O
import ai_based_frame_interpolation_amd as P
dev = torch.device("cuda:0")
sd = O.make_seeded_state_dict(1234)
m = P.FrameInterpolationUNet(bilinear=True); m.load_state_dict(sd); m = m.to(dev).eval()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
shapes = [(int(rng.integers(1, 5)), int(rng.integers(16, 320)), int(rng.integers(16, 420))) for _ in range(n)]
shapes += [(2, 270, 480), (3, 540, 960), (9, 135, 241), (1, 1080, 1920), (5, 272, 528)]
worst32 = worst16 = worstx2 = 0.0
t0 = time.time()
for i, (b, h, w) in enumerate(shapes):
    f1, f2 = O.make_frames(9000 + i, b, h, w)
    ref = O.unet_forward(sd, f1, f2)
    g1, g2 = f1.to(dev), f2.to(dev)
    m.precision = "fp32"; m.set_options()
    o32 = m(g1, g2).cpu()
    d = (o32 - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    m.precision = "bf16x2"
    ox2 = m(g1, g2).cpu()
    dx2 = (ox2 - ref).abs().max().item()
    worstx2 = max(worstx2, dx2 / max(1.0, ref.abs().max().item()))
    m.precision = "bf16"
    o16 = m(g1, g2)
    rel = ((o16.cpu() - ref).norm() / ref.norm()).item()
    m.set_options(gather_upsample=True); og = m(g1, g2)
    m.set_options(unfused=True); ou = m(g1, g2)
    m.set_options()
    eq = ((og - o16).norm() / o16.norm()).item() <= 1e-2 and ((ou - o16).norm() / o16.norm()).item() <= 3e-2
    worst32, worst16 = max(worst32, d), max(worst16, rel)
    okx2 = dx2 <= 1e-3 and dx2 <= 2e-4 * max(1.0, ref.abs().max().item())   # the fp32 contract, and relative
    flag = "" if (d <= 1e-4 and rel <= 2.5e-2 and eq and okx2 and torch.isfinite(o16).all()) else "  <-- FAIL"
    print(f"{b}x{h}x{w}: fp32 rel-max {d:.2e}  bf16x2 max-abs {dx2:.2e}  bf16 rel-L2 {rel:.3e}  gather-upsample and unfused close to default: {eq}{flag}", flush=True)
    if flag: sys.exit(1)
print(f"SWEEP OK: {len(shapes)} shapes, worst fp32 {worst32:.2e}, worst bf16x2 (relative to max(1, |ref|)) {worstx2:.2e}, worst bf16 {worst16:.3e}, {time.time() - t0:.0f} s")
