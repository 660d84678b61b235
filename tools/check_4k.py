"""Manual one-off (GPU box): 4K single pair, HIP path vs the CPU oracle."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import unet_oracle as O
import ai_based_frame_interpolation_amd as P
dev = torch.device("cuda:0")
sd = O.make_seeded_state_dict(1234)
m = P.FrameInterpolationUNet(bilinear=True); m.load_state_dict(sd); m = m.to(dev).eval()
f1, f2 = O.make_frames(5, 1, 2160, 3840)
t0 = time.time(); ref = O.unet_forward(sd, f1, f2); print("cpu oracle 4K: %.1f s" % (time.time() - t0))
for prec in ("fp32", "bf16"):
    m.precision = prec
    out = m(f1.to(dev), f2.to(dev)); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): out = m(f1.to(dev), f2.to(dev))
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    d = (out.cpu() - ref)
    print(f"4K {prec}: max|d| {d.abs().max():.3e} rel-L2 {d.norm() / ref.norm():.3e}  {dt * 1e3:.1f} ms/pair")
