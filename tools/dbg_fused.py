import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import unet_oracle as O
import ai_based_frame_interpolation_amd as P
dev = torch.device("cuda:0")
sd = O.make_seeded_state_dict(1234)
m = P.FrameInterpolationUNet(bilinear=True); m.load_state_dict(sd); m = m.to(dev).eval()
f1, f2 = O.make_frames(21, 2, 50, 70)
taps = {}; O.unet_forward(sd, f1, f2, taps)
for prec in ("fp32",):
    m.precision = prec
    m.set_options(unfused=False); A, _ = m.debug_activations(f1.to(dev), f2.to(dev))
    m.set_options(unfused=True); Bv, _ = m.debug_activations(f1.to(dev), f2.to(dev))
    for k in A:
        d = (A[k] - Bv[k]).abs()
        if d.max() > 0:
            idx = (d > 0).nonzero()
            print(k, "ndiff", idx.shape[0], "of", d.numel(), "max", d.max().item())
            print("  batch", idx[:, 0].unique().tolist(), "ch range", idx[:, 1].min().item(), idx[:, 1].max().item(),
                  "y", idx[:, 2].unique().tolist(), "x", idx[:, 3].unique().tolist())
            r = taps[k].to(dev)
            print("  fused-vs-oracle", (A[k] - r).abs().max().item(), "unfused-vs-oracle", (Bv[k] - r).abs().max().item())
            break
