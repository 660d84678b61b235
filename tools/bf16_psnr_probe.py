"""Where does the bf16 PSNR delta on the interpolating checkpoint come from?  Uses the fp32 HIP path
(== the CPU reference to 1e-6) as the reference on the GPU box.
usage: python tools/bf16_psnr_probe.py [h w]"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import unet_oracle as O
import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import synthetic as S
dev = torch.device("cuda:0")
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
sd = O.make_interpolating_state_dict()
m = P.FrameInterpolationUNet(bilinear=True); m.load_state_dict(sd); m = m.to(dev).eval()
for seed in (3, 4, 5):
    a, t, c = S.triplet(h, w, device="cpu", seed=seed)
    fa = (a.float() / 255 * 2 - 1)[None, None].to(dev); fc = (c.float() / 255 * 2 - 1)[None, None].to(dev)
    res = {}
    for name, prec, opt in (("fp32", "fp32", {}), ("bf16", "bf16", {}), ("bf16 unfused", "bf16", dict(unfused=True))):
        m.precision = prec; m.set_options(**opt)
        res[name] = m(fa, fc)[0, 0].cpu()
    m.set_options()
    tr = t.numpy()
    def u8(x): return O.postprocess_tensor(x[None, None])[0, 0] if hasattr(O, "postprocess_tensor") else None
    line = [f"seed {seed}:"]
    ref = res["fp32"]
    for k, v in res.items():
        img = ((v.clamp(-1, 1) + 1) / 2 * 255).numpy().astype(np.uint8)
        ps = O.psnr_u8(tr, img)
        d = (v - ref)
        # gain of v against ref around the linear part: least-squares slope of (v - lin) on (ref - lin)
        lin = (fa[0, 0].cpu() + fc[0, 0].cpu()) / 2
        rr, vv = (ref - lin).flatten().double(), (v - lin).flatten().double()
        gain = float((rr * vv).sum() / (rr * rr).sum())
        line.append(f"{k}: PSNR {ps:.4f} dB, mean(d) {float(d.mean()):+.2e}, rms(d) {float(d.pow(2).mean().sqrt()):.2e}, deep-part gain {gain:.4f} (rms {float(rr.pow(2).mean().sqrt()):.3f})")
    print("\n   ".join(line), flush=True)
