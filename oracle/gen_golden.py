"""Generate tests/golden/*.npz by running the REAL reference model -- build container only.

TEST INFRASTRUCTURE.  Imports /root/reference/model/unet.py (read-only, never copied), loads
the seeded checkpoint of oracle.unet_oracle.make_seeded_state_dict into it with
load_state_dict (strict), puts it in .eval() like inference.py:97 does, and records
inputs + expected outputs as small fixtures.  The reference cannot travel to the GPU box;
these vectors (data only) do.

    python oracle/gen_golden.py            # writes tests/golden/
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402

REF_DIR = "/root/reference/model"
GOLD = os.path.join(ROOT, "tests", "golden")
WEIGHT_SEED = 1234


def strided_sample(t: torch.Tensor, n: int = 2048):
    flat = t.reshape(-1)
    step = max(1, flat.numel() // n)
    idx = torch.arange(0, flat.numel(), step)[:n]
    return idx.numpy().astype(np.int64), flat[idx].numpy().astype(np.float32)


RGB_WEIGHT_SEED = 77


def write_module_tree(model):
    """Attribute tree of the reference module (callers reach into
    `model.unet.inc.double_conv[0]`, `down1.maxpool_conv[1]`, `up1.up`): one line per
    named_modules() entry with the class name and the number of direct children."""
    with open(os.path.join(GOLD, "named_modules.txt"), "w") as f:
        for name, mod in model.named_modules():
            f.write(f"{name}\t{type(mod).__name__}\t{len(list(mod.children()))}\n")


def gen_rgb():
    """RGB 6->3 variant: the reference's own parametric
    `UNet(n_channels=6, n_classes=3, bilinear=True)` (unet.py:66) on cat([frame1, frame2]) -- the
    exact computation FrameInterpolationUNet.forward (unet.py:105-112) would do with that inner
    network.  State-dict keys of the inner UNet are the seeded checkpoint's minus the `unet.` prefix."""
    sys.path.insert(0, REF_DIR)
    from unet import UNet  # the reference class (unet.py:65)

    torch.set_num_threads(8)
    sd = O.make_seeded_state_dict(RGB_WEIGHT_SEED, n_channels=6, n_classes=3)
    inner = UNet(n_channels=6, n_classes=3, bilinear=True)
    print("rgb load_state_dict:", inner.load_state_dict(
        {k[len("unet."):]: v for k, v in sd.items()}, strict=True))
    inner.eval()
    for name, seed, b, h, w in (("rgb_b2_40x56", 31, 2, 40, 56), ("rgb_b1_33x47", 32, 1, 33, 47)):
        f1, f2 = O.make_frames(seed, b, h, w, c=3)
        with torch.no_grad():
            out = inner(torch.cat([f1, f2], dim=1))
        mine = O.unet_forward(sd, f1, f2)
        print(f"{name}: out {tuple(out.shape)} std {out.std():.4f} min {out.min():.3f} max {out.max():.3f} "
              f"|restatement-ref| {float((mine - out).abs().max()):.3e}")
        np.savez_compressed(os.path.join(GOLD, f"out_{name}.npz"), seed=seed,
                            weight_seed=RGB_WEIGHT_SEED, frame1=f1.numpy(), frame2=f2.numpy(),
                            out=out.numpy())


SSIM_CASES = (  # name, seed, B, C, H, W, value range, noise
    ("b1c1_32x48", 41, 1, 1, 32, 48, "unit", 0.05),
    ("b2c1_64x64", 42, 2, 1, 64, 64, "unit", 0.10),
    ("b1c3_33x47", 43, 1, 3, 33, 47, "unit", 0.05),     # odd sizes, C = 3 (window rebuilt, train.py:59-70)
    ("b3c3_17x31", 44, 3, 3, 17, 31, "sym", 0.20),      # [-1, 1] tensors (inference normalisation)
    ("b1c1_7x9", 45, 1, 1, 7, 9, "unit", 0.10),         # smaller than the 11x11 window: all padding
    ("b1c1_256x256", 46, 1, 1, 256, 256, "unit", 0.02),  # the reference's training size
    ("b2c1_135x240", 47, 2, 1, 135, 240, "unit", 0.30),  # spans several 64x16 tiles, ragged edges
)


def gen_convt():
    """The constructor's DEFAULT variant, bilinear=False (unet.py:42-44,66,99): ConvTranspose2d(k=2, s=2) decoder,
    31 037 057 parameters.  No reference caller constructs it (inference.py:77 passes bilinear=True), but it is what
    `FrameInterpolationUNet()` builds.  Fixtures: the 118-tensor state-dict schema and whole-net outputs of the REAL
    class with the seeded checkpoint of make_seeded_state_dict(bilinear=False), incl. odd sizes (F.pad after the
    transposed conv) and one per-block set of activations."""
    sys.path.insert(0, REF_DIR)
    from unet import FrameInterpolationUNet
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sd = O.make_seeded_state_dict(WEIGHT_SEED, bilinear=False)
    model = FrameInterpolationUNet()          # the default constructor
    assert model.unet.bilinear is False
    print("load_state_dict:", model.load_state_dict(sd, strict=True))
    model.eval()
    schema = [(k, tuple(v.shape), str(v.dtype)) for k, v in model.state_dict().items()]
    assert [(k, s) for k, s, _ in schema] == [(k, tuple(s)) for k, s, _ in O.state_dict_schema(bilinear=False)]
    with open(os.path.join(GOLD, "state_dict_schema_convt.txt"), "w") as f:
        for k, shp, dt in schema:
            f.write(f"{k}\t{','.join(map(str, shp))}\t{dt}\n")
    print("params:", sum(p.numel() for p in model.parameters()))
    for name, seed, b, h, w in (("b1_32x48", 21, 1, 32, 48), ("b2_17x31", 22, 2, 17, 31), ("b1_135x240", 23, 1, 135, 240),
                                ("b1_70x86", 24, 1, 70, 86)):
        f1, f2 = O.make_frames(seed, b, h, w)
        with torch.no_grad():
            out = model(f1, f2)
        mine = O.unet_forward(sd, f1, f2)
        print(f"convt {name}: out std {out.std():.4f} min {out.min():.3f} max {out.max():.3f} "
              f"|restatement-ref| {float((mine - out).abs().max()):.3e}")
        if "--convt-layers-only" not in sys.argv:
            np.savez_compressed(os.path.join(GOLD, f"out_convt_{name}.npz"), seed=seed, weight_seed=WEIGHT_SEED,
                                frame1=f1.numpy(), frame2=f2.numpy(), out=out.numpy())
    gen_convt_layers(model, sd)


def gen_convt_layers(model, sd):
    """Per-block activations of the default-constructed reference class (hooks on its modules, as for the bilinear
    variant in main()): the 18 conv+BN+ReLU outputs, the four pools, the four `up{k}.up` ConvTranspose2d outputs
    (unet.py:43,47 - before F.pad) and the head.  34x52 -> 17x26 -> 8x13 -> 4x6 -> 2x3: F.pad is live after up2
    (one column) and up3 (one row) and absent after up1 / up4."""
    f1, f2 = O.make_frames(25, 1, 34, 52)
    acts = {}

    def hook(name):
        def fn(_m, _inp, outp):
            acts[name] = outp.detach().clone()
        return fn

    handles = []
    for mod_name, mod in model.named_modules():
        if mod_name.endswith("double_conv.2") or mod_name.endswith("double_conv.5"):   # inplace ReLUs: see main()
            conv_idx = "0" if mod_name.endswith(".2") else "3"
            handles.append(mod.register_forward_hook(hook(mod_name.rsplit(".", 1)[0] + "." + conv_idx)))
        if mod_name.endswith("maxpool_conv.0"):
            handles.append(mod.register_forward_hook(hook(mod_name.split(".maxpool_conv")[0] + ".pool")))
        if mod_name.startswith("unet.up") and mod_name.endswith(".up"):
            assert isinstance(mod, torch.nn.ConvTranspose2d), mod
            handles.append(mod.register_forward_hook(hook(mod_name)))
        if mod_name == "unet.outc":
            handles.append(mod.register_forward_hook(hook("unet.outc")))
    with torch.no_grad():
        model(f1, f2)
    for hd in handles:
        hd.remove()
    assert len(acts) == 18 + 4 + 4 + 1, sorted(acts)
    taps = {}
    O.unet_forward(sd, f1, f2, taps)
    fix = {}
    for k, v in acts.items():
        d = float((taps[k] - v).abs().max())
        idx, vals = strided_sample(v, 1024)
        fix[f"{k}|idx"] = idx
        fix[f"{k}|val"] = vals
        fix[f"{k}|sum"] = np.float64(v.double().sum().item())
        fix[f"{k}|abssum"] = np.float64(v.double().abs().sum().item())
        fix[f"{k}|shape"] = np.array(v.shape, dtype=np.int64)
        print(f"  convt layer {k:42s} shape {tuple(v.shape)} |restatement-ref| {d:.2e}")
    np.savez_compressed(os.path.join(GOLD, "layers_convt_b1_34x52.npz"), frame1=f1.numpy(), frame2=f2.numpy(), **fix)


def make_ssim_pair(seed, b, c, h, w, rng_kind, noise):
    """Smooth structure + noise (so the variance terms are neither 0 nor dominated by noise)."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(0, 3.0, h).view(1, 1, h, 1)
    xs = torch.linspace(0, 4.0, w).view(1, 1, 1, w)
    ph = torch.rand(b, c, 1, 1, generator=g) * 6.28
    base = 0.5 + 0.35 * torch.sin(2.1 * ys + ph) * torch.cos(1.7 * xs - ph)
    img1 = (base + 0.08 * torch.rand(b, c, h, w, generator=g)).clamp(0, 1)
    img2 = (img1 + noise * (torch.rand(b, c, h, w, generator=g) - 0.5)).clamp(0, 1)
    if rng_kind == "sym":
        img1, img2 = img1 * 2 - 1, img2 * 2 - 1
    return img1.float().contiguous(), img2.float().contiguous()


def gen_ssim():
    """Gaussian-window SSIM / CombinedLoss fixtures from the reference's own classes (train.py:18-87).
    `train.py` imports cv2 at module level (train.py:7) but neither class touches it; cv2 is not
    installed in this image, so an EMPTY placeholder module is registered for the duration of the import
    (nothing in it is ever called; the values below come from torch ops alone)."""
    import types
    sys.path.insert(0, REF_DIR)
    placeholder = "cv2" not in sys.modules
    if placeholder:
        try:
            import cv2  # noqa: F401
            placeholder = False
        except ImportError:
            sys.modules["cv2"] = types.ModuleType("cv2")
    import train as ref_train  # the reference module (train.py)
    if placeholder:
        del sys.modules["cv2"]
    torch.set_num_threads(8)
    from oracle import metrics_oracle as MO
    for name, seed, b, c, h, w, kind, noise in SSIM_CASES:
        img1, img2 = make_ssim_pair(seed, b, c, h, w, kind, noise)
        with torch.no_grad():
            loss_avg = ref_train.SSIMLoss()(img1, img2)                       # 1 - ssim_map.mean()
            loss_per = ref_train.SSIMLoss(size_average=False)(img1, img2)     # [B]
            comb = ref_train.CombinedLoss()(img1, img2)
            mse = torch.nn.MSELoss()(img1, img2)
        mine = MO.ssim_gauss(img1, img2)
        mine64 = MO.ssim_gauss(img1, img2, dtype=torch.float64)
        print(f"ssim {name}: ssim {1 - loss_avg.item():.7f} combined {comb.item():.7f} "
              f"|restatement-ref| {abs((1 - loss_avg.item()) - mine.item()):.2e} "
              f"|fp64 restatement-ref| {abs((1 - loss_avg.item()) - mine64.item()):.2e}")
        np.savez_compressed(os.path.join(GOLD, f"ssim_gauss_{name}.npz"), seed=seed, img1=img1.numpy(),
                            img2=img2.numpy(), ssim_loss=np.float32(loss_avg.item()),
                            ssim_loss_per_sample=loss_per.numpy(), combined_loss=np.float32(comb.item()),
                            mse=np.float32(mse.item()))


def main():
    if "--ssim-only" in sys.argv:
        os.makedirs(GOLD, exist_ok=True)
        return gen_ssim()
    if "--convt-only" in sys.argv or "--convt-layers-only" in sys.argv:  # add the bilinear=False fixtures without re-recording the others
        os.makedirs(GOLD, exist_ok=True)
        return gen_convt()
    if "--rgb-only" in sys.argv:  # add the RGB fixtures without re-recording the others
        os.makedirs(GOLD, exist_ok=True)
        return gen_rgb()
    if "--modules-only" in sys.argv:
        sys.path.insert(0, REF_DIR)
        from unet import FrameInterpolationUNet
        return write_module_tree(FrameInterpolationUNet(bilinear=True))
    sys.path.insert(0, REF_DIR)
    from unet import FrameInterpolationUNet  # the reference class (unet.py:97)

    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    sd = O.make_seeded_state_dict(WEIGHT_SEED)
    model = FrameInterpolationUNet(bilinear=True)
    missing = model.load_state_dict(sd, strict=True)
    print("load_state_dict:", missing)
    model.eval()
    # the schema itself is a pinned fact: names, shapes, dtypes, order
    ref_sd = model.state_dict()
    schema = [(k, tuple(v.shape), str(v.dtype)) for k, v in ref_sd.items()]
    assert [s[0] for s in schema] == [s[0] for s in O.state_dict_schema()]
    with open(os.path.join(GOLD, "state_dict_schema.txt"), "w") as f:
        for k, shp, dt in schema:
            f.write(f"{k}\t{','.join(map(str, shp))}\t{dt}\n")
    nparams = sum(p.numel() for p in model.parameters())
    print("params:", nparams)
    write_module_tree(model)

    # ---- whole-net outputs, stored in full (small sizes) --------------------------------
    full_cases = [  # name, seed, B, H, W
        ("b1_32x48", 11, 1, 32, 48),
        ("b2_64x64", 12, 2, 64, 64),
        ("b1_17x31", 13, 1, 17, 31),     # odd sizes: floor-pool + asymmetric pad (unet.py:49-53)
        ("b1_16x16", 14, 1, 16, 16),     # minimum legal size
        ("b1_135x240", 15, 1, 135, 240),  # 1080p/8: same pad pattern as 1080p (135->67->...)
        ("b1_256x256", 0, 1, 256, 256),  # BASELINE config 1
    ]
    for name, seed, b, h, w in full_cases:
        f1, f2 = O.make_frames(seed, b, h, w)
        with torch.no_grad():
            out = model(f1, f2)
        mine = O.unet_forward(sd, f1, f2)
        print(f"{name}: out std {out.std():.4f} min {out.min():.3f} max {out.max():.3f} "
              f"|restatement-ref| {float((mine - out).abs().max()):.3e}")
        np.savez_compressed(
            os.path.join(GOLD, f"out_{name}.npz"),
            seed=seed, weight_seed=WEIGHT_SEED,
            frame1=f1.numpy(), frame2=f2.numpy(), out=out.numpy(),
        )

    # ---- per-layer activations for one small pair (hooks on the reference modules) -------
    f1, f2 = O.make_frames(11, 1, 32, 48)
    acts = {}

    def hook(name):
        def fn(_m, _inp, outp):
            acts[name] = outp.detach().clone()
        return fn

    handles = []
    for mod_name, mod in model.named_modules():
        # the ReLU after each BN is inplace, so the BN module's output tensor as seen after the
        # forward already holds conv->BN->ReLU; hook the ReLUs (indices 2 and 5) explicitly.
        if mod_name.endswith("double_conv.2") or mod_name.endswith("double_conv.5"):
            conv_idx = "0" if mod_name.endswith(".2") else "3"
            handles.append(mod.register_forward_hook(
                hook(mod_name.rsplit(".", 1)[0] + "." + conv_idx)))
        if mod_name.endswith("maxpool_conv.0"):
            handles.append(mod.register_forward_hook(
                hook(mod_name.split(".maxpool_conv")[0] + ".pool")))
        if mod_name == "unet.outc":
            handles.append(mod.register_forward_hook(hook("unet.outc")))
    with torch.no_grad():
        out = model(f1, f2)
    for hd in handles:
        hd.remove()
    taps = {}
    O.unet_forward(sd, f1, f2, taps)
    layer_fix = {}
    for k, v in acts.items():
        assert k in taps, k
        d = float((taps[k] - v).abs().max())
        idx, vals = strided_sample(v, 1024)
        layer_fix[f"{k}|idx"] = idx
        layer_fix[f"{k}|val"] = vals
        layer_fix[f"{k}|sum"] = np.float64(v.double().sum().item())
        layer_fix[f"{k}|abssum"] = np.float64(v.double().abs().sum().item())
        layer_fix[f"{k}|shape"] = np.array(v.shape, dtype=np.int64)
        print(f"  layer {k:42s} shape {tuple(v.shape)} |restatement-ref| {d:.2e}")
    np.savez_compressed(os.path.join(GOLD, "layers_b1_32x48.npz"),
                        frame1=f1.numpy(), frame2=f2.numpy(), **layer_fix)

    # ---- 1080p: strided sample + float64 sums only ---------------------------------------
    f1, f2 = O.make_frames(16, 1, 1080, 1920)
    t0 = time.time()
    with torch.no_grad():
        out = model(f1, f2)
    print(f"1080p reference forward: {time.time() - t0:.1f} s")
    idx, vals = strided_sample(out, 4096)
    u8 = O.postprocess_tensor(out)
    np.savez_compressed(
        os.path.join(GOLD, "out_b1_1080x1920_sample.npz"),
        seed=16, weight_seed=WEIGHT_SEED, idx=idx, val=vals,
        sum=np.float64(out.double().sum().item()),
        abssum=np.float64(out.double().abs().sum().item()),
        u8_hist=np.bincount(u8.reshape(-1), minlength=256).astype(np.int64),
    )

    # ---- postprocess + PSNR fixture (inference.py:54-61; evaluation.py:194-205) -----------
    f1, f2 = O.make_frames(12, 2, 64, 64)
    with torch.no_grad():
        out = model(f1[:1], f2[:1])
    u8 = O.postprocess_tensor(out)
    gt = O.postprocess_tensor(0.5 * (f1[:1] + f2[:1]))
    np.savez_compressed(os.path.join(GOLD, "post_b1_64x64.npz"),
                        out=out.numpy(), u8=u8, gt_u8=gt, psnr=np.float64(O.psnr_u8(gt, u8)))
    gen_rgb()
    gen_ssim()
    print("done; files:", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    main()
