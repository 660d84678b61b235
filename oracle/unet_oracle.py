"""CPU oracle for the UNet frame-pair forward -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / the timed CPU baseline.  The shipped path
(ai_based_frame_interpolation_amd) never imports anything under oracle/.

This is a functional (table-driven) restatement in plain PyTorch fp32 CPU ops of the
algorithm in the reference's model/unet.py.  It is pinned against outputs of the real
reference (imported from /root/reference in the build container by oracle/gen_golden.py)
through the committed fixtures in tests/golden/ -- see tests/test_oracle.py.

Reference lines restated (all in /root/reference/model/unet.py):
  DoubleConv  (conv3x3 pad1 no-bias -> BatchNorm2d eval -> ReLU) x2      unet.py:5-21
  Down        MaxPool2d(2) -> DoubleConv                                 unet.py:23-33
  Up          bilinear x2 align_corners=True, F.pad to the skip's size,
              cat([skip, up]), DoubleConv(in, out, mid=in//2)            unet.py:35-55
  OutConv     conv1x1 with bias                                          unet.py:57-63
  UNet        wiring 64-128-256-512-512 / 256-128-64-64                  unet.py:65-95
  FrameInterpolationUNet.forward  cat([frame1, frame2], dim=1)           unet.py:97-112
Pre/post-processing restated from /root/reference/model/inference.py:31-39 and :54-61;
PSNR from /root/reference/model/evaluation.py:194-205 (scikit-image definition,
data_range=255, restated as 10*log10(255^2/MSE) in float64).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default (unet.py:13,16)

# (state-dict prefix, in_channels, mid_channels, out_channels).  bilinear=True is the only variant any
# reference caller constructs (inference.py:77): factor 2, Up's DoubleConv has mid = in // 2 (unet.py:41,77-81).
# bilinear=False is the constructor's DEFAULT (unet.py:66,99): factor 1, down4 goes to 1024 channels, Up is
# ConvTranspose2d(in, in // 2, 2, 2) + DoubleConv(in, out) with mid = out (unet.py:42-44).  `cin0` is n_channels.


def double_conv_table(n_channels: int = 2, bilinear: bool = True):
    if not bilinear:
        return [
            ("unet.inc", n_channels, 64, 64),
            ("unet.down1.maxpool_conv.1", 64, 128, 128),
            ("unet.down2.maxpool_conv.1", 128, 256, 256),
            ("unet.down3.maxpool_conv.1", 256, 512, 512),
            ("unet.down4.maxpool_conv.1", 512, 1024, 1024),
            ("unet.up1.conv", 1024, 512, 512),
            ("unet.up2.conv", 512, 256, 256),
            ("unet.up3.conv", 256, 128, 128),
            ("unet.up4.conv", 128, 64, 64),
        ]
    return [
        ("unet.inc", n_channels, 64, 64),
        ("unet.down1.maxpool_conv.1", 64, 128, 128),
        ("unet.down2.maxpool_conv.1", 128, 256, 256),
        ("unet.down3.maxpool_conv.1", 256, 512, 512),
        ("unet.down4.maxpool_conv.1", 512, 512, 512),
        ("unet.up1.conv", 1024, 512, 256),
        ("unet.up2.conv", 512, 256, 128),
        ("unet.up3.conv", 256, 128, 64),
        ("unet.up4.conv", 128, 64, 64),
    ]


def state_dict_schema(n_channels: int = 2, n_classes: int = 1, bilinear: bool = True):
    """Ordered (name, shape, dtype) list of the reference state-dict (110 tensors; bilinear=False:
    118 - every Up block's ConvTranspose2d `up.weight [in, in // 2, 2, 2]`, `up.bias [in // 2]` come first)."""
    out = []
    for prefix, cin, mid, cout in double_conv_table(n_channels, bilinear):
        if not bilinear and prefix.startswith("unet.up"):
            blk = prefix[:-len(".conv")]
            out.append((f"{blk}.up.weight", (cin, cin // 2, 2, 2), torch.float32))
            out.append((f"{blk}.up.bias", (cin // 2,), torch.float32))
        for conv_i, bn_i, ci, co in ((0, 1, cin, mid), (3, 4, mid, cout)):
            out.append((f"{prefix}.double_conv.{conv_i}.weight", (co, ci, 3, 3), torch.float32))
            out.append((f"{prefix}.double_conv.{bn_i}.weight", (co,), torch.float32))
            out.append((f"{prefix}.double_conv.{bn_i}.bias", (co,), torch.float32))
            out.append((f"{prefix}.double_conv.{bn_i}.running_mean", (co,), torch.float32))
            out.append((f"{prefix}.double_conv.{bn_i}.running_var", (co,), torch.float32))
            out.append((f"{prefix}.double_conv.{bn_i}.num_batches_tracked", (), torch.int64))
    out.append(("unet.outc.conv.weight", (n_classes, 64, 1, 1), torch.float32))
    out.append(("unet.outc.conv.bias", (n_classes,), torch.float32))
    return out


def make_seeded_state_dict(seed: int = 1234, n_channels: int = 2, n_classes: int = 1, bilinear: bool = True):
    """Deterministic non-trivial checkpoint: He-scaled conv weights, randomised BN affine and
    running statistics (a fresh model has identity BN and ~constant output, which would make an
    absolute-tolerance parity test vacuous -- SURVEY section 7 'Fixture design').  Only plain
    torch CPU RNG calls, so the same tensors can be rebuilt anywhere from the seed."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape, dtype in state_dict_schema(n_channels, n_classes, bilinear):
        if dtype == torch.int64:
            sd[name] = torch.tensor(7, dtype=torch.int64)
        elif name.endswith(".up.weight"):   # ConvTranspose2d [in, out, 2, 2]: every output is a sum over `in` terms
            sd[name] = torch.randn(shape, generator=g) * math.sqrt(1.0 / shape[0])
        elif name.endswith(".up.bias"):
            sd[name] = torch.randn(shape, generator=g) * 0.1
        elif name.endswith("conv.weight") and len(shape) == 4 and shape[2] == 1:
            sd[name] = torch.randn(shape, generator=g) * (0.2 / math.sqrt(shape[1]))
        elif len(shape) == 4:
            fan_in = shape[1] * 9
            sd[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith("running_var"):
            sd[name] = torch.rand(shape, generator=g) * 1.0 + 0.5
        elif name.endswith("running_mean"):
            sd[name] = torch.randn(shape, generator=g) * 0.2
        elif name.endswith("conv.bias"):
            sd[name] = torch.randn(shape, generator=g) * 0.1
        elif name.endswith(".weight"):  # BN gamma
            sd[name] = torch.rand(shape, generator=g) * 1.0 + 0.6
        elif name.endswith(".bias"):  # BN beta
            sd[name] = torch.randn(shape, generator=g) * 0.25
        else:
            raise AssertionError(name)
    return sd


def make_interpolating_state_dict(seed: int = 4321, n_channels: int = 2, n_classes: int = 1,
                                  perturb: float = 0.03, blend=(0.5, 0.5)):
    """A checkpoint that actually interpolates, so that PSNR against a ground-truth middle frame
    means something (a random-init network sits ~15 dB from any truth, which makes
    |PSNR_a - PSNR_b| <= 0.05 dB vacuous).  Analytic part: the stem splits each input channel c
    into relu(+x_c) and relu(-x_c) (centre tap, identity BatchNorm); those 2*n_channels feature
    maps are carried unchanged through inc's second conv, the x1 skip, and up4's two convs
    (identity centre taps; ReLU is the identity on non-negative maps); the 1x1 head recombines
    them into 0.5 * (frame1 + frame2).  Everything else is the seeded random network of
    make_seeded_state_dict, which runs through all 18 convs and is added to the output through
    the remaining head weights, scaled so that it perturbs the result by ~`perturb` rms: every
    layer and both skip / upsampled halves of every concat contribute to the output.
    The carried maps receive no other inputs and the random maps may read them.
    `blend` = (weight of frame1, weight of frame2) of the analytic part; (0.5, 0.5) is the symmetric
    interpolator, anything else is a checkpoint that does NOT blend symmetrically (a trained network
    need not), e.g. (0.7, 0.3) or the pure copy (1, 0)."""
    sd = make_seeded_state_dict(seed, n_channels, n_classes)
    cf = n_classes
    assert n_channels == 2 * cf, "frame-pair network: n_channels == 2 * n_classes"
    nk = 2 * n_channels  # carried maps: (+,-) of every input channel
    one_m_eps = 1.0 - BN_EPS

    def identity_bn(prefix, bn_i, rows):
        sd[f"{prefix}.double_conv.{bn_i}.weight"][rows] = 1.0
        sd[f"{prefix}.double_conv.{bn_i}.bias"][rows] = 0.0
        sd[f"{prefix}.double_conv.{bn_i}.running_mean"][rows] = 0.0
        sd[f"{prefix}.double_conv.{bn_i}.running_var"][rows] = one_m_eps

    rows = slice(0, nk)
    w = sd["unet.inc.double_conv.0.weight"]
    w[rows] = 0.0
    for c in range(n_channels):
        w[2 * c, c, 1, 1] = 1.0
        w[2 * c + 1, c, 1, 1] = -1.0
    identity_bn("unet.inc", 1, rows)
    for prefix, conv_i, bn_i in (("unet.inc", 3, 4), ("unet.up4.conv", 0, 1), ("unet.up4.conv", 3, 4)):
        w = sd[f"{prefix}.double_conv.{conv_i}.weight"]
        w[rows] = 0.0
        for k in range(nk):
            w[k, k, 1, 1] = 1.0  # up4.conv.0: input channel k of cat([x1, up]) is x1's channel k
        identity_bn(prefix, bn_i, rows)
    hw = sd["unet.outc.conv.weight"]
    hw *= perturb / 0.25  # the seeded head gives ~0.25 rms on the seeded features
    hw[:, :nk] = 0.0
    for o in range(cf):  # output channel o = blend of channel o of frame1 and of frame2
        for c, wgt in ((o, float(blend[0])), (cf + o, float(blend[1]))):
            hw[o, 2 * c, 0, 0] = wgt
            hw[o, 2 * c + 1, 0, 0] = -wgt
    sd["unet.outc.conv.bias"].zero_()
    return sd


def make_frames(seed: int, b: int, h: int, w: int, c: int = 1):
    """Uniform [-1,1] synthetic frame pair, seeded (SURVEY 8d 'Config 1/2')."""
    g = torch.Generator().manual_seed(seed)
    f1 = torch.rand(b, c, h, w, generator=g) * 2.0 - 1.0
    f2 = torch.rand(b, c, h, w, generator=g) * 2.0 - 1.0
    return f1, f2


def _conv_bn_relu(x, sd, prefix, conv_i, bn_i):
    x = F.conv2d(x, sd[f"{prefix}.double_conv.{conv_i}.weight"], bias=None, padding=1)
    x = F.batch_norm(
        x,
        sd[f"{prefix}.double_conv.{bn_i}.running_mean"],
        sd[f"{prefix}.double_conv.{bn_i}.running_var"],
        sd[f"{prefix}.double_conv.{bn_i}.weight"],
        sd[f"{prefix}.double_conv.{bn_i}.bias"],
        training=False,
        eps=BN_EPS,
    )
    return F.relu(x)


def _double_conv(x, sd, prefix, taps):
    x = _conv_bn_relu(x, sd, prefix, 0, 1)
    if taps is not None:
        taps[f"{prefix}.double_conv.0"] = x
    x = _conv_bn_relu(x, sd, prefix, 3, 4)
    if taps is not None:
        taps[f"{prefix}.double_conv.3"] = x
    return x


def upsample_pad_concat(x_low, x_skip, up_weight=None, up_bias=None, taps=None, tap_name=None):
    """unet.py:46-54 -- bilinear x2 (align_corners=True), or ConvTranspose2d(k=2, s=2) when the checkpoint has
    one (bilinear=False, unet.py:42-44); asymmetric zero pad, cat([skip, up]).  taps[tap_name] = `self.up(x1)`
    (unet.py:47), i.e. BEFORE F.pad."""
    if up_weight is not None:
        up = F.conv_transpose2d(x_low, up_weight, up_bias, stride=2)
    else:
        up = F.interpolate(x_low, scale_factor=2, mode="bilinear", align_corners=True)
    if taps is not None and tap_name:
        taps[tap_name] = up
    dy = x_skip.shape[2] - up.shape[2]
    dx = x_skip.shape[3] - up.shape[3]
    up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return torch.cat([x_skip, up], dim=1)


@torch.no_grad()
def unet_forward(sd, frame1, frame2, taps=None):
    """FrameInterpolationUNet.forward in eval mode (unet.py:105-112 -> unet.py:84-95)."""
    x = torch.cat([frame1, frame2], dim=1)
    x1 = _double_conv(x, sd, "unet.inc", taps)
    skips = [x1]
    cur = x1
    for k in (1, 2, 3, 4):
        cur = F.max_pool2d(cur, 2)
        if taps is not None:
            taps[f"unet.down{k}.pool"] = cur
        cur = _double_conv(cur, sd, f"unet.down{k}.maxpool_conv.1", taps)
        skips.append(cur)
    for k, skip in zip((1, 2, 3, 4), (skips[3], skips[2], skips[1], skips[0])):
        cat = upsample_pad_concat(cur, skip, sd.get(f"unet.up{k}.up.weight"), sd.get(f"unet.up{k}.up.bias"), taps,
                                  f"unet.up{k}.up")
        if taps is not None:
            taps[f"unet.up{k}.cat"] = cat
        cur = _double_conv(cat, sd, f"unet.up{k}.conv", taps)
    out = F.conv2d(cur, sd["unet.outc.conv.weight"], sd["unet.outc.conv.bias"])
    if taps is not None:
        taps["unet.outc"] = out
    return out


def preprocess_array(gray_u8: np.ndarray) -> torch.Tensor:
    """inference.py:31-39 without the cv2 read/resize: uint8 [H,W] -> fp32 [1,1,H,W] in [-1,1]."""
    img = gray_u8.astype(np.float32) / 255.0
    img = 2.0 * img - 1.0
    return torch.from_numpy(img).unsqueeze(0).unsqueeze(0)


def postprocess_tensor(t: torch.Tensor) -> np.ndarray:
    """inference.py:54-61: (x+1)/2, clamp[0,1], *255, TRUNCATING uint8 cast."""
    img = (t + 1.0) / 2.0
    img = torch.clamp(img, 0.0, 1.0)
    arr = img.squeeze().cpu().numpy()
    return (arr * 255).astype(np.uint8)


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    """evaluation.py:194-205 (skimage peak_signal_noise_ratio, data_range=255)."""
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return float(10.0 * np.log10(255.0 ** 2 / mse))


def conv_flops(h: int, w: int, n_channels: int = 2, n_classes: int = 1) -> float:
    """2*MAC of all convolutions for one frame pair at HxW."""
    sizes = [(h, w)]
    for _ in range(4):
        sizes.append((sizes[-1][0] // 2, sizes[-1][1] // 2))
    level = [0, 1, 2, 3, 4, 3, 2, 1, 0]
    total = 0.0
    for (prefix, cin, mid, cout), lv in zip(double_conv_table(n_channels), level):
        hh, ww = sizes[lv]
        total += 2.0 * hh * ww * 9 * (cin * mid + mid * cout)
    total += 2.0 * h * w * 64 * n_classes
    return total
