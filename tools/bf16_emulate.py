#!/usr/bin/env python3
"""CPU emulation of the bf16 path's rounding points (build container, no GPU): which of them decide the
north-star criterion |PSNR_bf16 - PSNR_fp32| <= 0.05 dB on an interpolating checkpoint?

Rounding points of the HIP bf16 path: BN scale folded into the conv weights, the
product rounded to bf16 (round-to-nearest or the error-feedback rounding of fiunet.hip); every stored
activation rounded to bf16 (RNE) - the stem output as inc.3's MFMA operand, conv outputs 1..16, the
interpolated (upsampled) values; the last conv's output and the 1x1 head stay fp32.  This script restates
that in torch on the CPU (test/diagnostic infrastructure: it imports oracle/) with one switch per layer
and rounding point, so that a single layer's contribution can be isolated.  Not bit-identical to the
kernels (summation order), statistically the same.

    python tools/bf16_emulate.py [--size 256] [--scenes 5] [--ckpts 3]
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_oracle as O  # noqa: E402
from ai_based_frame_interpolation_amd import synthetic as S  # noqa: E402

PREFIXES = ["unet.inc"] + [f"unet.down{k}.maxpool_conv.1" for k in (1, 2, 3, 4)] + [f"unet.up{k}.conv" for k in (1, 2, 3, 4)]
LAYERS = [(p, ci, bi) for p in PREFIXES for ci, bi in ((0, 1), (3, 4))]  # 18 convs in state-dict order


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def round_weights(w, mode):
    """w: [Cout, Cin, 3, 3] fp32 (BN scale already folded)."""
    if mode == "exact":
        return w
    if mode == "rne":
        return bf16(w)
    if mode == "split":   # hi + lo: two bf16 numbers per weight (~2^-17 relative)
        h = bf16(w)
        return h + bf16(w - h)
    assert mode == "feedback"
    # vectorised over filters: sequential over the K = Cin*9 weights of a filter (ci outer, taps inner)
    co = w.shape[0]
    flat = w.reshape(co, -1).double()
    u = w.reshape(co, -1).contiguous().view(torch.int32)
    lo = (u & ~0xFFFF).view(torch.float32).double()           # toward zero
    ulp = (torch.abs(lo) * 2.0 ** -7).clamp(min=0)            # not exact at binade edges; use next-after instead
    hi_bits = ((u >> 16) + 1) << 16
    hi = hi_bits.view(torch.float32).double()                 # away from zero
    e0, e1 = flat - lo, flat - hi
    out = torch.empty_like(flat)
    carry = torch.zeros(co, dtype=torch.float64)
    for k in range(flat.shape[1]):
        rep = e0[:, k] == 0
        pick0 = (carry + e0[:, k]).abs() <= (carry + e1[:, k]).abs()
        pick0 = pick0 | rep
        out[:, k] = torch.where(pick0, lo[:, k], hi[:, k])
        carry = carry + torch.where(pick0, e0[:, k], e1[:, k])
    return out.float().reshape(w.shape)


class Emu:
    def __init__(self, sd, wmode="feedback", act=True, wmode_by_layer=None, act_by_layer=None, up_round=True):
        self.sd, self.up_round = sd, up_round
        self.w, self.shift, self.act = [], [], []
        for i, (p, ci, bi) in enumerate(LAYERS):
            g = sd[f"{p}.double_conv.{bi}.weight"]
            b = sd[f"{p}.double_conv.{bi}.bias"]
            m = sd[f"{p}.double_conv.{bi}.running_mean"]
            v = sd[f"{p}.double_conv.{bi}.running_var"]
            sc = g / torch.sqrt(v + O.BN_EPS)
            w = sd[f"{p}.double_conv.{ci}.weight"] * sc.view(-1, 1, 1, 1)
            mode = (wmode_by_layer or {}).get(i, wmode)
            if i == 0:
                mode = "exact"  # the stem is evaluated with split operands: fp32-grade
            self.w.append(round_weights(w, mode))
            self.shift.append((b - m * sc).view(1, -1, 1, 1))
            a = (act_by_layer or {}).get(i, act)
            self.act.append(a if i != 17 else False)  # up4.3 feeds the fp32 head from registers

    def conv(self, i, x):
        y = F.relu(F.conv2d(x, self.w[i], padding=1) + self.shift[i])
        return bf16(y) if self.act[i] else y

    @torch.no_grad()
    def forward(self, f1, f2):
        x = torch.cat([f1, f2], 1)
        cur = self.conv(1, self.conv(0, x))
        skips = [cur]
        li = 2
        for _ in range(4):
            cur = F.max_pool2d(cur, 2)
            cur = self.conv(li + 1, self.conv(li, cur))
            li += 2
            skips.append(cur)
        for skip in (skips[3], skips[2], skips[1], skips[0]):
            up = F.interpolate(cur, scale_factor=2, mode="bilinear", align_corners=True)
            if self.up_round and self.act[li - 1]:
                up = bf16(up)
            dy, dx = skip.shape[2] - up.shape[2], skip.shape[3] - up.shape[3]
            up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
            cur = self.conv(li + 1, self.conv(li, torch.cat([skip, up], 1)))
            li += 2
        return F.conv2d(cur, self.sd["unet.outc.conv.weight"], self.sd["unet.outc.conv.bias"])


def psnr_delta(sd, emu, scenes, h, w, refs=None):
    out = []
    for s in scenes:
        a, truth, c = S.triplet(h, w, device="cpu", seed=s)
        fa, fc = O.preprocess_array(a.numpy()), O.preprocess_array(c.numpy())
        ref = O.unet_forward(sd, fa, fc)
        p_ref = O.psnr_u8(truth.numpy(), O.postprocess_tensor(ref))
        y = emu.forward(fa, fc)
        p_emu = O.psnr_u8(truth.numpy(), O.postprocess_tensor(y))
        e, d = (y - ref).flatten().double(), (ref - (0.5 * (fa + fc))).flatten().double()
        gain = float((e @ d) / (d @ d))
        out.append((p_emu - p_ref, p_ref, float(e.norm() / d.norm()), gain, float(e.mean())))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--scenes", type=int, default=5)
    ap.add_argument("--ckpts", type=int, default=3)
    ap.add_argument("--by-layer", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(8)
    h = w = args.size
    scenes = list(range(3, 3 + args.scenes))
    for ck in range(args.ckpts):
        sd = O.make_interpolating_state_dict(seed=4321 + 1000 * ck)
        print(f"== checkpoint seed {4321 + 1000 * ck}")
        variants = [("rne weights + bf16 act", dict(wmode="rne")),
                    ("feedback weights + bf16 act", dict(wmode="feedback")),
                    ("exact weights + bf16 act", dict(wmode="exact")),
                    ("feedback weights, fp32 act", dict(wmode="feedback", act=False)),
                    ("split weights + bf16 act", dict(wmode="split"))]
        for name, kw in variants:
            r = psnr_delta(sd, Emu(sd, **kw), scenes, h, w)
            print(f"  {name:34s} dPSNR " + " ".join(f"{x[0]:+.4f}" for x in r) +
                  f" | rel err {np.mean([x[2] for x in r]):.4f} gain {np.mean([x[3] for x in r]):+.5f} "
                  f"mean {np.mean([x[4] for x in r]):+.2e}  (PSNR ref {r[0][1]:.2f})")
        if args.by_layer:
            for i, (p, ci, _) in enumerate(LAYERS):
                if i == 0:
                    continue
                r = psnr_delta(sd, Emu(sd, wmode="exact", act=False, wmode_by_layer={i: "feedback"}), scenes[:2], h, w)
                r2 = psnr_delta(sd, Emu(sd, wmode="exact", act=False, act_by_layer={i: True}), scenes[:2], h, w)
                print(f"  only layer {i:2d} {p}.{ci}: weights dPSNR {r[0][0]:+.4f} {r[1][0]:+.4f} gain {r[0][3]:+.5f} | "
                      f"act dPSNR {r2[0][0]:+.4f} {r2[1][0]:+.4f} gain {r2[0][3]:+.5f}")


if __name__ == "__main__":
    main()
