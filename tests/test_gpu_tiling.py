"""GPU (MI355X): spatial tiling.  A forward cut into horizontal strips with a
112-row halo, each strip run through `fiunet_forward_strip` (whole-image upsample coordinates), must
reproduce the un-tiled forward.  Where both take the same kernel path the match is bit for bit;
very small bands can be K-split differently from the whole image (a different but deterministic
fp32 summation order), so the bound there is the fp32 contract with a 1e-5 relative margin.
The un-tiled forward itself is pinned to the reference by test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import tiling
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, seeded_sd):
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(seeded_sd)
    return m.to(dev).eval()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("b,h,w,n", [(1, 640, 96, 4), (2, 1080, 64, 3), (1, 1000, 130, 2), (1, 2160, 256, 4)])
def test_tiled_equals_untiled(model, dev, prec, b, h, w, n):
    f1, f2 = O.make_frames(21, b, h, w)
    f1, f2 = f1.to(dev), f2.to(dev)
    model.precision = prec
    model.set_options()
    whole = model(f1, f2)
    tiled = tiling.forward_tiled(model.forward_strip, f1, f2, n)
    scale = max(1.0, whole.abs().max().item())
    d = (tiled - whole).abs().max().item()
    if prec == "fp32":
        assert d <= 1e-5 * scale, d
    else:
        # bf16 storage: a differently K-split deep layer flips bf16 roundings, which then propagate;
        # the bound is the bf16 path's own contract vs the reference (test_gpu_parity.py)
        rel = ((tiled - whole).norm() / whole.norm()).item()
        assert rel <= 2e-2 and d <= 4e-2 * scale, (rel, d)


def test_tiled_4k_bitwise_and_vs_halo_free_cut(model, dev):
    """2160x3840 bf16, 4 strips (origins 0/544/1088/1632): bit-identical to the un-tiled forward,
    and a cut WITHOUT halo is visibly different (the test would notice a no-op implementation)."""
    f1, f2 = O.make_frames(5, 1, 2160, 3840)
    f1, f2 = f1.to(dev), f2.to(dev)
    model.precision = "bf16"
    model.set_options()
    whole = model(f1, f2)
    tiled = tiling.forward_tiled(model.forward_strip, f1, f2, 4)
    assert torch.equal(tiled, whole)
    plan = tiling.strip_plan(2160, 4)
    s = plan[1]
    naked = model.forward_strip(f1[..., s.core0:s.core1, :].contiguous(), f2[..., s.core0:s.core1, :].contiguous(),
                                s.core0, 2160)
    ref = whole[..., s.core0:s.core1, :]
    assert not torch.equal(naked[..., :64, :], ref[..., :64, :])          # cut edge: zero padding shows
    assert torch.equal(naked[..., 112:-112, :], ref[..., 112:-112, :])    # 112 rows in: exact again


def test_tiled_4k_four_bands_against_the_oracle(model, dev, seeded_sd):
    """BASELINE configs[4] without any process group: the 2160x3840 pair cut into the four bands of SURVEY 8d
    config 5 (origins 0/544/1088/1632) against the CPU ORACLE itself (not against the un-tiled HIP forward):
    fp32 within the 1e-3 contract and 1e-4 relative everywhere, with the rows either side of every cut
    checked on their own; bf16 within the bf16 contract; bf16x2 within the fp32 contract (1e-3), seams included.  The oracle's 4K forward takes ~20-40 s of host time."""
    f1, f2 = O.make_frames(9, 1, 2160, 3840)
    ref = O.unet_forward(seeded_sd, f1, f2)
    d1, d2 = f1.to(dev), f2.to(dev)
    scale = max(1.0, ref.abs().max().item())
    model.precision = "fp32"
    model.set_options()
    t32 = tiling.forward_tiled(model.forward_strip, d1, d2, 4).cpu()
    assert t32.shape == ref.shape
    d = (t32 - ref).abs()
    assert d.max().item() <= 1e-3 and d.max().item() <= 1e-4 * scale, d.max().item()
    for cut in (544, 1088, 1632):     # the seams: last rows of one band's core, first rows of the next one's
        assert d[..., cut - 8:cut + 8, :].max().item() <= 1e-4 * scale, (cut, d[..., cut - 8:cut + 8, :].max().item())
    model.precision = "bf16"
    model.set_options()
    t16 = tiling.forward_tiled(model.forward_strip, d1, d2, 4).cpu()
    rel = ((t16 - ref).norm() / ref.norm()).item()
    assert rel <= 2e-2, rel
    assert (t16 - ref).abs().max().item() <= 0.04 * (ref.max() - ref.min()).item()
    # strided sample of the kept rows: what a 4-GPU run would gather (every 7th row, every 5th column)
    sm = (t16[..., ::7, ::5] - ref[..., ::7, ::5]).norm() / ref[..., ::7, ::5].norm()
    assert sm.item() <= 2e-2, sm.item()
    # precision "bf16x2" (the fp32 contract on the bf16 pipe): the same four bands within the fp32 tolerance, seams included
    model.precision = "bf16x2"
    model.set_options()
    tx2 = tiling.forward_tiled(model.forward_strip, d1, d2, 4).cpu()
    dx = (tx2 - ref).abs()
    assert dx.max().item() <= 1e-3, dx.max().item()
    for cut in (544, 1088, 1632):
        assert dx[..., cut - 8:cut + 8, :].max().item() <= 1e-3, (cut, dx[..., cut - 8:cut + 8, :].max().item())
    model.precision = "fp32"


def test_strip_uses_global_upsample_coordinates(model, dev):
    """A band evaluated with its own (local) align_corners mapping differs from the whole image's;
    forward_strip with the true origin must not.  Guards against dropping the origin."""
    f1, f2 = O.make_frames(8, 1, 800, 64)
    f1, f2 = f1.to(dev), f2.to(dev)
    model.precision = "fp32"
    model.set_options()
    whole = model(f1, f2)
    lo, hi = 256, 800
    band_global = model.forward_strip(f1[..., lo:hi, :].contiguous(), f2[..., lo:hi, :].contiguous(), lo, 800)
    band_local = model(f1[..., lo:hi, :].contiguous(), f2[..., lo:hi, :].contiguous())
    core = slice(112, None)
    ref = whole[..., lo:hi, :][..., core, :]
    assert (band_global[..., core, :] - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    assert (band_local[..., core, :] - ref).abs().max().item() > 1e-4


def test_strip_argument_errors(model, dev):
    x = torch.zeros(1, 1, 64, 32, device=dev)
    model.precision = "fp32"
    with pytest.raises(RuntimeError, match="multiple of 16"):
        model.forward_strip(x, x, 8, 256)
    with pytest.raises(RuntimeError, match="multiple of 16"):
        model.forward_strip(x[..., :40, :].contiguous(), x[..., :40, :].contiguous(), 16, 256)
    with pytest.raises(RuntimeError):
        model.forward_strip(x, x, 224, 256)  # runs past the image
    out = model.forward_strip(x[..., :40, :].contiguous(), x[..., :40, :].contiguous(), 16, 56)  # ends the image
    assert out.shape == (1, 1, 40, 32)


def test_oversized_frame_is_refused_and_tiles_instead(model, dev):
    """H*W >= 2^26 pixels would overflow the kernels' 32-bit in-plane offsets: refused loudly."""
    model.precision = "bf16"
    x = torch.zeros(1, 1, 8192, 8192, device=dev)
    with pytest.raises(RuntimeError, match="row bands"):
        model(x, x)
