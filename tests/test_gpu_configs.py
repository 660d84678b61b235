"""GPU (MI355X): the BASELINE.json configs and kernel paths that round 1's GPU suite did not
exercise (VERDICT r01 "Next round" item 1), all through the C ABI:

  config 2   B=16 256x256 fp32 -- a different kernel path from the B=1 golden (the split-K
             decision depends on the number of workgroups, csrc/fiunet.hip launch_conv_maybe_split)
  config 4   the video loop at FULL frame size (1080p), short length, device and host variants
  RGB bf16   conv3x3_first_kernel<bf16,3> + EPI_HEAD3 in bf16, fused and unfused, odd size
  RGB fp32   pinned to the reference's own UNet(6,3) outputs (tests/golden/out_rgb_*.npz)
  PSNR       |PSNR_hip - PSNR_cpu| <= 0.05 dB against a TRUE middle frame, on a checkpoint that
             actually interpolates (oracle.make_interpolating_state_dict), at 256x256 and 1080p
Tolerances as in test_gpu_parity.py; the bf16 contract is restated where it is used.
"""
import os

import numpy as np
import pytest
import torch

import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import _native
from ai_based_frame_interpolation_amd import synthetic as S
from oracle import unet_oracle as O

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, seeded_sd):
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(seeded_sd)
    return m.to(dev).eval()


def test_config2_batch16_256_fp32(model, dev, seeded_sd, golden_dir):
    """BASELINE configs[1]: 16 pairs of 256x256, fp32 HIP path, vs the oracle on all 16 and vs the
    reference's own 256x256 output (golden) on the item that carries the golden pair."""
    g = np.load(os.path.join(golden_dir, "out_b1_256x256.npz"))
    f1, f2 = O.make_frames(1, 16, 256, 256)  # seed 1
    k = 11
    f1[k], f2[k] = torch.from_numpy(g["frame1"])[0], torch.from_numpy(g["frame2"])[0]
    ref = O.unet_forward(seeded_sd, f1, f2)
    model.precision = "fp32"
    model.set_options()
    out = model(f1.to(dev), f2.to(dev)).cpu()
    d = (out - ref).abs().max().item()
    assert d <= FP32_TOL and d <= 1e-4 * max(1.0, ref.abs().max().item()), d
    dg = np.abs(out[k].numpy() - g["out"][0]).max()
    assert dg <= FP32_TOL and dg <= 1e-4 * max(1.0, np.abs(g["out"]).max()), dg
    # item k alone takes the B=1 path, whose deep layers are K-split (another fp32 summation order)
    single = model(f1[k:k + 1].to(dev), f2[k:k + 1].to(dev)).cpu()
    assert (single[0] - out[k]).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    model.precision = "bf16"
    out16 = model(f1.to(dev), f2.to(dev)).cpu()
    assert ((out16 - ref).norm() / ref.norm()).item() <= 2e-2


def test_config4_video_loop_full_size_1080p(model, dev, seeded_sd):
    """BASELINE configs[3] at full frame size, short length: 17 synthetic 1080p uint8 frames through
    the device-resident and the host-resident loops (bit-equal to each other and to per-pair
    forward_u8); one pair against the oracle's post-processed frame."""
    n, h, w = 17, 1080, 1920
    frames = S.moving_frames(0, n, h, w, device="cpu", seed=7)
    assert frames.shape == (n, h, w) and frames.dtype == torch.uint8
    model.precision = "bf16"
    model.set_options()
    dseq = P.interpolate_sequence(model, frames.to(dev), batch=8)
    hseq = P.interpolate_sequence_host(model, frames, batch=8)
    assert dseq.shape == (2 * n - 1, h, w)
    assert torch.equal(dseq.cpu(), hseq)
    assert torch.equal(dseq[0::2].cpu(), frames)
    for i in (0, 7, 8, 15):  # first/last of a full chunk, first of the ragged last chunk, last pair
        mid = model.forward_u8(frames[i][None, None].to(dev), frames[i + 1][None, None].to(dev))[0, 0]
        assert torch.equal(dseq[2 * i + 1], mid), i
    # one pair vs the oracle (CPU, ~20-30 s): fp32 path within +-1 code on <= 1e-3 of the pixels,
    # bf16 path >= 45 dB from the oracle's frame
    i = 8
    fa, fb = O.preprocess_array(frames[i].numpy()), O.preprocess_array(frames[i + 1].numpy())
    want = O.postprocess_tensor(O.unet_forward(seeded_sd, fa, fb))
    model.precision = "fp32"
    got32 = model.forward_u8(frames[i][None, None].to(dev), frames[i + 1][None, None].to(dev))[0, 0].cpu().numpy()
    diff = np.abs(got32.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() <= 1e-3, (diff.max(), (diff != 0).mean())
    assert O.psnr_u8(want, dseq[2 * i + 1].cpu().numpy()) >= 45.0
    model.precision = "fp32"


def _rgb_model(dev, seed=77, precision="fp32"):
    sd = O.make_seeded_state_dict(seed, n_channels=6, n_classes=3)
    m = P.FrameInterpolationUNet(bilinear=True, frame_channels=3, precision=precision)
    m.load_state_dict(sd)
    return m.to(dev).eval(), sd


@pytest.mark.parametrize("name", ["rgb_b2_40x56", "rgb_b1_33x47"])
def test_rgb_fp32_matches_reference_unet_6_3(dev, golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"out_{name}.npz"))
    m, _ = _rgb_model(dev, int(g["weight_seed"]))
    f1, f2, ref = torch.from_numpy(g["frame1"]), torch.from_numpy(g["frame2"]), torch.from_numpy(g["out"])
    for unfused in (False, True):
        m.set_options(unfused=unfused)
        out = m(f1.to(dev), f2.to(dev)).cpu()
        d = (out - ref).abs().max().item()
        assert out.shape == ref.shape and d <= FP32_TOL and d <= 1e-4 * max(1.0, ref.abs().max().item()), d


def test_rgb_bf16_odd_size_fused_unfused_oracle(dev):
    """The RGB stem (round 4: stem_rgb_split_kernel, hi + lo split bf16 MFMA) and the 3-class fused head in bf16,
    at an odd size (floor-pool + asymmetric F.pad), fused vs unfused vs oracle; the stem's own output against the
    oracle's first activation to bf16 rounding (its arithmetic is ~2^-16 relative before that rounding)."""
    m, sd = _rgb_model(dev, 77, "bf16")
    f1, f2 = O.make_frames(33, 2, 45, 71, c=3)
    ref = O.unet_forward(sd, f1, f2)
    rng = (ref.max() - ref.min()).item()
    outs = {}
    for unfused in (False, True):
        m.set_options(unfused=unfused)
        acts, out = m.debug_activations(f1.to(dev), f2.to(dev))
        outs[unfused] = (acts, out.cpu())
        o = outs[unfused][1]
        assert o.shape == (2, 3, 45, 71) and torch.isfinite(o).all()
        assert ((o - ref).norm() / ref.norm()).item() <= 2e-2
        assert (o - ref).abs().max().item() <= 0.04 * rng
    m.set_options()
    # the stem alone: relu(bn(conv(cat(f1, f2)))) of the oracle vs tap 0 (dither off: it perturbs the INPUT by +-2^-9)
    m.set_options(no_dither=True)
    acts_nd, _ = m.debug_activations(f1.to(dev), f2.to(dev), taps=[0])
    m.set_options()
    taps = {}
    O.unet_forward(sd, f1, f2, taps=taps)
    k0 = "unet.inc.double_conv.0"
    got0, want0 = acts_nd[k0].cpu().float(), taps[k0]
    err = (got0 - want0).abs()
    assert got0.shape == want0.shape
    assert (err <= (2.0 ** -8 + 2.0 ** -13) * want0.abs() + 5e-4).all(), float(err.max())   # one bf16 rounding (<= 2^-8 relative) + the split arithmetic (2^-16 of the summed |terms|, which also decides the sign next to relu's kink)
    # RGB has no fused stem, so in bf16 all 18 stage outputs are bit-identical fused vs unfused
    for k in outs[False][0]:
        assert torch.equal(outs[False][0][k], outs[True][0][k]), k
    assert (outs[False][1] - outs[True][1]).abs().max().item() <= 2e-2
    # the uint8 path on RGB frames
    gen = torch.Generator().manual_seed(5)
    a = torch.randint(0, 256, (1, 3, 45, 71), dtype=torch.uint8, generator=gen)
    b = torch.randint(0, 256, (1, 3, 45, 71), dtype=torch.uint8, generator=gen)
    got = m.forward_u8(a.to(dev), b.to(dev)).cpu().numpy()
    fa = torch.stack([O.preprocess_array(a[0, c].numpy())[0, 0] for c in range(3)])[None]
    fb = torch.stack([O.preprocess_array(b[0, c].numpy())[0, 0] for c in range(3)])[None]
    want = np.stack([O.postprocess_tensor(O.unet_forward(sd, fa, fb)[:, c:c + 1]) for c in range(3)])[None]
    assert O.psnr_u8(want, got) >= 35.0


_PSNR_CASES = ([(256, 256, s, ck) for s in (3, 4, 5, 6, 7) for ck in (4321, 5321)] +
               [(540, 960, s, ck) for s in (3, 5, 7) for ck in (4321, 5321)] +
               [(1080, 1920, 3, 4321), (1080, 1920, 6, 5321)])


@pytest.mark.parametrize("h,w,scene,ckpt", _PSNR_CASES)
def test_psnr_within_0p05_db_of_cpu_reference(dev, h, w, scene, ckpt):
    """north_star: "PSNR within 0.05 dB of the CPU reference".  Frames t and t+2 of a synthetic
    scene in, frame t+1 is the truth; the checkpoint really interpolates (>= 27 dB), so a bf16
    error of a couple of uint8 codes WOULD move the PSNR by more than the bound.  Swept over scene
    seeds, three sizes and two checkpoint seeds (reference: inference.py:54-61 postprocess,
    evaluation.py:194-205 PSNR).  fp32: exactly the CPU reference's PSNR.  bf16: asserted at 0.03 dB -
    the measured worst case is ~0.015 dB now that the stem's input is dithered (DESIGN.md section 4;
    without the dither 0.02-0.05 dB, with round-to-nearest weights on top 0.04-0.07: both A/B options)."""
    sd = O.make_interpolating_state_dict(seed=ckpt)
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    a, truth, c = S.triplet(h, w, device="cpu", seed=scene)
    fa, fc = O.preprocess_array(a.numpy()), O.preprocess_array(c.numpy())
    torch.set_num_threads(16)
    ref_u8 = O.postprocess_tensor(O.unet_forward(sd, fa, fc))
    psnr_cpu = O.psnr_u8(truth.numpy(), ref_u8)
    assert psnr_cpu >= 27.0, psnr_cpu
    for prec, bound in (("fp32", 0.002), ("bf16", 0.03)):
        m.precision = prec
        hip_u8 = m.forward_u8(a[None, None].to(dev), c[None, None].to(dev))[0, 0].cpu().numpy()
        psnr_hip = O.psnr_u8(truth.numpy(), hip_u8)
        print(f"PSNR vs truth {h}x{w} scene {scene} ckpt {ckpt} {prec}: hip {psnr_hip:.4f} dB, cpu {psnr_cpu:.4f} dB, "
              f"delta {psnr_hip - psnr_cpu:+.4f}")
        assert abs(psnr_hip - psnr_cpu) <= bound, (prec, psnr_hip, psnr_cpu)
        assert O.psnr_u8(ref_u8, hip_u8) >= (60.0 if prec == "fp32" else 55.0)
    # sensitivity check: the criterion is not vacuous -- two uint8 codes of error break it
    noisy = np.clip(ref_u8.astype(int) + np.random.default_rng(0).integers(-3, 4, ref_u8.shape), 0, 255)
    assert abs(O.psnr_u8(truth.numpy(), noisy.astype(np.uint8)) - psnr_cpu) > 0.05


def test_psnr_options_dither_and_weight_rounding(dev):
    """The two A/B switches behind the PSNR figure: no_dither / rne_weights change the bf16 result (and
    only it), stay inside the looser historical bounds, and switching them back restores the default
    bits (the weight-rounding mode is applied at load time: the module re-uploads)."""
    sd = O.make_interpolating_state_dict()
    m = P.FrameInterpolationUNet(bilinear=True, precision="bf16")
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    a, truth, c = S.triplet(256, 256, device="cpu", seed=3)
    A, C = a[None, None].to(dev), c[None, None].to(dev)
    fa, fc = O.preprocess_array(a.numpy()), O.preprocess_array(c.numpy())
    psnr_cpu = O.psnr_u8(truth.numpy(), O.postprocess_tensor(O.unet_forward(sd, fa, fc)))
    base = m.forward_u8(A, C).clone()
    deltas = {}
    for name, kw in (("default", {}), ("no_dither", {"no_dither": True}), ("rne_weights", {"rne_weights": True}),
                     ("neither", {"no_dither": True, "rne_weights": True})):
        m.set_options(**kw)
        u8 = m.forward_u8(A, C)
        deltas[name] = O.psnr_u8(truth.numpy(), u8[0, 0].cpu().numpy()) - psnr_cpu
        if name == "default":
            assert torch.equal(u8, base)
        else:
            assert not torch.equal(u8, base)
    print("PSNR delta vs CPU reference, 256x256:", {k: round(v, 4) for k, v in deltas.items()})
    assert abs(deltas["default"]) <= 0.03 and all(abs(v) <= 0.12 for v in deltas.values())
    m.set_options()
    assert torch.equal(m.forward_u8(A, C), base)
    m.precision = "fp32"   # the fp32 path has neither rounding point
    f32 = m.forward_u8(A, C).clone()
    m.set_options(no_dither=True, rne_weights=True)
    assert torch.equal(m.forward_u8(A, C), f32)


@pytest.mark.parametrize("blend", [(0.7, 0.3), (1.0, 0.0)])
def test_input_dither_on_checkpoints_that_do_not_blend_symmetrically(dev, blend):
    """The ordered input dither (+d on frame 1, -d on frame 2, |d| <= 2^-9) cancels exactly in a network that
    outputs 0.5 (f1 + f2).  A real checkpoint need not blend symmetrically: here the analytic path is
    0.7 f1 + 0.3 f2 and the pure copy of frame 1 (the dither does not cancel at all), and the seeded RANDOM
    checkpoint has no carried path whatsoever.  With the dither ON (the default) the PSNR criterion must
    still hold at the north-star bound and be no worse than with it off (measured, round 4: 0.010 / 0.008 dB
    on, 0.045 / 0.034 dB off - the dither still breaks the intensity-correlated sawtooth of the carried
    values), and the raw bf16 error against the CPU reference, which now contains the uncancelled part of the
    dither itself (|b1 - b2| * 2^-9 / sqrt(3) rms = at most half the rms of the frames' own 8-bit
    quantisation; measured rel-L2 2.4e-3 / 4.7e-3 on, 2.0e-3 / 2.1e-3 off), stays 4x inside the bf16
    contract of 2e-2."""
    sd = O.make_interpolating_state_dict(seed=4321, blend=blend)
    m = P.FrameInterpolationUNet(bilinear=True, precision="bf16")
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    worst = {}
    for scene in (3, 5):
        a, truth, c = S.triplet(256, 256, device="cpu", seed=scene)
        A, C = a[None, None].to(dev), c[None, None].to(dev)
        ref = O.unet_forward(sd, O.preprocess_array(a.numpy()), O.preprocess_array(c.numpy()))
        ref_u8 = O.postprocess_tensor(ref)
        psnr_cpu = O.psnr_u8(truth.numpy(), ref_u8)
        for name, kw in (("dither", {}), ("no_dither", {"no_dither": True})):
            m.set_options(**kw)
            u8 = m.forward_u8(A, C)[0, 0].cpu().numpy()
            raw = m(O.preprocess_array(a.numpy()).to(dev), O.preprocess_array(c.numpy()).to(dev)).cpu()
            dp = abs(O.psnr_u8(truth.numpy(), u8) - psnr_cpu)
            rel = ((raw - ref).norm() / ref.norm()).item()
            w = worst.setdefault(name, [0.0, 0.0])
            w[0], w[1] = max(w[0], dp), max(w[1], rel)
    print(f"blend {blend}: worst |dPSNR| / rel-L2 vs CPU reference:", {k: (round(v[0], 4), round(v[1], 5)) for k, v in worst.items()})
    assert worst["dither"][0] <= 0.05 and worst["dither"][0] <= worst["no_dither"][0] + 0.005, worst
    assert worst["dither"][1] <= 5e-3 and worst["no_dither"][1] <= 5e-3, worst


def test_input_dither_on_the_random_checkpoint(model, dev, seeded_sd):
    """Seeded random checkpoint (no interpolating path at all): dither on vs off, raw bf16 error against the
    CPU reference - both inside the bf16 contract, on not worse than off by more than 5 %."""
    f1, f2 = O.make_frames(31, 1, 256, 256)
    ref = O.unet_forward(seeded_sd, f1, f2)
    model.precision = "bf16"
    rel = {}
    for name, kw in (("dither", {}), ("no_dither", {"no_dither": True})):
        model.set_options(**kw)
        out = model(f1.to(dev), f2.to(dev)).cpu()
        rel[name] = ((out - ref).norm() / ref.norm()).item()
    model.set_options()
    print("random checkpoint, bf16 rel-L2 vs CPU reference:", {k: round(v, 5) for k, v in rel.items()})
    assert rel["dither"] <= 2e-2 and rel["no_dither"] <= 2e-2
    assert rel["dither"] <= 1.05 * rel["no_dither"], rel


def test_bf16_error_contract_on_bench_network(dev):
    """bench.py's own random-init network (He-scaled convs, wide BatchNorm statistics) is harder on
    bf16 than the seeded test checkpoint (its output is a small residual, |out| <= 2.3, of activations
    of magnitude 5-8): the contract stated in DESIGN.md section 4 is rel-L2 <= 6e-2 and uint8
    PSNR(hip, cpu reference) >= 40 dB on uniform-random frames (measured 2.4e-2 / 43.4 dB at
    540x960 with the error-feedback weight rounding; 3.3e-2 / 40.7 dB with round-to-nearest)."""
    import bench
    model = bench.make_bench_model("bf16").to(dev).eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    f1, f2 = O.make_frames(2, 1, 270, 480)
    ref = O.unet_forward(sd, f1, f2)
    out = model(f1.to(dev), f2.to(dev)).cpu()
    assert ((out - ref).norm() / ref.norm()).item() <= 6e-2
    assert O.psnr_u8(O.postprocess_tensor(ref), O.postprocess_tensor(out)) >= 40.0
    model.precision = "fp32"
    out32 = model(f1.to(dev), f2.to(dev)).cpu()
    assert (out32 - ref).abs().max().item() <= FP32_TOL


@pytest.mark.parametrize("prec", ["bf16", "fp32"])
def test_tile_pair_kernel_is_bit_identical(model, dev, prec):
    """conv3x3_pair_kernel (8 waves on two pixel tiles, shared weight ring, double-buffered in-tile;
    FIUNET_OPT_PAIR_TILES) sums in the same order as conv3x3_mfma_kernel: every stage and the output
    are bit-identical.  B=8 270x480 puts levels 1-2 on the pair kernel; an odd tile count (B=3)
    exercises the idle second group of its last workgroup."""
    model.precision = prec
    for b, h, w in ((8, 270, 480), (3, 264, 480)):
        f1, f2 = O.make_frames(60 + b, b, h, w)
        f1, f2 = f1.to(dev), f2.to(dev)
        model.set_options()
        acts_a, a = model.debug_activations(f1, f2)
        model.set_options(pair_tiles=True)
        acts_b, bb = model.debug_activations(f1, f2)
        for k in acts_a:
            assert torch.equal(acts_a[k], acts_b[k]), (prec, b, k)
        assert torch.equal(a, bb)
    model.set_options()
    model.precision = "fp32"


@pytest.mark.gpu
def test_materialised_upsample_is_bit_identical_to_the_fused_gather(model, dev, seeded_sd):
    """bf16 concat convs with >= 2 cout tiles on >= 64k pixels (up2.0 here: 3 x 135 x 240 = 97k) read the
    upsampled half from a tensor written once by upsample_kernel (two-source direct conv) instead of
    interpolating it in every cout tile's gather (FIUNET_OPT_GATHER_UPSAMPLE forces the latter): every
    stage and the output are bit-identical, and the default really took the other kernel."""
    model.precision = "bf16"
    f1, f2 = O.make_frames(91, 3, 540, 960)
    f1, f2 = f1.to(dev), f2.to(dev)
    try:
        model.set_options()
        acts_a, a = model.debug_activations(f1, f2)
        model._ctx.profile_enable(True)
        model(f1, f2)
        torch.cuda.synchronize()
        _, rows = model._ctx.profile_read()
        model._ctx.profile_enable(False)
        names_default = [r[0] for r in rows]
        model.set_options(gather_upsample=True)
        acts_b, bb = model.debug_activations(f1, f2)
        model._ctx.profile_enable(True)
        model(f1, f2)
        torch.cuda.synchronize()
        _, rows = model._ctx.profile_read()
        model._ctx.profile_enable(False)
        names_gather = [r[0] for r in rows]
    finally:
        model.set_options()
        model.precision = "fp32"
    for k in acts_a:
        assert torch.equal(acts_a[k], acts_b[k]), k
    assert torch.equal(a, bb)
    # stage 12 = up2.conv.double_conv.0: template arguments <T, BN, TH, TW, MODE, EPI>, MODE 0 = direct, 2 = concat+upsample
    assert names_default[12].split(",")[4] == "0" and names_gather[12].split(",")[4] == "2", (names_default[12], names_gather[12])
    # one frame against the CPU oracle (bf16 contract of the seeded checkpoint)
    ref = O.unet_forward(seeded_sd, f1[:1].cpu(), f2[:1].cpu())
    rel = float((a[:1].cpu() - ref).norm() / ref.norm())
    assert rel <= 2e-2, rel


# ---- bilinear=False: the reference's DEFAULT constructor (ConvTranspose2d decoder, unet.py:42-44,66,99) ----------
_CONVT_GOLD = ["b1_32x48", "b2_17x31", "b1_135x240", "b1_70x86"]


@pytest.fixture(scope="module")
def convt_model(dev):
    m = P.FrameInterpolationUNet()          # default constructor = bilinear=False
    m.load_state_dict(O.make_seeded_state_dict(1234, bilinear=False))
    return m.to(dev).eval()


@pytest.mark.parametrize("name", _CONVT_GOLD)
def test_convtranspose_variant_matches_reference_golden(convt_model, dev, golden_dir, name):
    """fp32 within the 1e-3 contract (and 1e-4 relative) of the real class's output, bf16 within the bf16 contract;
    17x31 / 135x240 / 70x86 exercise F.pad after the transposed conv at one, several and all levels."""
    g = np.load(os.path.join(golden_dir, f"out_convt_{name}.npz"))
    f1, f2, ref = torch.from_numpy(g["frame1"]), torch.from_numpy(g["frame2"]), torch.from_numpy(g["out"])
    convt_model.precision = "fp32"
    convt_model.set_options()
    out = convt_model(f1.to(dev), f2.to(dev)).cpu()
    d = (out - ref).abs().max().item()
    assert out.shape == ref.shape and d <= FP32_TOL and d <= 1e-4 * max(1.0, ref.abs().max().item()), d
    convt_model.precision = "bf16"
    o16 = convt_model(f1.to(dev), f2.to(dev)).cpu()
    assert ((o16 - ref).norm() / ref.norm()).item() <= 2e-2
    assert (o16 - ref).abs().max().item() <= 0.04 * (ref.max() - ref.min()).item()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_convtranspose_variant_larger_frame_bands_and_u8(convt_model, dev, prec):
    """A frame large enough for the un-split kernels (and two row bands of it through forward_strip: the F.pad
    offsets and the low-res row mapping are evaluated in whole-image coordinates) against the oracle; the uint8
    path on top."""
    from ai_based_frame_interpolation_amd import tiling
    sd = O.make_seeded_state_dict(1234, bilinear=False)
    f1, f2 = O.make_frames(41, 2, 270, 200)
    ref = O.unet_forward(sd, f1, f2)
    convt_model.precision = prec
    convt_model.set_options()
    out = convt_model(f1.to(dev), f2.to(dev))
    if prec == "fp32":
        d = (out.cpu() - ref).abs().max().item()
        assert d <= FP32_TOL and d <= 1e-4 * max(1.0, ref.abs().max().item()), d
    else:
        assert ((out.cpu() - ref).norm() / ref.norm()).item() <= 2e-2
    tiled = tiling.forward_tiled(convt_model.forward_strip, f1.to(dev), f2.to(dev), 2)
    scale = max(1.0, ref.abs().max().item())
    assert (tiled - out).abs().max().item() <= (1e-5 if prec == "fp32" else 4e-2) * scale
    gen = torch.Generator().manual_seed(6)
    a = torch.randint(0, 256, (1, 1, 64, 80), dtype=torch.uint8, generator=gen)
    b = torch.randint(0, 256, (1, 1, 64, 80), dtype=torch.uint8, generator=gen)
    got = convt_model.forward_u8(a.to(dev), b.to(dev)).cpu().numpy()
    want = O.postprocess_tensor(O.unet_forward(sd, O.preprocess_array(a[0, 0].numpy()), O.preprocess_array(b[0, 0].numpy())))
    assert O.psnr_u8(want, got[0, 0]) >= (60.0 if prec == "fp32" else 35.0)


# ---- precision "bf16x2": the fp32 CONTRACT on the bf16 matrix cores (two-piece activations and weights) -------------
@pytest.mark.parametrize("name", ["b1_32x48", "b2_64x64", "b1_17x31", "b1_16x16", "b1_135x240", "b1_256x256"])
def test_bf16x2_meets_the_fp32_contract_on_the_reference_goldens(model, dev, golden_dir, name):
    """north_star: |d|_inf <= 1e-3 against the reference's PyTorch-CPU forward.  bf16x2 computes every product as
    wh*xh + wl*xh + wh*xl on the bf16 MFMAs (fp32 accumulation; activations stored with 16 significant bits), the
    stem and the head in exact fp32: measured max |d| 2e-5 .. 1.8e-4 on outputs of magnitude 1-3.5 (rel-L2 1.5e-5;
    the exact-fp32 path: 1e-5 / 1e-6).  Asserted at 1e-3 absolute AND 2e-4 relative to the output range."""
    g = np.load(os.path.join(golden_dir, f"out_{name}.npz"))
    f1, f2, ref = torch.from_numpy(g["frame1"]), torch.from_numpy(g["frame2"]), torch.from_numpy(g["out"])
    model.precision = "bf16x2"
    model.set_options()
    out = model(f1.to(dev), f2.to(dev)).cpu()
    model.precision = "fp32"
    d = (out - ref).abs().max().item()
    assert out.shape == ref.shape and out.dtype == torch.float32
    assert d <= FP32_TOL and d <= 2e-4 * max(1.0, ref.abs().max().item()), d
    assert ((out - ref).norm() / ref.norm()).item() <= 1e-4


def test_bf16x2_larger_frames_bands_u8_rgb_and_psnr(dev, seeded_sd):
    """A frame with 32-wide tiles, two row bands of it (whole-image upsample coordinates), the uint8 path, the RGB 6 -> 3 variant, and the PSNR criterion
    on an interpolating checkpoint (identical to the CPU reference's, as for the exact-fp32 path)."""
    from ai_based_frame_interpolation_amd import tiling
    m = P.FrameInterpolationUNet(bilinear=True, precision="bf16x2")
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    f1, f2 = O.make_frames(43, 2, 272, 208)
    ref = O.unet_forward(seeded_sd, f1, f2)
    out = m(f1.to(dev), f2.to(dev))
    d = (out.cpu() - ref).abs().max().item()
    assert d <= FP32_TOL and d <= 2e-4 * max(1.0, ref.abs().max().item()), d
    tiled = tiling.forward_tiled(m.forward_strip, f1.to(dev), f2.to(dev), 2)
    # small bands / small batches K-split their deep layers differently (a different fp32 summation order, then the
    # 16-bit re-rounding of the stored pieces): the same values within this precision's own accuracy class
    assert (tiled - out).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    assert (m(f1[:1].to(dev), f2[:1].to(dev)) - out[:1]).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    # uint8 in / out
    gen = torch.Generator().manual_seed(8)
    a = torch.randint(0, 256, (1, 1, 64, 80), dtype=torch.uint8, generator=gen)
    b = torch.randint(0, 256, (1, 1, 64, 80), dtype=torch.uint8, generator=gen)
    got = m.forward_u8(a.to(dev), b.to(dev)).cpu().numpy()
    want = O.postprocess_tensor(O.unet_forward(seeded_sd, O.preprocess_array(a[0, 0].numpy()), O.preprocess_array(b[0, 0].numpy())))
    assert O.psnr_u8(want, got[0, 0]) >= 60.0
    # uint8 in / out at a size where the stem is evaluated inside inc.3's gather (SRC_STEM_X2 reads the uint8 frames itself):
    # bit for bit preprocess -> forward -> postprocess, and the fused-stem forward against the unfused one (KEEP_ALL)
    a2 = torch.randint(0, 256, (2, 1, 256, 256), dtype=torch.uint8, generator=gen).to(dev)
    b2 = torch.randint(0, 256, (2, 1, 256, 256), dtype=torch.uint8, generator=gen).to(dev)
    pa, pb = _native.preprocess_u8(a2), _native.preprocess_u8(b2)
    fused = m(pa, pb)
    assert torch.equal(m.forward_u8(a2, b2), _native.postprocess_u8(fused))
    _, unfused = m.debug_activations(pa, pb)
    assert (fused - unfused).abs().max().item() <= 2e-4 * max(1.0, unfused.abs().max().item())
    ref2 = O.unet_forward(seeded_sd, pa.cpu(), pb.cpu())
    assert (fused.cpu() - ref2).abs().max().item() <= min(FP32_TOL, 2e-4 * max(1.0, ref2.abs().max().item()))
    # RGB 6 -> 3
    mr, sdr = _rgb_model(dev, 77, "bf16x2")
    r1, r2 = O.make_frames(34, 1, 45, 71, c=3)
    rr = O.unet_forward(sdr, r1, r2)
    dr = (mr(r1.to(dev), r2.to(dev)).cpu() - rr).abs().max().item()
    assert dr <= FP32_TOL and dr <= 2e-4 * max(1.0, rr.abs().max().item()), dr
    # PSNR criterion (north_star: within 0.05 dB of the CPU reference)
    sdi = O.make_interpolating_state_dict()
    mi = P.FrameInterpolationUNet(bilinear=True, precision="bf16x2")
    mi.load_state_dict(sdi)
    mi = mi.to(dev).eval()
    x, truth, z = S.triplet(256, 256, device="cpu", seed=3)
    ref_u8 = O.postprocess_tensor(O.unet_forward(sdi, O.preprocess_array(x.numpy()), O.preprocess_array(z.numpy())))
    hip_u8 = mi.forward_u8(x[None, None].to(dev), z[None, None].to(dev))[0, 0].cpu().numpy()
    assert abs(O.psnr_u8(truth.numpy(), hip_u8) - O.psnr_u8(truth.numpy(), ref_u8)) <= 0.002
    # the per-layer read-back at a size with un-split kernels and several tiles (taps + the four materialised
    # upsampled halves, which the oracle records before F.pad)
    taps = {}
    O.unet_forward(seeded_sd, f1[:1], f2[:1], taps)
    acts, _ = m.debug_activations(f1[:1].to(dev), f2[:1].to(dev), with_up=True)
    assert len(acts) == 22
    for name, a in acts.items():
        r = taps[name]
        if name.endswith(".up"):
            dy, dx = a.shape[2] - r.shape[2], a.shape[3] - r.shape[3]
            r = torch.nn.functional.pad(r, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        assert a.shape == r.shape, name
        assert (a.cpu() - r).abs().max().item() <= 2e-4 * r.abs().max().item(), name


@pytest.mark.parametrize("name", _CONVT_GOLD)
def test_bf16x2_convtranspose_variant_meets_the_fp32_contract(convt_model, dev, golden_dir, name):
    """bilinear=False (the reference's default constructor, unet.py:42-44,66,99) in precision bf16x2: the transposed
    convs run on two-piece operands too (convt2x2_kernel<bf16, X2>); against the real class's outputs."""
    g = np.load(os.path.join(golden_dir, f"out_convt_{name}.npz"))
    f1, f2, ref = torch.from_numpy(g["frame1"]), torch.from_numpy(g["frame2"]), torch.from_numpy(g["out"])
    convt_model.precision = "bf16x2"
    convt_model.set_options()
    try:
        out = convt_model(f1.to(dev), f2.to(dev)).cpu()
    finally:
        convt_model.precision = "fp32"
    d = (out - ref).abs().max().item()
    assert out.shape == ref.shape and d <= FP32_TOL and d <= 2e-4 * max(1.0, ref.abs().max().item()), d


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x2"])
def test_convtranspose_variant_per_layer_golden_fixture(convt_model, dev, golden_dir, prec):
    """`FrameInterpolationUNet()` (bilinear=False) stage by stage against activations recorded from the reference's
    own default-constructed class (oracle/gen_golden.py gen_convt_layers: hooks on its ReLUs and on its four
    ConvTranspose2d modules): 18 conv+BN+ReLU outputs + the four `up{k}.up` outputs (ours are stored already
    padded: the fixture's window is cut out of them, and the padding must be exactly zero) + the head.
    34x52: F.pad is live after up2 (a column) and up3 (a row).  This is also the regression test of the read-back's
    buffer sizing for this wider decoder (taps 8, 9, 11, 13, 15 have twice the bilinear variant's channels)."""
    g = np.load(os.path.join(golden_dir, "layers_convt_b1_34x52.npz"))
    convt_model.precision = prec
    convt_model.set_options()
    try:
        acts, out = convt_model.debug_activations(torch.from_numpy(g["frame1"]).to(dev),
                                                  torch.from_numpy(g["frame2"]).to(dev), with_up=True)
    finally:
        convt_model.precision = "fp32"
    assert len(acts) == 22
    rel_tol = {"fp32": 1e-5, "bf16x2": 2e-4, "bf16": 2.5e-2}[prec]
    for name, a in acts.items():
        shape = tuple(g[f"{name}|shape"])
        a = a.cpu()
        if name.endswith(".up"):   # ours: after F.pad (unet.py:49-53: left/top get diff // 2)
            dy, dx = a.shape[2] - shape[2], a.shape[3] - shape[3]
            assert 0 <= dy <= 1 and 0 <= dx <= 1, (name, a.shape, shape)
            inner = a[:, :, dy // 2:dy // 2 + shape[2], dx // 2:dx // 2 + shape[3]]
            outside = torch.ones_like(a, dtype=torch.bool)
            outside[:, :, dy // 2:dy // 2 + shape[2], dx // 2:dx // 2 + shape[3]] = False
            assert not a[outside].any(), name   # the padding is exactly zero
            a = inner.contiguous()
        assert tuple(a.shape) == shape, (name, a.shape, shape)
        got = a.reshape(-1)[torch.from_numpy(g[f"{name}|idx"])].numpy()
        want = g[f"{name}|val"]
        assert np.abs(got - want).max() <= rel_tol * max(1.0, np.abs(want).max()), (name, np.abs(got - want).max())
    got = out.cpu().reshape(-1)[torch.from_numpy(g["unet.outc|idx"])].numpy()
    want = g["unet.outc|val"]
    tol = FP32_TOL if prec != "bf16" else 0.04 * (want.max() - want.min())
    assert np.abs(got - want).max() <= tol


# ---- round 6: the small-problem configuration (tile family, K cut) -----------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x2"])
def test_tile_family_never_changes_a_bit(model, dev, prec):
    """The small tile (64 couts x 8x32 pixels, 64 x 64 wave tiles, three workgroups per CU, its own fused-stem and fused-head
    forms) against the tuned tiles on the same K cut: bit-identical on every stage - the summation order of an output
    element is (plane, kx, ky) in every tile shape and the fused head reduces in the same association.  Sizes: one that
    the default configuration runs on small tiles (64x96) and one it runs on tuned tiles (2 x 270x480), each forced
    both ways with the K loop whole."""
    model.precision = prec
    model.set_options()
    for b, h, w in ((1, 64, 96), (2, 270, 480)):
        f1, f2 = O.make_frames(41, b, h, w)
        f1, f2 = f1.to(dev), f2.to(dev)
        model(f1, f2)   # (creates the context)
        outs = []
        for tile in (1, 2):
            for layer in range(1, 18):
                model._ctx.force_cfg(layer, tile, 1)
            outs.append(model(f1, f2).clone())
        model._ctx.force_cfg(-1)
        assert torch.equal(outs[0], outs[1]), (prec, b, h, w, (outs[0] - outs[1]).abs().max().item())
    model.precision = "fp32"


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x2"])
def test_k_cut_is_deterministic_and_within_contract(model, dev, seeded_sd, prec):
    """Every K cut (2 .. 16 slices, both tile families, the tile finalize with its fused pool) against the oracle: the
    cut changes only where the fp32 partial sums meet, so fp32 / bf16x2 stay inside the 1e-3 contract and bf16 inside
    its own; and a cut forward is deterministic."""
    f1, f2 = O.make_frames(43, 2, 48, 80)
    ref = O.unet_forward(seeded_sd, f1, f2)
    d1, d2 = f1.to(dev), f2.to(dev)
    model.precision = prec
    model.set_options()
    model(d1, d2)
    for tile in (1, 2):
        for k in (2, 4, 16):
            for layer in range(1, 18):
                model._ctx.force_cfg(layer, tile, k)
            a = model(d1, d2)
            assert torch.equal(a, model(d1, d2))
            err = (a.cpu() - ref).abs().max().item()
            if prec == "bf16":
                assert ((a.cpu() - ref).norm() / ref.norm()).item() <= 2e-2
            else:
                assert err <= 1e-3, (prec, tile, k, err)
    model._ctx.force_cfg(-1)
    model.precision = "fp32"


@pytest.mark.parametrize("prec", ["fp32", "bf16", "bf16x2"])
def test_in_workgroup_k_cut_kernel(model, dev, seeded_sd, prec):
    """conv3x3_kwave_kernel (direct sources, the K loop cut over the four waves of a workgroup, reduced through LDS;
    16-channel planes in fp32, 32-channel in bf16, in bf16x2 three virtual planes per real plane and a two-piece
    epilogue): forced onto every conv it
    covers - plain and pooled epilogues, two-source direct convs (the materialised upsampled half), odd sizes with
    partial tiles - it must be deterministic, position-invariant, and within the precision's contract of the oracle,
    close to the default configuration's result (same products, another fp32 association)."""
    model.precision = prec
    model.set_options()
    for b, h, w in ((1, 64, 96), (3, 33, 47), (2, 135, 240)):
        f1, f2 = O.make_frames(47, b, h, w)
        ref = O.unet_forward(seeded_sd, f1, f2)
        d1, d2 = f1.to(dev), f2.to(dev)
        base = model(d1, d2).clone()
        for layer in range(1, 18):
            model._ctx.force_cfg(layer, 3, 0)
        model._ctx.profile_enable(True)
        out = model(d1, d2).clone()
        _, rows = model._ctx.profile_read()
        model._ctx.profile_enable(False)
        # the direct convs with >= 4 planes
        assert sum("kwave" in r[0] for r in rows) >= 8, [r[0] for r in rows]
        assert torch.equal(out, model(d1, d2))                                       # deterministic
        perm = torch.roll(torch.arange(b), 1).to(dev)
        rolled = model(d1[perm].contiguous(), d2[perm].contiguous())                # same batch size: the other layers' K cuts stay put
        model._ctx.force_cfg(-1)
        assert torch.equal(rolled, out[perm])                                        # position invariant
        if prec == "bf16":
            rel = ((out.cpu() - ref).norm() / ref.norm()).item()
            assert rel <= 2e-2, (b, h, w, rel)
            assert ((out - base).norm() / base.norm()).item() <= 1e-2
        else:
            assert (out.cpu() - ref).abs().max().item() <= 1e-3, (b, h, w, (out.cpu() - ref).abs().max().item())
    model.precision = "fp32"
