"""PSNR / SSIM.  CPU: the oracle's restatement of skimage's algorithm against an
independent window-by-window evaluation of the definition.  GPU: the device kernels against the
oracle (fp64; 1e-9 absolute on SSIM, 1e-9 dB on PSNR), bit-deterministic across runs."""
import math

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as M


def _pair(seed, h, w, noise=20):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-noise, noise + 1, (h, w)), 0, 255).astype(np.uint8)
    return a, b


@pytest.mark.parametrize("h,w", [(7, 7), (8, 11), (16, 16), (21, 13)])
def test_oracle_ssim_matches_definition(h, w):
    a, b = _pair(h * 100 + w, h, w)
    assert abs(M.ssim_u8(b, a) - M.ssim_bruteforce(b, a)) < 1e-12
    assert M.ssim_u8(a, a) == pytest.approx(1.0, abs=1e-15)
    # SSIM is symmetric in its arguments; PSNR too
    assert M.ssim_u8(a, b) == pytest.approx(M.ssim_u8(b, a), abs=1e-14)
    assert M.psnr_u8(a, b) == M.psnr_u8(b, a)


def test_oracle_psnr_known_values():
    a = np.zeros((4, 4), np.uint8)
    b = np.full((4, 4), 255, np.uint8)
    assert M.psnr_u8(a, b) == 0.0                       # mse = 255^2
    c = a.copy(); c[0, 0] = 16                          # mse = 256/16 = 16
    assert M.psnr_u8(a, c) == pytest.approx(10 * math.log10(255.0 ** 2 / 16.0), abs=1e-12)
    assert M.psnr_u8(a, a) == float("inf")
    with pytest.raises(ValueError):
        M.ssim_u8(np.zeros((6, 9), np.uint8), np.zeros((6, 9), np.uint8))


# ---- GPU -------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 7, 7), (3, 16, 16), (2, 23, 70), (1, 135, 240), (2, 64, 129), (1, 1080, 1920)])
def test_device_metrics_match_oracle(shape):
    from ai_based_frame_interpolation_amd import metrics
    dev = torch.device("cuda:0")
    n, h, w = shape
    pairs = [_pair(7 + i, h, w, noise=5 + 30 * i) for i in range(n)]
    pred = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    targ = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    ps = metrics.psnr_u8(pred, targ).cpu().numpy()
    ss = metrics.ssim_u8(pred, targ).cpu().numpy()
    assert ps.shape == (n,) and ss.shape == (n,) and ps.dtype == np.float64
    for i, (a, b) in enumerate(pairs):
        assert ps[i] == pytest.approx(M.psnr_u8(b, a), abs=1e-9)
        assert ss[i] == pytest.approx(M.ssim_u8(b, a), abs=1e-9)
    # deterministic: fixed-order fp64 reduction, integer atomics
    assert np.array_equal(metrics.ssim_u8(pred, targ).cpu().numpy(), ss)
    assert np.array_equal(metrics.psnr_u8(pred, targ).cpu().numpy(), ps)


@pytest.mark.gpu
def test_device_metrics_edge_cases():
    from ai_based_frame_interpolation_amd import metrics
    dev = torch.device("cuda:0")
    a = torch.randint(0, 256, (2, 3, 40, 50), dtype=torch.uint8, device=dev)
    assert torch.isinf(metrics.psnr_u8(a, a)).all()
    assert torch.allclose(metrics.ssim_u8(a, a), torch.ones(2, 3, dtype=torch.float64, device=dev), atol=1e-15)
    assert metrics.psnr_u8(a, a).shape == (2, 3)          # one value per [H, W] plane
    # unaligned plane bases (odd H*W) take the byte path
    b = torch.randint(0, 256, (3, 9, 11), dtype=torch.uint8, device=dev)
    c = torch.randint(0, 256, (3, 9, 11), dtype=torch.uint8, device=dev)
    got = metrics.psnr_u8(b, c).cpu().numpy()
    for i in range(3):
        assert got[i] == pytest.approx(M.psnr_u8(b[i].cpu().numpy(), c[i].cpu().numpy()), abs=1e-9)
    with pytest.raises(RuntimeError, match="7x7"):
        metrics.ssim_u8(a[..., :6, :], a[..., :6, :])
    with pytest.raises(RuntimeError, match="uint8"):
        metrics.psnr_u8(a.float(), a.float())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        metrics.psnr_u8(a.cpu(), a.cpu())


@pytest.mark.gpu
def test_metrics_on_the_forward_u8_output(seeded_sd):
    """The evaluation loop's shape: uint8 frames -> forward_u8 -> PSNR/SSIM vs a ground-truth frame,
    all on the device; equals the host-side scoring of the same uint8 output."""
    import ai_based_frame_interpolation_amd as P
    dev = torch.device("cuda:0")
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(2)
    f0 = torch.randint(0, 256, (2, 1, 96, 160), dtype=torch.uint8, generator=g)
    f2 = torch.randint(0, 256, (2, 1, 96, 160), dtype=torch.uint8, generator=g)
    gt = ((f0.int() + f2.int()) // 2).to(torch.uint8)
    out = m.forward_u8(f0.to(dev), f2.to(dev))
    ps = P.metrics.psnr_u8(out, gt.to(dev)).cpu()
    ss = P.metrics.ssim_u8(out, gt.to(dev)).cpu()
    o = out.cpu().numpy()
    for b in range(2):
        assert ps[b, 0].item() == pytest.approx(M.psnr_u8(o[b, 0], gt[b, 0].numpy()), abs=1e-9)
        assert ss[b, 0].item() == pytest.approx(M.ssim_u8(o[b, 0], gt[b, 0].numpy()), abs=1e-9)


@pytest.mark.gpu
def test_evaluate_triplets_matches_host_scoring(seeded_sd):
    """evaluation_simple.py:134-244 on device == the same loop scored on the host with the oracle."""
    import ai_based_frame_interpolation_amd as P
    from ai_based_frame_interpolation_amd import evaluation
    from oracle import unet_oracle as O
    dev = torch.device("cuda:0")
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(seeded_sd)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(4)
    n, h, w = 5, 48, 80
    base = torch.randint(0, 256, (n + 2, 1, h, w), dtype=torch.uint8, generator=g)
    f0, gt, f1 = base[:-2], base[1:-1], base[2:]
    res = evaluation.evaluate_triplets(m, f0.to(dev), f1.to(dev), gt.to(dev), batch=2)
    assert res["total_triplets"] == n and res["methods"] == ["unet", "linear"]
    # same chunks as the evaluator: at this tiny size the K-split of the deep layers (hence the fp32
    # summation order, hence a pixel on a truncation boundary) depends on the batch size
    unet_u8 = torch.cat([m.forward_u8(f0[s:s + 2].to(dev), f1[s:s + 2].to(dev)) for s in range(0, n, 2)]).cpu().numpy()
    lin = O.postprocess_tensor((O.preprocess_array(f0.numpy()) + O.preprocess_array(f1.numpy())) / 2.0)
    for name, pred in (("unet", unet_u8), ("linear", lin.reshape(n, 1, h, w))):
        ps = np.array([M.psnr_u8(pred[i, 0], gt[i, 0].numpy()) for i in range(n)])
        ss = np.array([M.ssim_u8(pred[i, 0], gt[i, 0].numpy()) for i in range(n)])
        got = res["per_triplet"][name]
        assert np.allclose(got["psnr"], ps, atol=1e-9) and np.allclose(got["ssim"], ss, atol=1e-9)
        st = res["metrics_by_method"][name]
        assert st["average_psnr"] == pytest.approx(np.mean(ps), abs=1e-9)
        assert st["std_ssim"] == pytest.approx(np.std(ss), abs=1e-9)
        assert st["min_psnr"] == pytest.approx(ps.min(), abs=1e-9) and st["max_ssim"] == pytest.approx(ss.max(), abs=1e-9)
    # the reference's third method: OpenCV when importable, else the restatement of optical_flow.py (parity
    # unpinned against OpenCV; tests/test_host.py pins its behaviour) - scored on the device like the others
    r3 = evaluation.evaluate_triplets(m, f0.to(dev), f1.to(dev), gt.to(dev), methods=evaluation.ALL_METHODS, batch=2)
    assert r3["methods"] == ["unet", "linear", "optical_flow"] and "optical_flow_backend" in r3
    of = r3["per_triplet"]["optical_flow"]
    assert of["psnr"].shape == (n,) and np.isfinite(of["psnr"]).all() and np.isfinite(of["ssim"]).all()
    assert np.allclose(r3["per_triplet"]["unet"]["psnr"], res["per_triplet"]["unet"]["psnr"])


# ---- Gaussian-window SSIM / CombinedLoss of the training loss (train.py:18-87) --------------------
# Fixtures: oracle/gen_golden.py --ssim-only ran the reference's own SSIMLoss / CombinedLoss.
GAUSS = ["b1c1_32x48", "b2c1_64x64", "b1c3_33x47", "b3c3_17x31", "b1c1_7x9", "b1c1_256x256", "b2c1_135x240"]


def _gauss_fixture(golden_dir, name):
    import os
    g = np.load(os.path.join(golden_dir, f"ssim_gauss_{name}.npz"))
    return (torch.from_numpy(g["img1"]), torch.from_numpy(g["img2"]), float(g["ssim_loss"]),
            g["ssim_loss_per_sample"], float(g["combined_loss"]), float(g["mse"]))


@pytest.mark.parametrize("name", GAUSS)
def test_oracle_gauss_ssim_equals_reference(golden_dir, name):
    """The restatement against the values the reference's own classes produced (fp32: same aten ops, so
    only thread-count-dependent summation order can differ)."""
    torch.set_num_threads(8)
    a, b, loss, per, comb, mse = _gauss_fixture(golden_dir, name)
    assert float(1 - M.ssim_gauss(a, b)) == pytest.approx(loss, abs=2e-6)
    assert np.allclose((1 - M.ssim_gauss(a, b, size_average=False)).numpy(), per, atol=2e-6)
    assert float(M.combined_loss(a, b)) == pytest.approx(comb, abs=2e-6)
    # evaluated in double the formula sits within fp32 rounding of the reference's value
    assert float(1 - M.ssim_gauss(a, b, dtype=torch.float64)) == pytest.approx(loss, abs=5e-6)
    # identical images: ssim == 1 up to rounding; the window is normalised
    assert float(M.ssim_gauss(a, a)) == pytest.approx(1.0, abs=1e-6)
    assert float(M.gauss_window().sum()) == pytest.approx(1.0, abs=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GAUSS)
def test_device_gauss_ssim_matches_reference_fixtures(golden_dir, name):
    """The HIP kernel against the reference's own SSIMLoss / CombinedLoss values: <= 1e-5."""
    from ai_based_frame_interpolation_amd import metrics
    dev = torch.device("cuda:0")
    a, b, loss, per, comb, mse = _gauss_fixture(golden_dir, name)
    A, B = a.to(dev), b.to(dev)
    got = metrics.SSIMLoss()(A, B)
    assert got.dtype == torch.float32 and got.dim() == 0
    assert float(got) == pytest.approx(loss, abs=1e-5)
    got_per = metrics.SSIMLoss(size_average=False)(A, B).cpu().numpy()
    assert got_per.shape == per.shape and np.allclose(got_per, per, atol=1e-5)
    assert float(metrics.CombinedLoss()(A, B)) == pytest.approx(comb, abs=1e-5)
    m, s = metrics.CombinedLoss().terms(A, B)
    assert float(m) == pytest.approx(mse, rel=1e-5, abs=1e-8)
    # fp64 device value vs the oracle's fp64 evaluation of the same formula: 1e-9
    v64 = float(metrics.ssim_gauss(A, B, dtype=torch.float64))
    assert v64 == pytest.approx(float(M.ssim_gauss(a, b, dtype=torch.float64)), abs=1e-9)
    assert float(s) == v64
    # deterministic, symmetric, 1 on identical inputs
    assert float(metrics.ssim_gauss(A, B, dtype=torch.float64)) == v64
    assert float(metrics.ssim_gauss(B, A, dtype=torch.float64)) == pytest.approx(v64, abs=1e-12)
    assert float(metrics.ssim_gauss(A, A, dtype=torch.float64)) == pytest.approx(1.0, abs=1e-12)


@pytest.mark.gpu
def test_device_gauss_ssim_full_size_and_errors():
    from ai_based_frame_interpolation_amd import metrics
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    a = torch.rand(2, 1, 1080, 1920, generator=g)
    b = (a + 0.1 * (torch.rand(2, 1, 1080, 1920, generator=g) - 0.5)).clamp(0, 1)
    torch.set_num_threads(16)
    ref = M.ssim_gauss(a, b, size_average=False, dtype=torch.float64).numpy()
    got = metrics.ssim_gauss(a.to(dev), b.to(dev), size_average=False, dtype=torch.float64).cpu().numpy()
    assert np.allclose(got, ref, atol=1e-9)
    # other odd windows run the generic-radius kernel
    for ws in (3, 7, 15):
        r = float(M.ssim_gauss(a[:, :, :70, :90], b[:, :, :70, :90], window_size=ws, dtype=torch.float64))
        d = float(metrics.ssim_gauss(a[:, :, :70, :90].to(dev), b[:, :, :70, :90].to(dev), window_size=ws,
                                     dtype=torch.float64))
        assert d == pytest.approx(r, abs=1e-9)
    with pytest.raises(RuntimeError, match="odd"):
        metrics.ssim_gauss(a.to(dev), b.to(dev), window_size=10)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        metrics.ssim_gauss(a, b)
    with pytest.raises(RuntimeError, match="fp32"):
        metrics.ssim_gauss(a.to(dev).double(), b.to(dev).double())
