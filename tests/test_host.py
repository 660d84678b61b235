"""CPU: host-side mirror of the reference interface (no device work)."""
import os

import numpy as np
import pytest
import torch

import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import video
from oracle import unet_oracle as O


def test_module_schema_equals_reference(golden_dir):
    ref = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "state_dict_schema.txt"))]
    m = P.FrameInterpolationUNet(bilinear=True)
    sd = m.state_dict()
    assert list(sd.keys()) == [r[0] for r in ref]
    for k, shp, dt in ref:
        assert tuple(sd[k].shape) == tuple(int(x) for x in shp.split(",") if x)
        assert str(sd[k].dtype) == dt
    assert P.count_parameters(m) == 17262401  # SURVEY section 0 (bilinear=True)
    assert m.unet.n_channels == 2 and m.unet.n_classes == 1 and m.unet.bilinear is True


def test_rgb_variant_schema():
    m = P.FrameInterpolationUNet(bilinear=True, frame_channels=3)
    sd = m.state_dict()
    assert tuple(sd["unet.inc.double_conv.0.weight"].shape) == (64, 6, 3, 3)
    assert tuple(sd["unet.outc.conv.weight"].shape) == (3, 64, 1, 1)


def test_constructor_default_is_the_convtranspose_variant(golden_dir):
    """The reference's default is bilinear=False (unet.py:99 -> :66): ConvTranspose2d decoder, 31 037 057 parameters
    (QUICK_START.md's "31,031,809" matches neither variant).  Same 118-tensor state-dict - names,
    shapes, order - as the real class (fixture dumped from it by oracle/gen_golden.py --convt-only), strict load of
    the seeded checkpoint, and the attribute tree the reference exposes (`up1.up` is the transposed conv)."""
    m = P.FrameInterpolationUNet()
    assert m.unet.bilinear is False and P.count_parameters(m) == 31037057
    want = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "state_dict_schema_convt.txt"))]
    got = [(k, ",".join(map(str, v.shape)), str(v.dtype)) for k, v in m.state_dict().items()]
    assert got == [tuple(w) for w in want]
    res = m.load_state_dict(O.make_seeded_state_dict(1234, bilinear=False), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert tuple(m.unet.up1.up.weight.shape) == (1024, 512, 2, 2) and tuple(m.unet.up4.up.bias.shape) == (64,)
    assert tuple(m.unet.down4.maxpool_conv[1].double_conv[3].weight.shape) == (1024, 1024, 3, 3)
    with pytest.raises(Exception):   # a bilinear=True checkpoint does not fit (different widths, no `up` tensors)
        m.load_state_dict(O.make_seeded_state_dict(1234), strict=True)


def test_load_state_dict_strict_roundtrip(seeded_sd):
    m = P.FrameInterpolationUNet(bilinear=True)
    res = m.load_state_dict(seeded_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in m.state_dict().items():
        assert torch.equal(v, seeded_sd[k])


def test_checkpoint_formats(tmp_path, seeded_sd, capsys):
    """train.py:234-243 wrapped dict and bare state-dict both load (inference.py:83-94)."""
    wrapped = tmp_path / "best_model.pth"
    torch.save({"epoch": 3, "model_state_dict": seeded_sd, "optimizer_state_dict": {},
                "train_loss": 0.1, "val_loss": 0.25, "train_losses": [0.1], "val_losses": [0.25]},
               wrapped)
    bare = tmp_path / "bare.pth"
    torch.save(seeded_sd, bare)
    for p in (wrapped, bare):
        m = P.load_model(str(p), "cpu")
        assert not m.training
        assert torch.equal(m.state_dict()["unet.outc.conv.bias"], seeded_sd["unet.outc.conv.bias"])
    assert "Best validation loss: 0.250000" in capsys.readouterr().out
    with pytest.raises(FileNotFoundError):
        P.load_model(str(tmp_path / "nope.pth"), "cpu")


def test_no_cpu_fallback(seeded_sd):
    m = P.FrameInterpolationUNet(bilinear=True).eval()
    x = torch.zeros(1, 1, 32, 32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.postprocess_image(x)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 32, 32), torch.zeros(1, 1, 32, 31))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))


def test_product_code_never_imports_oracle():
    import re
    pkg = os.path.dirname(P.__file__)
    pat = re.compile(r"^\s*(from|import)\s+oracle|unet_oracle|c_oracle|libunet_oracle", re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                assert not pat.search(open(os.path.join(root, f)).read()), f


def test_preprocess_matches_reference_arithmetic(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(256, 256), dtype=np.uint8)
    t = P.preprocess_image(img)
    assert t.shape == (1, 1, 256, 256) and t.dtype == torch.float32
    assert torch.equal(t, O.preprocess_array(img))
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    np.save(tmp_path / "a.npy", img)
    assert torch.equal(P.preprocess_image(str(tmp_path / "a.npy")), t)
    with open(tmp_path / "a.pgm", "wb") as f:
        f.write(b"P5\n# c\n256 256\n255\n" + img.tobytes())
    assert torch.equal(P.preprocess_image(str(tmp_path / "a.pgm")), t)
    small = P.preprocess_image(img, target_size=(64, 32))
    assert small.shape == (1, 1, 32, 64)
    with pytest.raises(ValueError, match="Could not read image"):
        P.preprocess_image(str(tmp_path / "missing.png"))


def test_partition_pairs():
    parts = video.partition_pairs(3000, 8)
    assert parts[0] == (0, 375) and parts[-1] == (2625, 374)
    assert sum(c for _, c in parts) == 2999
    assert video.partition_pairs(2, 4) == [(0, 1), (1, 0), (1, 0), (1, 0)]
    assert video.partition_pairs(1, 2) == [(0, 0), (0, 0)]
    for n in (2, 3, 7, 16, 17, 100):
        for w in (1, 2, 3, 8):
            parts = video.partition_pairs(n, w)
            covered = [i for s, c in parts for i in range(s, s + c)]
            assert covered == list(range(n - 1))


def test_module_tree_matches_reference(golden_dir):
    """named_modules() names, child counts and nn.Sequential-style indexing equal the reference's
    (tests/golden/named_modules.txt, dumped from /root/reference/model/unet.py by gen_golden.py)."""
    ref = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "named_modules.txt"))]
    m = P.FrameInterpolationUNet(bilinear=True)
    mine = {n: mod for n, mod in m.named_modules()}
    assert [n for n, _ in m.named_modules()] == [r[0] for r in ref]
    for name, cls, nchild in ref:
        assert len(list(mine[name].children())) == int(nchild), name
        if cls in ("Conv2d", "BatchNorm2d"):
            assert type(mine[name]).__name__ == cls, name
    dc = m.unet.inc.double_conv
    assert len(dc) == 6 and dc[0] is dc["0"] and dc[-2] is dc["4"] and len(dc[:2]) == 2
    assert isinstance(dc[0], torch.nn.Conv2d) and isinstance(dc[4], torch.nn.BatchNorm2d)
    assert m.unet.down1.maxpool_conv[1].double_conv[3].weight.shape == (128, 128, 3, 3)
    assert m.unet.up1.conv.double_conv[0].weight.shape == (512, 1024, 3, 3)
    assert hasattr(m.unet.up1, "up") and [type(x).__name__ for x in dc][0] == "Conv2d"
    with pytest.raises(IndexError):
        dc[6]


def test_module_copies_and_pickles_without_the_hip_context(seeded_sd):
    import copy, io
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(seeded_sd)
    m._ctx = object()  # stands in for a live ctypes handle (not picklable in the real case either)
    m2 = copy.deepcopy(m)
    assert m2._ctx is None and m2._ctx_dirty
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m3 = torch.load(buf, weights_only=False)
    assert m3._ctx is None and torch.equal(m3.state_dict()["unet.outc.conv.bias"], seeded_sd["unet.outc.conv.bias"])


def test_image_file_io_without_opencv(tmp_path):
    """PNG / BMP readers, the PNG writer and cv2.resize's fixed-point INTER_LINEAR (restated; unpinned
    against cv2 itself, pinned to their definitions here)."""
    import struct, zlib
    from ai_based_frame_interpolation_amd import imageio_lite as IO
    rng = np.random.default_rng(5)
    gray = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    rgb = rng.integers(0, 256, (20, 31, 3), dtype=np.uint8)
    IO.write_png(str(tmp_path / "g.png"), gray)
    IO.write_png(str(tmp_path / "c.png"), rgb)
    assert np.array_equal(IO.read_png(str(tmp_path / "g.png")), gray)
    assert np.array_equal(IO.read_png(str(tmp_path / "c.png")), rgb)
    # a PNG whose rows use every filter type (what real encoders emit)
    h, w = 10, 16
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    raw = bytearray()
    prev = np.zeros(w * 3, dtype=np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = y % 5
        left = np.concatenate([np.zeros(3, np.int32), cur[:-3]])
        ul = np.concatenate([np.zeros(3, np.int32), prev[:-3]])
        if ft == 0: line = cur
        elif ft == 1: line = cur - left
        elif ft == 2: line = cur - prev
        elif ft == 3: line = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            line = cur - pred
        raw += bytes([ft]) + (line & 255).astype(np.uint8).tobytes()
        prev = cur
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    with open(tmp_path / "f.png", "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    assert np.array_equal(IO.read_png(str(tmp_path / "f.png")), img)
    # BMP, bottom-up 24-bit with row padding
    bw, bh = 5, 3
    bimg = rng.integers(0, 256, (bh, bw, 3), dtype=np.uint8)
    stride = (bw * 3 + 3) // 4 * 4
    rows = b"".join(bimg[y, :, ::-1].tobytes() + b"\0" * (stride - bw * 3) for y in range(bh - 1, -1, -1))
    with open(tmp_path / "x.bmp", "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", 54 + len(rows), 0, 0, 54) +
                struct.pack("<IiiHHIIiiII", 40, bw, bh, 1, 24, 0, len(rows), 2835, 2835, 0, 0) + rows)
    assert np.array_equal(IO.read_bmp(str(tmp_path / "x.bmp")), bimg)
    # grayscale conversion = OpenCV's fixed-point weights; the preprocess path reads files end to end
    g = IO.read_gray(str(tmp_path / "c.png"))
    want = ((rgb[..., 0].astype(int) * 4899 + rgb[..., 1].astype(int) * 9617 + rgb[..., 2].astype(int) * 1868 + 8192) >> 14)
    assert np.array_equal(g, want.astype(np.uint8))
    t = P.preprocess_image(str(tmp_path / "g.png"), target_size=None)
    assert torch.equal(t, O.preprocess_array(gray))
    # resize: identity at equal size, exact 2x up-sampling of a ramp, constant images stay constant
    assert IO.resize_linear_u8(gray, (53, 37)) is gray
    const = np.full((9, 7), 200, np.uint8)
    assert np.all(IO.resize_linear_u8(const, (256, 256)) == 200)
    ramp = (np.arange(8, dtype=np.uint8) * 16)[None, :].repeat(4, 0)
    up = IO.resize_linear_u8(ramp, (16, 8))
    # half-pixel centres: dst x -> src (x + 0.5) / 2 - 0.5 = -0.25, 0.25, 0.75, 1.25 ...
    assert up[0, :6].tolist() == [0, 4, 12, 20, 28, 36] and up.shape == (8, 16)
    down = IO.resize_linear_u8(gray, (26, 18))
    assert down.shape == (18, 26) and abs(float(down.mean()) - float(gray.mean())) < 6.0
    assert P.preprocess_image(gray).shape == (1, 1, 256, 256)
    with pytest.raises(ValueError, match="Could not read image"):
        (tmp_path / "bad.png").write_bytes(b"not a png")
        P.preprocess_image(str(tmp_path / "bad.png"))


def test_optical_flow_baseline_delegates_to_opencv_with_the_reference_parameters(monkeypatch):
    """evaluation_simple.py:76-103: Farneback(f0, f1, pyr_scale 0.5, levels 3, winsize 15, iterations 3,
    poly_n 5, poly_sigma 1.1, flags 0), then frame 0 remapped at grid + flow/2 (clipped), INTER_LINEAR,
    BORDER_REPLICATE.  OpenCV is not installed here, so a recording stand-in checks the call contract."""
    import sys
    import types
    from ai_based_frame_interpolation_amd import evaluation

    calls = {}
    fake = types.ModuleType("cv2")
    fake.INTER_LINEAR, fake.BORDER_REPLICATE = 1, 1

    def farneback(a, b, flow, **kw):
        calls["farneback"] = (a.copy(), b.copy(), flow, kw)
        f = np.zeros(a.shape + (2,), dtype=np.float32)
        f[..., 0], f[..., 1] = 4.0, -2.0        # 4 px right, 2 px up
        return f

    def remap(src, mx, my, interp, borderMode=None):
        calls["remap"] = (mx.copy(), my.copy(), interp, borderMode)
        return src[np.rint(my).astype(int), np.rint(mx).astype(int)]

    fake.calcOpticalFlowFarneback, fake.remap = farneback, remap
    monkeypatch.setitem(sys.modules, "cv2", fake)
    g = torch.Generator().manual_seed(3)
    f0 = torch.randint(0, 256, (2, 1, 12, 20), dtype=torch.uint8, generator=g)
    f1 = torch.randint(0, 256, (2, 1, 12, 20), dtype=torch.uint8, generator=g)
    out = evaluation._optical_flow_u8(f0, f1)
    assert out.shape == f0.shape and out.dtype == torch.uint8
    a, b, flow, kw = calls["farneback"]
    assert flow is None and np.array_equal(a, f0[1, 0].numpy()) and np.array_equal(b, f1[1, 0].numpy())
    assert kw == dict(pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.1, flags=0)
    mx, my, interp, border = calls["remap"]
    ys, xs = np.mgrid[0:12, 0:20].astype(np.float32)
    assert np.array_equal(mx, np.clip(xs + 2.0, 0, 19)) and np.array_equal(my, np.clip(ys - 1.0, 0, 11))
    assert interp == fake.INTER_LINEAR and border == fake.BORDER_REPLICATE
    exp = f0[1, 0].numpy()[np.clip(np.arange(12) - 1, 0, 11)][:, np.clip(np.arange(20) + 2, 0, 19)]
    assert np.array_equal(out[1, 0].numpy(), exp)
    with pytest.raises(RuntimeError, match="grayscale"):
        evaluation._optical_flow_u8(f0.repeat(1, 3, 1, 1), f1.repeat(1, 3, 1, 1))


def test_npy_frames_bypass_opencv_and_imread_none_falls_back(tmp_path, monkeypatch):
    """On a box WITH OpenCV (the reference lists it as a requirement): cv2.imread returns None for
    `.npy`, the format save_frames / the serving path use, so `.npy` never goes through cv2, and a None
    from cv2.imread falls through to the built-in readers instead of "Could not read image"."""
    import sys
    import types
    from ai_based_frame_interpolation_amd import imageio_lite, inference
    calls = []
    fake = types.ModuleType("cv2")
    fake.IMREAD_GRAYSCALE = 0
    fake.imread = lambda path, flag=None: calls.append(path)  # returns None, like OpenCV on an unknown format
    fake.resize = lambda img, size: imageio_lite.resize_linear_u8(img, size)
    monkeypatch.setitem(sys.modules, "cv2", fake)
    img = (np.arange(20 * 24).reshape(20, 24) % 251).astype(np.uint8)
    npy = tmp_path / "frame.npy"
    np.save(npy, img)
    t = inference.preprocess_image(str(npy), target_size=None)
    assert calls == [] and t.shape == (1, 1, 20, 24)
    assert torch.equal(t, torch.from_numpy(2.0 * (img.astype(np.float32) / 255.0) - 1.0)[None, None])
    pgm = tmp_path / "frame.pgm"
    with open(pgm, "wb") as f:
        f.write(b"P5\n24 20\n255\n" + img.tobytes())
    t2 = inference.preprocess_image(str(pgm), target_size=None)
    assert calls == [str(pgm)] and torch.equal(t2, t)
    with pytest.raises(ValueError, match="Could not read image"):
        inference.preprocess_image(str(tmp_path / "missing.png"))


def test_weight_fingerprint_sees_replaced_and_inference_tensors(seeded_sd):
    """The live-parameter fingerprint (unet.py): in-place edits, `.data =`, replaced Parameters and
    load_state_dict(assign=True) change it; parameters created under inference_mode do not make it raise;
    `.data.add_()` is the documented blind spot (own version counter) - refresh_weights() covers it."""
    import ai_based_frame_interpolation_amd as P
    m = P.FrameInterpolationUNet(bilinear=True)
    m.load_state_dict(seeded_sd)
    fp0 = m._current_fingerprint()
    assert m._current_fingerprint() == fp0
    p = m.unet.outc.conv.bias
    with torch.no_grad():
        p.add_(1.0)
    fp1 = m._current_fingerprint()
    assert fp1 != fp0
    p.data = p.data.clone()
    fp2 = m._current_fingerprint()
    assert fp2 != fp1
    m.unet.outc.conv.bias = torch.nn.Parameter(p.detach().clone())
    fp3 = m._current_fingerprint()
    assert fp3 != fp2
    m.load_state_dict({k: v.clone() for k, v in seeded_sd.items()}, assign=True)
    fp4 = m._current_fingerprint()
    assert fp4 != fp3 and m._ctx_dirty
    p = m.unet.outc.conv.bias
    p.data.add_(1.0)                      # documented blind spot ...
    assert m._current_fingerprint() == fp4
    m._ctx_dirty = False
    m.refresh_weights()                   # ... covered by the explicit call
    assert m._ctx_dirty
    with torch.inference_mode():
        m2 = P.FrameInterpolationUNet(bilinear=True)
    assert len(m2._current_fingerprint()) == len(fp0)  # no RuntimeError from `_version`


def test_y4m_round_trip_and_header_parsing(tmp_path):
    """Uncompressed YUV4MPEG2 container (imageio_lite.read_y4m / write_y4m): 4:2:0 with odd sizes, mono,
    frame rate and colourspace tags survive a round trip; malformed streams raise ValueError."""
    from ai_based_frame_interpolation_amd import imageio_lite as IO
    rng = np.random.default_rng(0)
    y = rng.integers(0, 256, (3, 17, 23), dtype=np.uint8)
    u = rng.integers(0, 256, (3, 9, 12), dtype=np.uint8)
    v = rng.integers(0, 256, (3, 9, 12), dtype=np.uint8)
    p = str(tmp_path / "a.y4m")
    IO.write_y4m(p, y, (u, v), fps=(30000, 1001), colourspace="420mpeg2")
    y2, ch, fps, cs = IO.read_y4m(p)
    assert np.array_equal(y2, y) and np.array_equal(ch[0], u) and np.array_equal(ch[1], v)
    assert fps == (30000, 1001) and cs == "420mpeg2"
    assert open(p, "rb").read(60).startswith(b"YUV4MPEG2 W23 H17 F30000:1001 Ip A1:1 C420mpeg2\nFRAME\n")
    q = str(tmp_path / "m.y4m")
    IO.write_y4m(q, y)
    y3, ch3, fps3, cs3 = IO.read_y4m(q)
    assert np.array_equal(y3, y) and ch3 is None and cs3 == "mono" and fps3 == (30, 1)
    bad = tmp_path / "bad.y4m"
    bad.write_bytes(b"YUV4MPEG2 W4 H4 F25:1 C420p10\nFRAME\n" + bytes(24))
    with pytest.raises(ValueError, match="bit depth|colourspace"):
        IO.read_y4m(str(bad))
    bad.write_bytes(b"YUV4MPEG2 W4 H4 F25:1 Cmono\nFRAME\n" + bytes(10))
    with pytest.raises(ValueError, match="truncated"):
        IO.read_y4m(str(bad))
    bad.write_bytes(b"RIFF....")
    with pytest.raises(ValueError):
        IO.read_y4m(str(bad))


# ---- optical_flow.py: restatement of the reference's third evaluator method (parity unpinned against OpenCV) ----
def _flow_texture(h, w, dx, dy, seed=0):
    """smooth random texture sampled at (x - dx, y - dy): the content moves by (+dx, +dy)"""
    import torch.nn.functional as F
    from ai_based_frame_interpolation_amd import optical_flow as OF
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(h + 40, w + 40, generator=g)
    k = OF._gauss_kernel_cv(19, 3.0, torch.float32, "cpu")
    big = F.conv2d(F.conv2d(x[None, None], k.view(1, 1, 1, -1)), k.view(1, 1, -1, 1))[0, 0]
    big = (big - big.min()) / (big.max() - big.min())
    H, W = big.shape
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    grid = torch.stack([(xs + 10 - dx) / (W - 1) * 2 - 1, (ys + 10 - dy) / (H - 1) * 2 - 1], -1)[None]
    out = F.grid_sample(big[None, None], grid, mode="bicubic", align_corners=True)[0, 0]
    return (out * 255).round().clamp(0, 255).to(torch.uint8)


@pytest.mark.parametrize("dx,dy", [(2.5, -1.5), (0.0, 3.0), (-4.0, 0.5)])
def test_farneback_restatement_recovers_a_translation(dx, dy):
    """The behaviour the algorithm must have (cv2's convention: prev(y, x) ~ next(y + flow_y, x + flow_x)): a
    smooth texture translated by (dx, dy) yields that flow in the interior, to a few hundredths of a pixel."""
    from ai_based_frame_interpolation_amd import optical_flow as OF
    a, b = _flow_texture(96, 128, 0, 0), _flow_texture(96, 128, dx, dy)
    flow = OF.calc_optical_flow_farneback(a, b)
    assert flow.shape == (96, 128, 2) and flow.dtype == torch.float32
    core = flow[20:-20, 20:-20]
    assert abs(core[..., 0].mean().item() - dx) < 0.05 and abs(core[..., 1].mean().item() - dy) < 0.05
    assert core[..., 0].std().item() < 0.05 and core[..., 1].std().item() < 0.05
    # identical frames: no motion (the last row / column count as "outside" in the update step, as in OpenCV,
    # which leaves a few hundredths of a pixel of flow in the window next to them)
    z = OF.calc_optical_flow_farneback(a, a)
    assert z[20:-20, 20:-20].abs().max().item() < 1e-3 and z.abs().max().item() < 0.2


def test_optical_flow_baseline_is_the_references_formula():
    """evaluation_simple.py:92-101 as written: frame 0 sampled at grid + flow / 2 (clipped), i.e. the content is
    moved by -flow / 2 - AGAINST its motion; so on a translating texture the baseline equals frame 0 shifted by
    -d / 2 and scores BELOW the linear blend.  remap with integer coordinates is the identity; fractional
    coordinates follow the 5-bit fixed-point bilinear rule."""
    from ai_based_frame_interpolation_amd import optical_flow as OF
    h, w, dx, dy = 96, 128, 4.0, 2.0
    a, b = _flow_texture(h, w, 0, 0), _flow_texture(h, w, dx, dy)
    out = OF.optical_flow_interpolation_baseline(a, b)
    assert out.shape == a.shape and out.dtype == torch.uint8
    back = _flow_texture(h, w, -dx / 2, -dy / 2)            # frame 0's content moved by -d/2
    true_mid = _flow_texture(h, w, dx / 2, dy / 2)
    c = (slice(16, -16), slice(16, -16))
    mse = lambda p, q: ((p[c].float() - q[c].float()) ** 2).mean().item()
    assert mse(out, back) < 2.0                               # ~1 code rms: it IS the backward-shifted frame
    lin = ((a.float() + b.float()) / 2).to(torch.uint8)
    assert mse(out, true_mid) > mse(lin, true_mid)            # the reference's quirk, kept
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    assert torch.equal(OF.remap_bilinear_u8(a, xs, ys), a)
    half = OF.remap_bilinear_u8(a, (xs + 0.5).clamp(0, w - 1), ys)
    want = ((a[:, :-1].long() * 16384 + a[:, 1:].long() * 16384 + 16384) >> 15).to(torch.uint8)
    assert torch.equal(half[:, :-1], want)
    q = OF.remap_bilinear_u8(a, (xs - 7.3).clamp(0, w - 1), (ys + 200).clamp(0, h - 1))   # clipped: replicated border
    assert torch.equal(q[:, :7], a[-1:, :1].expand(h, 7))


def test_evaluator_uses_the_restatement_without_opencv():
    from ai_based_frame_interpolation_amd import evaluation
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV is installed: the evaluator calls it instead")
    except ImportError:
        pass
    assert "parity unpinned" in evaluation.optical_flow_backend()
    a, b = _flow_texture(64, 80, 0, 0, seed=2), _flow_texture(64, 80, 3.0, 0.0, seed=2)
    f0, f1 = torch.stack([a, b])[:, None], torch.stack([b, a])[:, None]
    out = evaluation._optical_flow_u8(f0, f1)
    from ai_based_frame_interpolation_amd import optical_flow as OF
    assert out.shape == f0.shape and torch.equal(out[0, 0], OF.optical_flow_interpolation_baseline(a, b))
    with pytest.raises(RuntimeError, match="grayscale"):
        evaluation._optical_flow_u8(f0.repeat(1, 3, 1, 1), f1.repeat(1, 3, 1, 1))
