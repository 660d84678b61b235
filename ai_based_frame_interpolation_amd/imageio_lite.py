"""Image file I/O for the inference helpers without OpenCV / imageio (neither is in this image).

The reference reads its frames with `cv2.imread(path, cv2.IMREAD_GRAYSCALE)`, resizes them with
`cv2.resize(image, (256, 256))` and writes results with `cv2.imwrite`
(/root/reference/model/inference.py:23,29,251,284).  This module is the host-side stand-in:

  read_gray(path)            PNG (8/16-bit gray, gray+alpha, RGB, RGBA, palette; non-interlaced), BMP
                             (24/32-bit and 8-bit palette), binary PGM/PPM, `.npy`; colour images are
                             converted with OpenCV's BGR2GRAY weights in its fixed-point form.
  resize_linear_u8(img, wh)  cv2.resize's INTER_LINEAR for uint8: half-pixel centres, edge clamp,
                             11-bit fixed-point coefficients and its two-stage rounding.
  write_png(path, img)       8-bit gray or RGB PNG.

Restated from OpenCV's published implementation (imgproc/resize.cpp `resizeGeneric_` with
`HResizeLinear` / `VResizeLinear<uchar,int,short>`, `INTER_RESIZE_COEF_BITS = 11`; color.cpp RGB2Gray
with `yuv_shift = 14` coefficients 4899 / 9617 / 1868).  OpenCV is not installed, so these are
**unpinned against cv2 itself**; the tests pin them to their own definition (identity at equal size,
exact values on hand-computed cases, PNG round trips).  Host glue, not part of the device hot path:
if `cv2` is importable the callers use it instead.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_PNG_SIG = b"\x89PNG\r\n\x1a\n"


def _to_gray(rgb: np.ndarray) -> np.ndarray:
    """OpenCV RGB2GRAY on uint8: (R*4899 + G*9617 + B*1868 + 8192) >> 14."""
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)


def _unfilter(raw: bytes, height: int, stride: int, bpp: int) -> np.ndarray:
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    pos = 0
    for y in range(height):
        ft = raw[pos]
        line = np.frombuffer(raw, dtype=np.uint8, count=stride, offset=pos + 1).astype(np.int32)
        pos += stride + 1
        if ft == 0:
            cur = line
        elif ft == 2:  # Up
            cur = (line + prev) & 255
        elif ft == 1:  # Sub: running sum per byte lane
            cur = line.copy()
            for c in range(bpp):
                cur[c::bpp] = np.cumsum(line[c::bpp]) & 255
        else:  # Average / Paeth need the already reconstructed left neighbour: byte-serial
            cur = np.zeros(stride, dtype=np.int32)
            ln, pv = line.tolist(), prev.tolist()
            res = [0] * stride
            for i in range(stride):
                a = res[i - bpp] if i >= bpp else 0
                b = pv[i]
                if ft == 3:
                    res[i] = (ln[i] + ((a + b) >> 1)) & 255
                elif ft == 4:
                    c = pv[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    res[i] = (ln[i] + pr) & 255
                else:
                    raise ValueError(f"bad PNG filter type {ft}")
            cur = np.asarray(res, dtype=np.int32)
        out[y] = cur
        prev = cur
    return out


def read_png(path: str) -> np.ndarray:
    """-> uint8 [H,W] (gray) or [H,W,3] (RGB; alpha dropped, palette expanded)."""
    data = open(path, "rb").read()
    if data[:8] != _PNG_SIG:
        raise ValueError("not a PNG file")
    pos, idat, plte = 8, [], None
    w = h = depth = ctype = interlace = None
    while pos < len(data):
        ln, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + ln]
        pos += 12 + ln
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
        elif tag == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    if interlace:
        raise ValueError("interlaced PNG is not supported")
    nch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    if depth not in (8, 16) or (ctype == 3 and depth != 8):
        raise ValueError(f"PNG bit depth {depth} / colour type {ctype} is not supported")
    bps = depth // 8
    px = _unfilter(zlib.decompress(b"".join(idat)), h, w * nch * bps, nch * bps).reshape(h, w, nch, bps)
    px = px[..., 0]  # 16-bit: keep the high byte (cv2.imread without IMREAD_ANYDEPTH also narrows to 8 bits)
    if ctype == 3:
        return plte[px[..., 0]]
    if ctype in (0, 4):
        return np.ascontiguousarray(px[..., 0])
    return np.ascontiguousarray(px[..., :3])


def read_bmp(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    if data[:2] != b"BM":
        raise ValueError("not a BMP file")
    off = struct.unpack("<I", data[10:14])[0]
    hdr, w, h, _, bpp, comp = struct.unpack("<IiiHHI", data[14:34])
    if comp not in (0, 3) or bpp not in (8, 24, 32):
        raise ValueError("compressed / unusual BMP is not supported")
    flip = h > 0
    h = abs(h)
    stride = ((w * bpp + 31) // 32) * 4
    rows = np.frombuffer(data, dtype=np.uint8, count=stride * h, offset=off).reshape(h, stride)
    if flip:
        rows = rows[::-1]
    if bpp == 8:
        pal = np.frombuffer(data, dtype=np.uint8, count=256 * 4, offset=14 + hdr).reshape(256, 4)[:, 2::-1]
        return np.ascontiguousarray(pal[rows[:, :w]])
    px = rows[:, :w * (bpp // 8)].reshape(h, w, bpp // 8)
    return np.ascontiguousarray(px[..., 2::-1])  # BGR(A) -> RGB


def _read_pnm(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    tok, pos = [], 0
    while len(tok) < 4:  # magic, width, height, maxval
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            pos = data.index(b"\n", pos) + 1
            continue
        end = pos
        while not data[end:end + 1].isspace():
            end += 1
        tok.append(data[pos:end]); pos = end
    pos += 1
    w, h = int(tok[1]), int(tok[2])
    ch = 3 if tok[0] == b"P6" else 1
    img = np.frombuffer(data, dtype=np.uint8, count=w * h * ch, offset=pos).reshape(h, w, ch)
    return img[..., 0] if ch == 1 else img


def read_gray(path: str):
    """cv2.imread(path, IMREAD_GRAYSCALE) for the formats above; None if the format is unknown."""
    ext = path.lower().rsplit(".", 1)[-1] if "." in path else ""
    if ext == "npy":
        img = np.load(path)
    elif ext == "png":
        img = read_png(path)
    elif ext == "bmp":
        img = read_bmp(path)
    elif ext in ("pgm", "ppm", "pnm"):
        img = _read_pnm(path)
    else:
        return None
    if img.ndim == 3:
        img = _to_gray(np.clip(img[..., :3], 0, 255).astype(np.uint8))
    return np.clip(img, 0, 255).astype(np.uint8)


def resize_linear_u8(img: np.ndarray, target_size) -> np.ndarray:
    """cv2.resize(img, (W, H)) with the default INTER_LINEAR on a uint8 [H,W] image."""
    tw, th = int(target_size[0]), int(target_size[1])
    sh, sw = img.shape[:2]
    if (sh, sw) == (th, tw):
        return img
    bits = 11
    one = 1 << bits

    def axis(dst_n, src_n):
        scale = src_n / dst_n
        f = ((np.arange(dst_n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)  # fx is a float
        i0 = np.floor(f).astype(np.int64)
        frac = f - i0.astype(np.float32)
        lo = i0 < 0
        frac[lo] = 0.0
        i0[lo] = 0
        hi = i0 >= src_n - 1
        frac[hi] = 0.0
        i0[hi] = src_n - 1
        i1 = np.minimum(i0 + 1, src_n - 1)
        # saturate_cast<short>(cvRound(w * 2048)): round half to even, like cvRound
        c1 = np.rint(frac.astype(np.float64) * one).astype(np.int64)
        c0 = np.rint((1.0 - frac).astype(np.float64) * one).astype(np.int64)
        return i0, i1, c0, c1

    x0, x1, a0, a1 = axis(tw, sw)
    y0, y1, b0, b1 = axis(th, sh)
    src = img.astype(np.int64)
    rows = src[:, x0] * a0 + src[:, x1] * a1                  # HResizeLinear: int, scaled by 2^11
    r0, r1 = rows[y0], rows[y1]
    # VResizeLinear<uchar,int,short>: ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    out = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def write_png(path: str, img: np.ndarray) -> None:
    """8-bit gray [H,W] or RGB [H,W,3] -> PNG (filter 0, zlib level 6)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim == 2:
        ctype, rowbytes = 0, img.shape[1]
    elif img.ndim == 3 and img.shape[2] == 3:
        ctype, rowbytes = 2, img.shape[1] * 3
    else:
        raise ValueError("expected uint8 [H,W] or [H,W,3]")
    h, w = img.shape[:2]
    raw = np.zeros((h, rowbytes + 1), dtype=np.uint8)
    raw[:, 1:] = img.reshape(h, rowbytes)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(_PNG_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)) + chunk(b"IEND", b""))


# ---- uncompressed video: YUV4MPEG2 (.y4m) ------------------------------------------------------------
# The reference writes its result videos with imageio.mimsave (codec via ffmpeg,
# /root/reference/model/inference.py:176-202) and has no video READER at all (main.py's `video` command
# imports a class that does not exist).  There is no codec in this image; Y4M is the
# uncompressed container every player / ffmpeg reads and writes (`ffmpeg -i in.mp4 in.y4m`), which makes
# FrameInterpolator.interpolate_video work on real video files: header line
# "YUV4MPEG2 W<w> H<h> F<num>:<den> [I<p>] [A<n>:<d>] [C<colourspace>]", then per frame "FRAME\n" + planes.
def read_y4m(path: str):
    """-> (y [N, H, W] uint8, chroma or None, fps (num, den), colourspace tag).  `chroma` is a pair of
    [N, Hc, Wc] uint8 arrays (U, V) for the 4:2:0 / 4:2:2 / 4:4:4 layouts, None for mono."""
    with open(path, "rb") as f:
        data = f.read()
    nl = data.index(b"\n")
    head = data[:nl].split(b" ")
    if head[0] != b"YUV4MPEG2":
        raise ValueError("not a YUV4MPEG2 stream")
    w = h = None
    fps, cs = (30, 1), "420jpeg"
    for tok in head[1:]:
        if tok[:1] == b"W":
            w = int(tok[1:])
        elif tok[:1] == b"H":
            h = int(tok[1:])
        elif tok[:1] == b"F":
            n, d = tok[1:].split(b":")
            fps = (int(n), int(d))
        elif tok[:1] == b"C":
            cs = tok[1:].decode()
    if not w or not h:
        raise ValueError("Y4M header without W/H")
    if cs.startswith("mono"):
        cw = ch = 0
    elif cs.startswith("420"):
        cw, ch = (w + 1) // 2, (h + 1) // 2
    elif cs.startswith("422"):
        cw, ch = (w + 1) // 2, h
    elif cs.startswith("444") and "alpha" not in cs:
        cw, ch = w, h
    else:
        raise ValueError(f"unsupported Y4M colourspace C{cs}")
    if any(c in cs for c in ("p10", "p12", "p14", "p16", "mono16")):
        raise ValueError(f"unsupported Y4M bit depth C{cs}")
    fsz = w * h + 2 * cw * ch
    ys, us, vs = [], [], []
    pos = nl + 1
    while pos < len(data):
        e = data.index(b"\n", pos)
        if not data[pos:e].startswith(b"FRAME"):
            raise ValueError("Y4M: FRAME marker expected")
        pos = e + 1
        if pos + fsz > len(data):
            raise ValueError("Y4M: truncated frame")
        fr = np.frombuffer(data, np.uint8, fsz, pos)
        ys.append(fr[:w * h].reshape(h, w))
        if cw:
            us.append(fr[w * h:w * h + cw * ch].reshape(ch, cw))
            vs.append(fr[w * h + cw * ch:].reshape(ch, cw))
        pos += fsz
    if not ys:
        raise ValueError("Y4M: no frames")
    chroma = (np.stack(us), np.stack(vs)) if cw else None
    return np.stack(ys), chroma, fps, cs


def write_y4m(path: str, y: np.ndarray, chroma=None, fps=(30, 1), colourspace: str = None) -> None:
    """y: [N, H, W] uint8; chroma: None (-> Cmono) or (U, V) planes as read_y4m returns them."""
    y = np.ascontiguousarray(y, dtype=np.uint8)
    n, h, w = y.shape
    cs = colourspace or ("mono" if chroma is None else "420jpeg")
    if (chroma is None) != cs.startswith("mono"):
        raise ValueError("chroma planes and colourspace tag disagree")
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{w} H{h} F{int(fps[0])}:{int(fps[1])} Ip A1:1 C{cs}\n".encode())
        for i in range(n):
            f.write(b"FRAME\n")
            f.write(y[i].tobytes())
            if chroma is not None:
                f.write(np.ascontiguousarray(chroma[0][i], dtype=np.uint8).tobytes())
                f.write(np.ascontiguousarray(chroma[1][i], dtype=np.uint8).tobytes())
