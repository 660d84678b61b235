"""Host-side counterpart of the reference's model/inference.py hot-path helpers.

Same names, argument meaning and error behaviour as the reference functions they replace:
  preprocess_image   /root/reference/model/inference.py:11-41
  postprocess_image  /root/reference/model/inference.py:43-63
  load_model         /root/reference/model/inference.py:65-99
  interpolate_frames /root/reference/model/inference.py:101-122
  generate_multiple_intermediate_frames  inference.py:124-149
plus the `FrameInterpolator` class that the reference's main.py imports (main.py:100,122) but
never defines: `.interpolate_frames(img1, img2)` and
`.interpolate_video(input, output, factor)`.

The arithmetic of pre/post-processing and the network runs in HIP kernels; this file is
plumbing (file I/O, shapes, batching).  No cv2/imageio in this image: image files go through cv2 if
it happens to be importable, else through imageio_lite (PNG, BMP, PGM/PPM, `.npy`; cv2.resize's
fixed-point INTER_LINEAR restated); videos are raw `.npy` frame stacks [N,H,W] / [N,H,W,3] uint8.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from struct import error as struct_error
from zlib import error as zlib_error

from . import _native, imageio_lite
from .unet import FrameInterpolationUNet


def _read_gray(path: str) -> np.ndarray:
    """cv2.imread(path, cv2.IMREAD_GRAYSCALE) (inference.py:23): OpenCV when it is installed, else the
    readers of imageio_lite (PNG, BMP, PGM/PPM, .npy).  None when the file cannot be decoded."""
    img = None
    # formats OpenCV does not know (.npy frames are what save_frames and the serving path write) never go
    # through it; for everything else a None from cv2.imread falls through to the built-in readers
    if not str(path).lower().endswith(".npy"):
        try:
            import cv2  # type: ignore
            img = cv2.imread(path, cv2.IMREAD_GRAYSCALE)
        except ImportError:
            pass
    if img is not None:
        return img
    try:
        return imageio_lite.read_gray(path)
    except (ValueError, KeyError, struct_error, zlib_error, OSError):
        return None


def _resize_linear_u8(img: np.ndarray, target_size) -> np.ndarray:
    """cv2.resize(img, (W, H)) (inference.py:29, default INTER_LINEAR): OpenCV when installed, else its
    fixed-point algorithm restated in imageio_lite.  Host glue, not part of the device hot path."""
    try:
        import cv2  # type: ignore
        return cv2.resize(img, (int(target_size[0]), int(target_size[1])))
    except ImportError:
        return imageio_lite.resize_linear_u8(img, target_size)


def preprocess_image(image_path, target_size=(256, 256)):
    """Gray read -> resize to target_size (width, height) -> /255 -> 2x-1 -> [1,1,H,W] fp32.

    Accepts a path (reference behaviour, inference.py:23) or an already decoded uint8 array.
    Raises ValueError("Could not read image from ...") like inference.py:25-26."""
    if isinstance(image_path, np.ndarray):
        image = image_path
    else:
        image = _read_gray(image_path) if os.path.exists(str(image_path)) else None
        if image is None:
            raise ValueError(f"Could not read image from {image_path}")
    if target_size is not None:
        image = _resize_linear_u8(image, target_size)
    image = image.astype(np.float32) / 255.0
    image = 2.0 * image - 1.0
    return torch.from_numpy(image).unsqueeze(0).unsqueeze(0)


def postprocess_image(tensor: torch.Tensor) -> np.ndarray:
    """[-1,1] fp32 tensor -> uint8 image: (x+1)/2, clamp [0,1], *255, truncating cast
    (inference.py:54-61).  Runs the HIP post-processing kernel; the tensor must be on the GPU
    (interpolate_frames returns it there)."""
    if not tensor.is_cuda:
        raise RuntimeError("postprocess_image (MI355X build) expects the device tensor that "
                           "interpolate_frames returned; there is no CPU fallback")
    t = tensor.detach().to(torch.float32).contiguous()
    return _native.postprocess_u8(t).squeeze().cpu().numpy()


def load_model(model_path, device, precision=None, frame_channels=1):
    """Construct bilinear=True, load `{'model_state_dict': ...}` or a bare state-dict, move to
    device, eval (inference.py:77-97).  FileNotFoundError if the file is missing (:80-81)."""
    model = FrameInterpolationUNet(bilinear=True, frame_channels=frame_channels, precision=precision)
    if not os.path.exists(model_path):
        raise FileNotFoundError(f"Model file not found: {model_path}")
    checkpoint = torch.load(model_path, map_location="cpu")
    if "model_state_dict" in checkpoint:
        model.load_state_dict(checkpoint["model_state_dict"])
        print(f"Model loaded from {model_path}")
        print(f"Trained for {checkpoint.get('epoch', 'Unknown')} epochs")
        val = checkpoint.get("val_loss", "Unknown")
        print(f"Best validation loss: {val:.6f}" if isinstance(val, float) else
              f"Best validation loss: {val}")
    else:
        model.load_state_dict(checkpoint)
        print(f"Model state dict loaded from {model_path}")
    model = model.to(device)
    model.eval()
    return model


def interpolate_frames(model, frame1, frame2, device):
    """inference.py:115-120: move to device, no_grad, model(frame1, frame2)."""
    frame1 = frame1.to(device)
    frame2 = frame2.to(device)
    with torch.no_grad():
        return model(frame1, frame2)


def generate_multiple_intermediate_frames(model, frame1, frame2, num_intermediate, device):
    """inference.py:124-149: the reference runs the SAME pair N times (the network has no time
    input), so all N frames are identical; one forward is enough."""
    frame = interpolate_frames(model, frame1, frame2, device)
    return [frame for _ in range(num_intermediate)]


def _pair_batches(n_pairs: int, batch: int):
    for s in range(0, n_pairs, batch):
        yield s, min(batch, n_pairs - s)


def _forward_u8_chunk(model, a: torch.Tensor, b: torch.Tensor, batch: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """forward_u8 of one chunk of a sequence.  Below 1080p some layers of a forward have fewer workgroups
    than the chip has CUs and cut their K loop over several (split-K, fiunet.hip); how many depends on the
    batch (at 720p batches of fewer than five pairs still split the deepest level), so the fp32 summation
    order - hence a pixel sitting on a uint8 truncation boundary - of a pair may depend on how many pairs
    share its call.  A ragged chunk - the last one of a sequence, and the only one of a sequence shorter than
    a batch - is therefore padded (its last pair repeated, the extra outputs dropped) up to the smallest batch
    at which no layer splits (`model.batch_invariant_from`, the library's own rule: 1 from 1080p up, 5 at 720p),
    or to the full `batch` where even that still splits (256x256 at batch 8): every pair of a sequence is
    computed exactly as in a full batch, the result does not depend on the sequence length or on how the
    sequence is sharded over ranks (`sequence_pair_fn`) - and a one-pair 720p clip costs 5 forwards, not 8."""
    cnt = a.shape[0]
    target = cnt
    if cnt < batch:
        bmin = model.batch_invariant_from(a.shape[-2], a.shape[-1], a.device)
        target = batch if bmin > batch else max(cnt, bmin)
    if target > cnt:
        rep = [1] * a.dim()
        rep[0] = target - cnt
        a = torch.cat([a, a[-1:].repeat(*rep)])
        b = torch.cat([b, b[-1:].repeat(*rep)])
        res = model.forward_u8(a, b)[:cnt]
        if out is not None:
            out.copy_(res)
            return out
        return res
    # `out` (the video loops: every second frame of the interleaved result): the fused head stores each frame in place
    return model.forward_u8(a, b, out=out)


def sequence_pair_fn(model, batch: int = 8):
    """`pair_fn` for `video.interpolate_video_sharded`: uint8 `[b, H, W]` frame stacks in, uint8 middles out,
    through the same chunk helper as the single-process loops, so a rank's ragged sub-batches of small
    frames are padded to `batch` exactly as `interpolate_sequence` pads its own (use the same `batch` for
    both and the sharded result equals the single-process one bit for bit at every frame size)."""
    @torch.no_grad()
    def pair_fn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        if a.dim() == 3:
            return _forward_u8_chunk(model, a.unsqueeze(1), b.unsqueeze(1), batch,
                                     None if out is None else out.unsqueeze(1)).squeeze(1)
        return _forward_u8_chunk(model, a, b, batch, out)
    pair_fn.accepts_out = True   # video.interpolate_video_sharded: the root's own middles go straight into the interleaved result
    return pair_fn


@torch.no_grad()
def interpolate_sequence(model, frames_u8: torch.Tensor, batch: int = 8) -> torch.Tensor:
    """factor-2 video loop on one GPU: device uint8 frames [N,H,W] (or [N,C,H,W]) ->
    [2N-1, ...] = F0, M0, F1, M1, ..., F(N-1), where Mi = model(Fi, Fi+1) through the fused
    uint8 path (pre/post-processing on device).  Semantics per SURVEY.md 8a row 11."""
    squeeze = frames_u8.dim() == 3
    fr = frames_u8.unsqueeze(1) if squeeze else frames_u8
    n = fr.shape[0]
    out = torch.empty((2 * n - 1,) + tuple(fr.shape[1:]), dtype=torch.uint8, device=fr.device)
    out[0::2] = fr
    for s, cnt in _pair_batches(n - 1, batch):   # each middle is written where it belongs (no temporary, no strided copy)
        _forward_u8_chunk(model, fr[s:s + cnt], fr[s + 1:s + cnt + 1], batch, out=out[2 * s + 1: 2 * (s + cnt): 2])
    return out.squeeze(1) if squeeze else out


@torch.no_grad()
def interpolate_sequence_host(model, frames_u8_cpu: torch.Tensor, batch: int = 8,
                              out: torch.Tensor | None = None) -> torch.Tensor:
    """factor-2 video loop for frames that live in HOST memory: uint8 [N,H,W] (CPU) -> uint8
    [2N-1,H,W] (CPU, pinned).  Chunks of `batch` pairs are double-buffered: while the GPU runs
    chunk i on the compute stream, chunk i+1 goes host->device and the results of chunk i-1 go
    device->host on a copy stream (pinned buffers, PCIe Gen5).  Same result as
    interpolate_sequence(model, frames.cuda()).cpu().  Pass a pinned `out` [2N-1,H,W] to reuse it
    across calls (pinning 1.6 GB for 400 1080p frames costs more than interpolating them)."""
    dev = next(model.parameters()).device
    n = frames_u8_cpu.shape[0]
    src = frames_u8_cpu if frames_u8_cpu.is_pinned() else frames_u8_cpu.contiguous().pin_memory()
    if out is None:
        out = torch.empty((2 * n - 1,) + tuple(src.shape[1:]), dtype=torch.uint8).pin_memory()
    compute, copy = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
    chunks = list(_pair_batches(n - 1, batch))
    dbuf, dmid, up_done, comp_done = {}, {}, {}, {}

    def upload(i):
        s, cnt = chunks[i]
        with torch.cuda.stream(copy):
            dbuf[i] = src[s:s + cnt + 1].to(dev, non_blocking=True).unsqueeze(1)
            up_done[i] = torch.cuda.Event(); up_done[i].record(copy)

    def download(i):
        s, cnt = chunks[i]
        with torch.cuda.stream(copy):
            copy.wait_event(comp_done[i])
            dmid[i].record_stream(copy)  # allocated on the compute stream, read by the copy stream
            for j in range(cnt):  # one contiguous 2-MB copy per frame (a strided view would be staged)
                out[2 * (s + j) + 1].copy_(dmid[i][j, 0], non_blocking=True)
        dbuf.pop(i, None)

    if chunks:
        upload(0)
    for i in range(len(chunks)):
        if i + 1 < len(chunks):
            upload(i + 1)
        compute.wait_event(up_done[i])
        fr = dbuf[i]
        dmid[i] = _forward_u8_chunk(model, fr[:-1], fr[1:], batch)
        comp_done[i] = torch.cuda.Event(); comp_done[i].record(compute)
        fr.record_stream(compute)
        download(i)
        if i >= 1:
            dmid.pop(i - 1, None)
    out[0::2] = src  # host-side interleave of the original frames, while the GPU is still busy
    copy.synchronize()
    torch.cuda.current_stream(dev).synchronize()
    return out


def _interleave_average_u8(planes: torch.Tensor) -> torch.Tensor:
    """[N, h, w] uint8 -> [2N-1, h, w]: the originals with the rounded average of each neighbouring pair
    in between (chroma of an inserted frame)."""
    n = planes.shape[0]
    out = torch.empty((2 * n - 1,) + tuple(planes.shape[1:]), dtype=torch.uint8, device=planes.device)
    out[0::2] = planes
    if n > 1:
        out[1::2] = ((planes[:-1].to(torch.int16) + planes[1:].to(torch.int16) + 1) >> 1).to(torch.uint8)
    return out


class FrameInterpolator:
    """What main.py:95-129 expects from `model.inference` (it is missing in the reference).

    interpolate_frames(img1, img2): uint8 [H,W] (gray) or [H,W,3] images -> uint8 image of the
    same shape; colour images are processed per channel with the 2->1 grayscale network unless
    the checkpoint is the 6->3 variant.
    interpolate_video(input_path, output_path, factor=2): raw .npy frame stack or uncompressed
    YUV4MPEG2 (`.y4m`) video in/out; factor must be a power of two (recursive bisection; factor 2 is the only semantics the
    reference's flags imply, main.py:57-62)."""

    def __init__(self, model_path=None, device="cuda", precision=None, model=None, batch=8):
        self.device = torch.device("cuda" if device in ("auto", None) else device)
        self.model = model if model is not None else load_model(model_path, self.device, precision)
        self.batch = batch

    def _as_planes(self, img: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
        if t.dim() == 2:
            return t[None, None]
        planes = t.permute(2, 0, 1)  # [C,H,W]
        return planes[None] if self.model.frame_channels == planes.shape[0] else planes[:, None]

    def interpolate_frames(self, img1: np.ndarray, img2: np.ndarray) -> np.ndarray:
        if img1.shape != img2.shape or img1.dtype != np.uint8:
            raise ValueError("expected two uint8 images of equal shape")
        a, b = self._as_planes(img1), self._as_planes(img2)
        o = self.model.forward_u8(a, b)
        if img1.ndim == 2:
            return o[0, 0].cpu().numpy()
        o = o[0] if self.model.frame_channels == img1.shape[2] else o[:, 0]
        return o.permute(1, 2, 0).contiguous().cpu().numpy()

    def _interpolate_y4m(self, input_path, output_path, factor):
        """Uncompressed YUV4MPEG2 in -> out (`ffmpeg -i in.mp4 in.y4m` makes one; no codec exists in this
        image).  The network is the reference's grayscale 2->1 model, so it interpolates the LUMA plane;
        the chroma planes of an inserted frame are the rounded average of its neighbours' (an extension:
        the reference has no colour or video path to be faithful to).  The frame rate is multiplied by
        `factor`.  Output: `.y4m`, or a `.npy` stack of the luma frames."""
        if self.model.frame_channels != 1:
            raise ValueError("Y4M video goes through the grayscale (2->1) network")
        y, chroma, fps, cs = imageio_lite.read_y4m(input_path)
        t = torch.from_numpy(y).to(self.device)
        cu = cv = None
        if chroma is not None:
            cu, cv = (torch.from_numpy(c).to(self.device) for c in chroma)
        f = factor
        while f > 1:
            t = interpolate_sequence(self.model, t, self.batch)
            if cu is not None:
                cu, cv = (_interleave_average_u8(c) for c in (cu, cv))
            f //= 2
        if str(output_path).lower().endswith(".y4m"):
            imageio_lite.write_y4m(output_path, t.cpu().numpy(),
                                   None if cu is None else (cu.cpu().numpy(), cv.cpu().numpy()),
                                   (fps[0] * factor, fps[1]), cs)
        else:
            np.save(output_path, t.cpu().numpy())
        return t.shape[0]

    def interpolate_video(self, input_path, output_path, factor=2):
        if factor < 2 or factor & (factor - 1):
            raise ValueError("factor must be a power of two (the network has no time input)")
        if not os.path.exists(input_path):
            raise FileNotFoundError(f"Video file not found: {input_path}")
        if str(input_path).lower().endswith(".y4m"):
            return self._interpolate_y4m(input_path, output_path, factor)
        frames = np.load(input_path)
        if frames.dtype != np.uint8 or frames.ndim not in (3, 4):
            raise ValueError("expected a uint8 .npy stack [N,H,W] or [N,H,W,3]")
        t = torch.from_numpy(frames).to(self.device)
        if t.dim() == 4:
            t = t.permute(0, 3, 1, 2).contiguous()
            if self.model.frame_channels == 1:  # per-channel application of the 2->1 network
                n, c, h, w = t.shape
                t = t.permute(1, 0, 2, 3).reshape(c * n, h, w)
                outs = []
                for ci in range(c):
                    seq = t[ci * n:(ci + 1) * n]
                    f = factor
                    while f > 1:
                        seq = interpolate_sequence(self.model, seq, self.batch); f //= 2
                    outs.append(seq)
                res = torch.stack(outs, dim=-1)
                np.save(output_path, res.cpu().numpy())
                return res.shape[0]
        f = factor
        while f > 1:
            t = interpolate_sequence(self.model, t, self.batch); f //= 2
        if t.dim() == 4:
            t = t.permute(0, 2, 3, 1)
        np.save(output_path, t.cpu().numpy())
        return t.shape[0]
