"""ctypes binding of libfiunet_hip.so (C ABI declared in include/fiunet.h).

PyTorch is only the plumbing here: it owns device memory (caching allocator) and the HIP
stream; every arithmetic op of the forward runs in the hand-written HIP kernels behind this
ABI.  There is NO fallback: if the shared library is missing or a call fails, a RuntimeError
is raised (the reference raises RuntimeError from torch for bad shapes too).

`import torch` must happen before the library is loaded so that the process-wide HIP runtime
is the one PyTorch-ROCm ships (same SONAME libamdhip64.so.7); the library then shares
torch's device context, allocator pointers and streams.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FIUNET_LIB") or os.path.join(_PKG, "libfiunet_hip.so")  # FIUNET_LIB: A/B builds
CSRC = os.path.join(_PKG, "csrc")

ABI_VERSION = 5        # include/fiunet.h FIUNET_ABI_VERSION this binding is written for
ABI_MIN_COMPAT = 4     # oldest A/B library (FIUNET_LIB) whose shared entry points have today's signatures
FP32, BF16, BF16X2 = 0, 1, 2   # include/fiunet.h: enum fiunet_precision
OPT_UNFUSED, OPT_KEEP_ALL, OPT_PAIR_TILES, OPT_GATHER_UPSAMPLE = 1, 2, 8, 16
OPT_RNE_WEIGHTS, OPT_NO_DITHER = 32, 64

#: every symbol include/fiunet.h declares (tests/test_abi.py checks the header against this)
SYMBOLS = (
    "fiunet_abi_version", "fiunet_last_error_string", "fiunet_create", "fiunet_destroy",
    "fiunet_set_options", "fiunet_load_weights", "fiunet_prepare_precision", "fiunet_workspace_bytes", "fiunet_forward",
    "fiunet_min_unsplit_batch",
    "fiunet_forward_strip",
    "fiunet_workspace_bytes_u8", "fiunet_forward_u8", "fiunet_forward_u8_strided", "fiunet_preprocess_u8",
    "fiunet_postprocess_u8", "fiunet_debug_read_activation", "fiunet_profile_enable",
    "fiunet_profile_read", "fiunet_metrics_workspace_bytes", "fiunet_psnr_u8", "fiunet_ssim_u8",
    "fiunet_ssim_gauss_workspace_bytes", "fiunet_ssim_gauss_f32",
)

_lib = None


def build(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    res = subprocess.run(["make", "-C", CSRC], capture_output=True, text=True)
    if res.returncode != 0 or not os.path.exists(LIB_PATH):
        raise RuntimeError("building libfiunet_hip.so failed:\n" + res.stdout + res.stderr)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the MI355X HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C "
            f"{CSRC}`).  There is no CPU fallback for this path.")
    L = ctypes.CDLL(LIB_PATH)
    # the argument lists below are this ABI version's: a library of another version would take misaligned arguments
    # silently (v4 inserted a parameter into fiunet_debug_read_activation under the same symbol name)
    L.fiunet_abi_version.restype = ctypes.c_int
    ver = int(L.fiunet_abi_version())
    ab = bool(os.environ.get("FIUNET_LIB"))
    if ver != ABI_VERSION and not (ab and ABI_MIN_COMPAT <= ver < ABI_VERSION):
        raise RuntimeError(f"{LIB_PATH} implements fiunet ABI v{ver}; this binding is written for v{ABI_VERSION}"
                           + (f" (A/B libraries from v{ABI_MIN_COMPAT} on are accepted)" if ab else "") + ": rebuild it")
    if ab:
        # an A/B build of an older source state (tools/ab_bench.py) may predate the newest entry points:
        # bind what it has; calling a missing one still fails loudly (AttributeError)
        have = [n for n in SYMBOLS if hasattr(L, n)]
        if len(have) != len(SYMBOLS):
            class _Partial:
                def __init__(self, lib): self.__dict__["_lib"] = lib
                def __getattr__(self, n):
                    if n in SYMBOLS and n not in have:
                        class _Missing:
                            def __setattr__(self, *a): pass
                            def __call__(self, *a): raise AttributeError(f"{LIB_PATH} does not export {n}")
                        return _Missing()
                    return getattr(self._lib, n)
            L = _Partial(L)
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    L.fiunet_abi_version.restype = ci
    L.fiunet_last_error_string.restype = ctypes.c_char_p
    L.fiunet_create.argtypes = [ctypes.POINTER(vp), ci, ci, ci]
    L.fiunet_destroy.argtypes = [vp]
    L.fiunet_set_options.argtypes = [vp, ctypes.c_uint]
    L.fiunet_load_weights.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_char_p),
                                      ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int64)]
    L.fiunet_prepare_precision.argtypes = [vp, ci]
    L.fiunet_workspace_bytes.argtypes = [vp, ci, ci, ci, ci]
    L.fiunet_workspace_bytes.restype = sz
    L.fiunet_min_unsplit_batch.argtypes = [vp, ci, ci, ci]
    L.fiunet_min_unsplit_batch.restype = ci
    L.fiunet_workspace_bytes_u8.argtypes = [vp, ci, ci, ci, ci]
    L.fiunet_workspace_bytes_u8.restype = sz
    L.fiunet_forward.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, sz, vp]
    L.fiunet_forward_strip.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, sz, vp]
    L.fiunet_forward_u8.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp, sz, vp]
    L.fiunet_forward_u8_strided.argtypes = [vp, vp, vp, vp, sz, ci, ci, ci, ci, vp, sz, vp]
    L.fiunet_preprocess_u8.argtypes = [vp, vp, sz, vp]
    L.fiunet_postprocess_u8.argtypes = [vp, vp, sz, vp]
    L.fiunet_debug_read_activation.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, sz,
                                               ctypes.POINTER(ci), vp]
    L.fiunet_metrics_workspace_bytes.argtypes = [ci, ci, ci]
    L.fiunet_metrics_workspace_bytes.restype = sz
    L.fiunet_psnr_u8.argtypes = [vp, vp, ci, ci, ci, vp, vp, sz, vp]
    L.fiunet_ssim_u8.argtypes = [vp, vp, ci, ci, ci, vp, vp, sz, vp]
    L.fiunet_ssim_gauss_workspace_bytes.argtypes = [ci, ci, ci]
    L.fiunet_ssim_gauss_workspace_bytes.restype = sz
    L.fiunet_ssim_gauss_f32.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, vp, vp, sz, vp]
    L.fiunet_profile_enable.argtypes = [vp, ci]
    L.fiunet_profile_read.argtypes = [vp, ctypes.POINTER(ci), ctypes.POINTER(ctypes.c_float),
                                      ctypes.POINTER(ctypes.c_double), ctypes.c_char_p, ci]
    if not os.environ.get("FIUNET_LIB"):
        for name in SYMBOLS:
            getattr(L, name)  # AttributeError here = the .so does not export what the header declares
    _lib = L
    return L


ERR_UNSUPPORTED = 7   # include/fiunet.h: enum fiunet_status


class NativeError(RuntimeError):
    """A non-OK status from the C ABI; `.status` is the fiunet_status code (the message carries the library's text)."""

    def __init__(self, what: str, status: int, text: str):
        super().__init__(f"{what} failed (status {status}): {text}")
        self.status = status


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fiunet_last_error_string()
        raise NativeError(what, rc, msg.decode() if msg else "?")


class Context:
    """Owns one fiunet_ctx (device-resident prepared weights) on one GPU."""

    def __init__(self, device_index: int, frame_channels: int = 1, bilinear: bool = True):
        self._h = ctypes.c_void_p()
        check(lib().fiunet_create(ctypes.byref(self._h), device_index, frame_channels,
                                  1 if bilinear else 0), "fiunet_create")
        self.device_index = device_index
        self.frame_channels = frame_channels
        self._prepared = set()   # precisions whose extra weight copies exist (fiunet_prepare_precision)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            lib().fiunet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_options(self, flags: int):
        check(lib().fiunet_set_options(self._h, flags), "fiunet_set_options")

    def force_cfg(self, layer: int, tile: int = 0, ksplit: int = 0) -> None:
        """Diagnostic (tests, tools/cfg_sweep.py; not part of the ABI): override the launch configuration of conv
        `layer` (1..17; < 0 clears all) - tile 0 = choose, 1 = the tuned tile, 2 = the small tile, 3 = the in-workgroup
        K cut where the launch has its form; ksplit 0 = choose, k = cut the K loop k ways where the launch can be cut."""
        fn = lib().fiunet_debug_force_cfg
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        check(fn(self._h, layer, tile, ksplit), "fiunet_debug_force_cfg")

    def load_state_dict(self, sd) -> None:
        names, ptrs, numels, keep = [], [], [], []
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                continue
            t = v.detach().to(device="cpu", dtype=torch.float32).contiguous()
            keep.append(t)
            names.append(k.encode())
            ptrs.append(t.data_ptr())
            numels.append(t.numel())
        n = len(names)
        check(lib().fiunet_load_weights(
            self._h, n, (ctypes.c_char_p * n)(*names), (ctypes.c_void_p * n)(*ptrs),
            (ctypes.c_int64 * n)(*numels)), "fiunet_load_weights")
        self._prepared = set()

    def prepare(self, precision: int) -> None:
        """Weight copies a precision needs beyond the load's (bf16x2: the two-piece copies, built on first use so
        that fp32 / bf16 users pay neither their memory nor their packing time).  Allocates: not under capture."""
        if precision not in self._prepared:
            check(lib().fiunet_prepare_precision(self._h, precision), "fiunet_prepare_precision")
            self._prepared.add(precision)

    def workspace_bytes(self, b, h, w, precision, u8=False) -> int:
        self.prepare(precision)   # every forward path sizes its workspace first
        fn = lib().fiunet_workspace_bytes_u8 if u8 else lib().fiunet_workspace_bytes
        n = fn(self._h, b, h, w, precision)
        if n == 0:
            if h < 16 or w < 16:
                raise RuntimeError(f"input {h}x{w} is too small for four 2x2 max-pools (need >= 16)")
            if h * w >= 1 << 26:
                raise RuntimeError(f"input {h}x{w} has 2^26 pixels or more: cut it into row bands "
                                   "(tiling.forward_tiled / forward_strip)")
            check(1, "fiunet_workspace_bytes")
        return n

    def min_unsplit_batch(self, h, w, precision) -> int:
        """Smallest batch at which no layer of an h x w forward K-splits (65: none up to 64)."""
        n = lib().fiunet_min_unsplit_batch(self._h, h, w, precision)
        if n == 0:
            check(1, "fiunet_min_unsplit_batch")
        return n

    def forward(self, f1, f2, out, precision, workspace, stream=None):
        b, _, h, w = f1.shape
        s = torch.cuda.current_stream(f1.device).cuda_stream if stream is None else stream
        check(lib().fiunet_forward(self._h, f1.data_ptr(), f2.data_ptr(), out.data_ptr(), b, h, w,
                                   precision, workspace.data_ptr(), workspace.numel(), s),
              "fiunet_forward")

    def forward_strip(self, f1, f2, out, y_origin, h_image, precision, workspace, stream=None):
        """The forward on rows [y_origin, y_origin + h) of an image of h_image rows."""
        b, _, h, w = f1.shape
        s = torch.cuda.current_stream(f1.device).cuda_stream if stream is None else stream
        check(lib().fiunet_forward_strip(self._h, f1.data_ptr(), f2.data_ptr(), out.data_ptr(), b, h,
                                         w, y_origin, h_image, precision, workspace.data_ptr(),
                                         workspace.numel(), s), "fiunet_forward_strip")

    def forward_u8(self, f1, f2, out, precision, workspace, stream=None):
        """`out`: uint8 [B, C, H, W] whose images are contiguous; they may lie further apart than one image (a strided
        view such as every second frame of the video loop's interleaved result): the fused head writes them in place."""
        b, c, h, w = f1.shape
        s = torch.cuda.current_stream(f1.device).cuda_stream if stream is None else stream
        if out.is_contiguous():
            check(lib().fiunet_forward_u8(self._h, f1.data_ptr(), f2.data_ptr(), out.data_ptr(), b, h,
                                          w, precision, workspace.data_ptr(), workspace.numel(), s),
                  "fiunet_forward_u8")
            return
        st = out.stride()
        if tuple(st[1:]) != (h * w, w, 1) or (b > 1 and st[0] < c * h * w):
            raise ValueError(f"out: every image must be contiguous (strides {tuple(st)} for shape {tuple(out.shape)})")
        check(lib().fiunet_forward_u8_strided(self._h, f1.data_ptr(), f2.data_ptr(), out.data_ptr(), st[0], b, h,
                                              w, precision, workspace.data_ptr(), workspace.numel(), s),
              "fiunet_forward_u8_strided")

    def profile_enable(self, on: bool):
        check(lib().fiunet_profile_enable(self._h, 1 if on else 0), "fiunet_profile_enable")

    def profile_read(self):
        """-> (n_forwards, [(kernel name, avg ms, algorithmic flops)] for the 18 conv stages)"""
        n = ctypes.c_int(0)
        ms = (ctypes.c_float * 18)()
        fl = (ctypes.c_double * 18)()
        names = ctypes.create_string_buffer(18 * 96)
        check(lib().fiunet_profile_read(self._h, ctypes.byref(n), ms, fl, names, 96),
              "fiunet_profile_read")
        rows = []
        for i in range(18):
            nm = names.raw[i * 96:(i + 1) * 96].split(b"\0", 1)[0].decode()
            rows.append((nm, float(ms[i]), float(fl[i])))
        return n.value, rows

    def read_activation(self, workspace, b, h, w, precision, tap):
        dims = (ctypes.c_int * 3)()
        # the library knows the architecture (the ConvTranspose2d decoder is wider at taps 8, 9, 11, 13, 15): ask it
        # for the tap's dims first (dst = NULL), then hand over a buffer of exactly that size WITH its capacity
        check(lib().fiunet_debug_read_activation(self._h, None, b, h, w, precision, tap, None, 0, dims, None),
              "fiunet_debug_read_activation (dims query)")
        c, hh, ww = tuple(dims)
        dst = torch.empty((b, c, hh, ww), dtype=torch.float32, device=workspace.device)
        s = torch.cuda.current_stream(workspace.device).cuda_stream
        check(lib().fiunet_debug_read_activation(self._h, workspace.data_ptr(), b, h, w, precision,
                                                 tap, dst.data_ptr(), dst.numel(), dims, s),
              "fiunet_debug_read_activation")
        return dst


def preprocess_u8(src_u8: "torch.Tensor") -> "torch.Tensor":
    out = torch.empty(src_u8.shape, dtype=torch.float32, device=src_u8.device)
    s = torch.cuda.current_stream(src_u8.device).cuda_stream
    check(lib().fiunet_preprocess_u8(src_u8.data_ptr(), out.data_ptr(), src_u8.numel(), s),
          "fiunet_preprocess_u8")
    return out


def postprocess_u8(src_f32: "torch.Tensor") -> "torch.Tensor":
    out = torch.empty(src_f32.shape, dtype=torch.uint8, device=src_f32.device)
    s = torch.cuda.current_stream(src_f32.device).cuda_stream
    check(lib().fiunet_postprocess_u8(src_f32.data_ptr(), out.data_ptr(), src_f32.numel(), s),
          "fiunet_postprocess_u8")
    return out
