"""Diagnostic: per-stage share of wave time by phase, from the -DFIUNET_STAMP build.
FIUNET_LIB=ab/lib_stamp.so python tools/stamp_report.py"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import ai_based_frame_interpolation_amd as P
from ai_based_frame_interpolation_amd import _native
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = P.FrameInterpolationUNet(bilinear=True, precision="bf16")
with torch.no_grad():
    for n, p in m.named_parameters():
        if p.dim() == 4 and p.shape[-1] == 3: p.normal_(0, (2.0 / (p.shape[1] * 9)) ** 0.5)
m = m.to(dev).eval()
b, h, w = 8, 1080, 1920
f1 = torch.rand(b, 1, h, w, device=dev) * 2 - 1; f2 = torch.rand(b, 1, h, w, device=dev) * 2 - 1
for _ in range(3): m(f1, f2)
L = _native.lib()
buf = (ctypes.c_ulonglong * 8)()
L.fiunet_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
L.fiunet_debug_stamp_layer.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["total", "prolog", "mfma", "stepwait", "boundary", "epilog"]
print(f"{'stage':>5} {'waves':>9} {'cyc/wave':>9} " + " ".join(f"{n:>9}" for n in names[1:]))
for i in range(1, 18):
    L.fiunet_debug_stamp_layer(m._ctx._h, i)
    m(f1, f2)
    L.fiunet_debug_stamps(m._ctx._h, buf)
    v = list(buf)
    nw = max(v[6], 1); tot = max(v[0], 1)
    print(f"{i:5d} {nw:9d} {v[0] / nw:9.0f} " + " ".join(f"{100.0 * v[k] / tot:8.1f}%" for k in range(1, 6))
          + f"   up-dma {100.0 * (v[7] >> 32) / tot:5.1f}% up-lerp {100.0 * (v[7] & 0xffffffff) / tot:5.1f}%")
