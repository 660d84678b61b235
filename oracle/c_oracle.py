"""ctypes wrapper around oracle/libunet_oracle.so -- TEST INFRASTRUCTURE (see unet_oracle.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-C", _HERE, "libunet_oracle.so"], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libunet_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.fio_unet_forward.restype = ctypes.c_int
    return _LIB


def _weight_table(sd):
    """92 float pointers in state-dict order, skipping the int64 counters (unet_oracle.c)."""
    arrs = [np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32)
            for k, v in sd.items() if not k.endswith("num_batches_tracked")]
    assert len(arrs) == 92, len(arrs)
    tab = (ctypes.c_void_p * 92)(*[a.ctypes.data for a in arrs])
    return tab, arrs


def unet_forward(sd, frame1, frame2, n_classes=1):
    f1 = np.ascontiguousarray(frame1.detach().cpu().numpy(), dtype=np.float32)
    f2 = np.ascontiguousarray(frame2.detach().cpu().numpy(), dtype=np.float32)
    b, cf, h, w = f1.shape
    out = np.empty((b, n_classes, h, w), dtype=np.float32)
    tab, keep = _weight_table(sd)
    rc = lib().fio_unet_forward(tab, f1.ctypes.data_as(ctypes.c_void_p),
                                f2.ctypes.data_as(ctypes.c_void_p),
                                out.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_int(b), ctypes.c_int(cf), ctypes.c_int(n_classes),
                                ctypes.c_int(h), ctypes.c_int(w))
    if rc != 0:
        raise RuntimeError(f"fio_unet_forward rc={rc}")
    del keep
    return out
