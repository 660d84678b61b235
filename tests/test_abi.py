"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/fiunet.h
declares (no compute calls without a GPU)."""
import ctypes
import os
import re

from ai_based_frame_interpolation_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "fiunet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fiunet_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _header_symbols() == sorted(_native.SYMBOLS)


def test_library_exports_every_declared_symbol(hip_lib_built):
    lib = ctypes.CDLL(hip_lib_built)
    for sym in _header_symbols():
        assert hasattr(lib, sym), sym
    lib.fiunet_abi_version.restype = ctypes.c_int
    hdr = open(os.path.join(ROOT, 'include', 'fiunet.h')).read()
    assert lib.fiunet_abi_version() == int(re.search(r'#define FIUNET_ABI_VERSION (\d+)', hdr).group(1))


def test_code_object_targets_gfx950(hip_lib_built):
    blob = open(hip_lib_built, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"sm_" not in blob


def test_argument_validation_without_gpu(hip_lib_built):
    """Pure host-side checks that never touch the device."""
    lib = _native.lib()
    lib.fiunet_last_error_string.restype = ctypes.c_char_p
    assert lib.fiunet_create(None, 0, 1, 1) != 0
    h = ctypes.c_void_p()
    assert lib.fiunet_create(ctypes.byref(h), 0, 2, 1) == 1  # frame_channels must be 1 or 3
    assert b"frame_channels" in lib.fiunet_last_error_string()
    # bilinear=False (round 4: built) gets past the argument checks and fails on the device query here (no GPU)
    assert lib.fiunet_create(ctypes.byref(h), 0, 1, 0) != 7
    assert lib.fiunet_workspace_bytes(None, 1, 64, 64, 0) == 0
    assert lib.fiunet_preprocess_u8(None, None, 4, None) == 1
