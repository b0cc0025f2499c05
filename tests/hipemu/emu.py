"""Load the emulator build of the kernels (TEST ONLY) with the same ctypes prototypes as the product."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import build_emu  # noqa: E402

from eeg_image_decode_amd import _abi  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = build_emu.build()
        _lib = _abi.declare(ctypes.CDLL(path))
    return _lib


def ptr(a):
    """numpy array -> 'device' pointer for the emulator (host memory)."""
    assert a.flags["C_CONTIGUOUS"] or a.ndim <= 1
    return a.ctypes.data
