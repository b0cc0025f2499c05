"""Loader for libeegclip_hip.so.  There is NO fallback: a missing library or a missing GPU is an error."""
import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libeegclip_hip.so")
_lib = None


class EegclipError(RuntimeError):
    pass


def lib():
    """The hipcc-built kernel library (loaded once).  torch is imported first so that the library binds to the
    very libamdhip64 instance PyTorch-ROCm already loaded (same SONAME) -- streams and pointers are shared."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (must precede the dlopen, see docstring)
        if not os.path.exists(LIB_PATH):
            raise EegclipError(
                f"{LIB_PATH} is missing: build it with `python -m eeg_image_decode_amd.build` "
                "(hipcc --offload-arch=gfx950).  This package has no CPU or eager-PyTorch fallback.")
        l = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _abi.declare(l)
        v = l.eegclip_abi_version()
        if v != _abi.ABI_VERSION:
            raise EegclipError(f"ABI version mismatch: library {v}, python binding {_abi.ABI_VERSION}")
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise EegclipError(f"{what} failed with code {rc}" + (" (argument error)" if rc < 0 else " (hipError_t)"))


def require_cuda(t, name="tensor"):
    if not t.is_cuda:
        raise EegclipError(f"{name} must live on the GPU (got {t.device}); eeg_image_decode_amd has no CPU path")
    return t


def current_stream():
    """torch.cuda.current_stream() for the current device, without its device lookup: with no argument that call resolves the device through
    torch.cuda.is_available(), which reads an environment variable each time -- ~40 us per call, and the 11 calls of a training step were 0.45 ms
    of its 0.82 ms of host time (tools/host_profile.py)."""
    import torch
    get = getattr(torch._C, "_cuda_getDevice", None)          # (private accessor: fall back to the public one if a torch release drops it)
    return torch.cuda.current_stream(get() if get is not None else torch.cuda.current_device())


def raw_stream():
    """the current HIP stream handle (hipStream_t as an integer) the C ABI takes"""
    return current_stream().cuda_stream


_CUDA_OK = None


def cuda_available():
    """torch.cuda.is_available(), asked once (each call reads the environment: ~25 us)"""
    global _CUDA_OK
    if _CUDA_OK is None:
        import torch
        _CUDA_OK = bool(torch.cuda.is_available())
    return _CUDA_OK


class use_stream:
    """`with torch.cuda.stream(s)` without its two device lookups (see current_stream): make `s` the current stream, restore the previous one"""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        import torch
        self.prev = current_stream()
        torch.cuda.set_stream(self.stream)
        return self.stream

    def __exit__(self, *exc):
        import torch
        torch.cuda.set_stream(self.prev)
        return False
