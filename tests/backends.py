"""Two ways to run the SAME kernel tests: the CPU lane emulator build (marker `emu`, runs anywhere) and the hipcc-built
product library on a real MI355X (marker `gpu`).  Both go through the C ABI of include/eegclip.h."""
import ctypes

import numpy as np
import pytest


class EmuBackend:
    name = "emu"

    def __init__(self):
        from hipemu import emu
        self.lib = emu.lib()
        self.stream = None

    def dev(self, a):
        return np.ascontiguousarray(a).copy()

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype)

    def ptr(self, h):
        return h.ctypes.data if h is not None else None

    def host(self, h):
        return np.array(h)

    def sync(self):
        pass


class GpuBackend:
    name = "gpu"

    def __init__(self):
        import torch
        from eeg_image_decode_amd import _lib
        self.torch = torch
        self.lib = _lib.lib()
        self.stream = torch.cuda.current_stream().cuda_stream

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def zeros(self, shape, dtype=np.float32):
        return self.torch.from_numpy(np.zeros(shape, dtype)).cuda()

    def ptr(self, h):
        return h.data_ptr() if h is not None else None

    def host(self, h):
        return h.detach().cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()


_cache = {}


def get(name):
    if name not in _cache:
        _cache[name] = EmuBackend() if name == "emu" else GpuBackend()
    return _cache[name]


BACKENDS = [pytest.param("emu", marks=pytest.mark.emu), pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def be(request):
    return get(request.param)


def ok(rc):
    assert rc == 0, f"C ABI call returned {rc}"


byref = ctypes.byref
