"""Run the PRODUCT's host code (engines, launch plans, ClipLoss, optimizer, Pipe) on CPU tensors with the kernels executing under
the lane emulator (TEST ONLY).  Nothing in the product knows about this: the patch swaps the loaded library handle for the emulator
build of the same .hip sources, makes `require_cuda` a no-op and gives `torch.cuda.current_stream()` a dummy stream.  It is how the
`-m "not gpu"` suite covers host logic + kernel logic together, including 2-rank gloo runs of the data-parallel path."""
import contextlib
import types

import torch


@contextlib.contextmanager
def product_on_emulator():
    from hipemu import emu
    import eeg_image_decode_amd._lib as L
    import eeg_image_decode_amd.atms, eeg_image_decode_amd.loss, eeg_image_decode_amd.optim, eeg_image_decode_amd.plan  # noqa: E401
    import eeg_image_decode_amd.prior, eeg_image_decode_amd.retrieval, eeg_image_decode_amd.sdxl  # noqa: E401
    import sys
    mods = [m for n, m in sys.modules.items() if n.startswith("eeg_image_decode_amd.")]
    saved = []
    elib = emu.lib()
    dummy = types.SimpleNamespace(cuda_stream=None, wait_event=lambda *a, **k: None, wait_stream=lambda *a, **k: None)
    fake = {"lib": (lambda: elib), "require_cuda": (lambda t, name="tensor": t), "raw_stream": (lambda: None), "current_stream": (lambda: dummy)}
    # (use_stream / cuda_available are never reached on CPU: the side-stream paths are guarded by cuda_available(), False here)
    for m in mods:
        for k, v in fake.items():
            if hasattr(m, k):
                saved.append((m, k, getattr(m, k)))
                setattr(m, k, v)
    real_cs = torch.cuda.current_stream
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=None)
    try:
        yield elib
    finally:
        torch.cuda.current_stream = real_cs
        for m, k, v in saved:
            setattr(m, k, v)
