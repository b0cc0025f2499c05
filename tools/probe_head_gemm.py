"""where do the ~15 us of a projection-head GEMM go?  gemm_x3 at M = 256 over K and split-K (kernel timestamps via rocprofv3 --kernel-trace, or events here)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib
D = _abi.dim
L = lib()
st = torch.cuda.current_stream().cuda_stream


def ev(f, n=50):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 2)


for M, N in ((256, 1024), (1024, 1024)):
    for K in (64, 256, 1024, 1440):
        for sk in (1, 2, 4, 8):
            if K // sk < 32:
                continue
            A, B, C = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.zeros(M, N, device="cuda")
            d = _abi.GemmDesc(M=M, N=N, K=K, A=A.data_ptr(), Am=D(K), Ak=D(1), B=B.data_ptr(), Bk=D(1), Bn=D(K), C=C.data_ptr(), Cm=D(N), Cn=D(1), Rm=D(0), Rn=D(0), alpha=1.0,
                              accumulate=int(sk > 1), split_k=sk, precision=_abi.PREC_BF16X3)
            import ctypes
            t = ev(lambda: L.eegclip_gemm_f32(ctypes.byref(d), st))
            print(f"M={M} N={N} K={K} sk={sk}: {t} us (back to back, incl. launch gaps)", flush=True)
