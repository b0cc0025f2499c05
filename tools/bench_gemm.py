"""micro-benchmark of eegclip_gemm_f32 on the encoder's shapes (HIP events); also the target of rocprofv3 --pmc runs"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib
D = _abi.dim
L = lib()
st = torch.cuda.current_stream().cuda_stream
def run(M, N, K, kind, split=1, reps=20, drop=0.0):
    if kind == "nt":   # X W^T
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Am, Ak, Bk, Bn = D(K), D(1), D(1), D(K)
    elif kind == "nn": # dY W
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); Am, Ak, Bk, Bn = D(K), D(1), D(N), D(1)
    else:              # tn: dY^T X
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); Am, Ak, Bk, Bn = D(1), D(M), D(N), D(1)
    C = torch.zeros(M, N, device="cuda"); bias = torch.randn(N, device="cuda")
    d = _abi.GemmDesc(M=M, N=N, K=K, A=A.data_ptr(), Am=Am, Ak=Ak, B=B.data_ptr(), Bk=Bk, Bn=Bn, C=C.data_ptr(), Cm=D(N), Cn=D(1), Cpre=None,
                      bias_n=bias.data_ptr(), bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=int(split > 1), act=0, drop_p=drop, seed=1, drop_site=0, split_k=split)
    for _ in range(3): L.eegclip_gemm_f32(ctypes.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.eegclip_gemm_f32(ctypes.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{kind} {M}x{N}x{K} sk{split} drop{drop}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s")
if __name__ == "__main__":
    for args in [(16384, 744, 250, "nt"), (16384, 250, 248, "nt"), (16384, 256, 250, "nt"), (16384, 250, 256, "nt"), (16384, 250, 250, "nt", 1, 20, 0.25), (16384, 256, 250, "nt", 1, 20, 0.25),
                 (16384, 250, 744, "nn"), (16384, 250, 256, "nn"), (16384, 256, 250, "nn"), (16384, 248, 250, "nn"),
                 (744, 250, 16384, "tn", 32), (250, 256, 16384, "tn", 32), (256, 250, 16384, "tn", 32), (1024, 1440, 256, "tn"),
                 (256, 1024, 1440, "nt", 16), (256, 1440, 1024, "nn", 16), (2048, 2048, 1024, "nt"), (4096, 4096, 4096, "nt", 1, 5)]:
        run(*args)
