"""the projection head's four GEMM shapes at B = 256 on csrc/gemm_planes.hip (operands as planes, unsplit) against today's split-K gemm_x3 launches"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib
from eeg_image_decode_amd.plan import D
L = lib()

def ev_us(fn, reps=200, warm=10):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

st = torch.cuda.current_stream().cuda_stream
for M, N, K in ((256, 1024, 1440), (256, 1024, 1024), (256, 1472, 1024), (256, 1024, 512)):
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
    ah, al, bh, bl = (torch.empty(t.shape, dtype=torch.bfloat16, device="cuda") for t in (a, a, b, b))
    L.eegclip_split_bf16(a.data_ptr(), ah.data_ptr(), al.data_ptr(), a.numel(), st)
    L.eegclip_split_bf16(b.data_ptr(), bh.data_ptr(), bl.data_ptr(), b.numel(), st)
    c = torch.zeros(M, N, device="cuda")
    ph, plo = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    bias = torch.randn(N, device="cuda")
    d = _abi.GemmPlanesDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), lda=K, ldb=K, M=M, N=N, K=K, C=c.data_ptr(), ldc=N,
                            bias=bias.data_ptr(), p_hi=ph.data_ptr(), p_lo=plo.data_ptr(), ldp=N, planes_of=1)
    assert L.eegclip_gemm_planes(d, st) == 0
    res = {"planes_us": round(ev_us(lambda: L.eegclip_gemm_planes(d, st)), 2)}
    for sk in (1, 8):
        g = _abi.GemmDesc(M=M, N=N, K=K, A=a.data_ptr(), Am=D(K), Ak=D(1), B=b.data_ptr(), Bk=D(1), Bn=D(K), C=c.data_ptr(), Cm=D(N), Cn=D(1), Rm=D(0), Rn=D(0),
                          alpha=1.0, split_k=sk, accumulate=int(sk > 1), precision=_abi.PREC_BF16X3)
        assert L.eegclip_gemm_f32(g, st) == 0
        res[f"x3_sk{sk}_us"] = round(ev_us(lambda: L.eegclip_gemm_f32(g, st)), 2)
    print(f"{M}x{N}x{K}", res, flush=True)
