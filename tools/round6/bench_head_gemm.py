"""the M = 256 GEMM shapes of the step on csrc/head_gemm.hip (planes, K-parallel slabs, B row-major and k-major) over slice counts, against
csrc/gemm_planes.hip (unsplit) and the fp32-operand split-K gemm_x3 launches of round 5.  Back-to-back launches of one kernel under an event bracket: what
a launch costs inside a dependent chain (the step) is its duration in the trace (profiles/r6_step_timeline.txt).
(The first version of the kernel also combined the slabs inside the launch -- release fence + ticket: 18 - 58 us; CHANGELOG.md.)"""
import json, os, sys
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


def planes(t):
    h, l = torch.empty(t.shape, dtype=torch.bfloat16, device="cuda"), torch.empty(t.shape, dtype=torch.bfloat16, device="cuda")
    assert L.eegclip_split_bf16(t.data_ptr(), h.data_ptr(), l.data_ptr(), t.numel(), st) == 0
    return h, l


st = torch.cuda.current_stream().cuda_stream
out = {}
# (M, N, K): head Linear 1, head Linear 2 / dX of Linear 2, dX of Linear 1, the logits of two targets, the query gradient
for M, N, K in ((256, 1024, 1440), (256, 1024, 1024), (256, 1440, 1024), (256, 512, 1024), (256, 1024, 512)):
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
    (ah, al), (bh, bl) = planes(a), planes(b)
    bt = b.t().contiguous()                                   # the same operand k-major: [K][N]
    bth, btl = planes(bt)
    ref = (a.double() @ b.double().T).float()
    res = {"auto_slices": int(L.eegclip_head_gemm_slices(M, N, K))}
    for s in (1, 2, 4, 8, 16):
        if s > max(1, K // 128):
            continue
        ws = torch.empty(s, M, N, device="cuda")
        for tag, (xh, xl, ldb, km) in (("", (bh, bl, K, 0)), ("_kmajor", (bth, btl, N, 1))):
            d = _abi.HeadGemmDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=xh.data_ptr(), b_lo=xl.data_ptr(), lda=K, ldb=ldb, M=M, N=N, K=K, slices=s,
                                  slab_stride=M * N, C=ws.data_ptr(), ldc=N, b_kmajor=km)
            assert L.eegclip_head_gemm(d, st) == 0
            torch.cuda.synchronize()
            assert float((ws.sum(0) - ref).abs().max()) < 2e-4, (M, N, K, s, tag)
            res[f"slabs{tag}_s{s}_us"] = round(ev_us(lambda: L.eegclip_head_gemm(d, st)), 2)
    c = torch.zeros(M, N, device="cuda")
    if N % 64 == 0:
        d = _abi.GemmPlanesDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), lda=K, ldb=K, M=M, N=N, K=K, C=c.data_ptr(), ldc=N)
        assert L.eegclip_gemm_planes(d, st) == 0
        res["gemm_planes_unsplit_us"] = round(ev_us(lambda: L.eegclip_gemm_planes(d, st)), 2)
    for sk in (1, 8):
        g = _abi.GemmDesc(M=M, N=N, K=K, A=a.data_ptr(), Am=D(K), Ak=D(1), B=b.data_ptr(), Bk=D(1), Bn=D(K), C=c.data_ptr(), Cm=D(N), Cn=D(1), Rm=D(0), Rn=D(0),
                          alpha=1.0, split_k=sk, accumulate=int(sk > 1), precision=_abi.PREC_BF16X3)
        assert L.eegclip_gemm_f32(g, st) == 0
        res[f"gemm_x3_sk{sk}_us"] = round(ev_us(lambda: L.eegclip_gemm_f32(g, st)), 2)
    print(f"{M}x{N}x{K}", res, flush=True)
    out[f"{M}x{N}x{K}"] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r6_head_gemm_bench.json", "w"), indent=1)
