"""the projection head's GEMM shapes at B = 256 on csrc/head_gemm.hip (planes, K-parallel slabs + tickets) over slice counts, against csrc/gemm_planes.hip
(unsplit) and today's split-K gemm_x3 launches.  Back-to-back launches of one kernel: what a launch costs inside a dependent chain (the step) is its
duration in the trace, so the same loop is also run through rocprofv3 by tools/round6/gpu_a.sh."""
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
out = {}
for M, N, K in ((256, 1024, 1440), (256, 1024, 1024), (256, 1440, 1024), (256, 1024, 512)):
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
    ah, al, bh, bl = (torch.empty(t.shape, dtype=torch.bfloat16, device="cuda") for t in (a, a, b, b))
    L.eegclip_split_bf16(a.data_ptr(), ah.data_ptr(), al.data_ptr(), a.numel(), st)
    L.eegclip_split_bf16(b.data_ptr(), bh.data_ptr(), bl.data_ptr(), b.numel(), st)
    c = torch.zeros(M, N, device="cuda")
    cpre = torch.zeros(M, N, device="cuda")
    ph, plo = torch.empty(M, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    bias = torch.randn(N, device="cuda")
    res = {}
    ref = (a.double() @ b.double().T + bias.double()).float()
    for s in (1, 2, 3, 4, 6, 8, 12, 16):
        if s > K // 32:
            continue
        ws = torch.empty(max(4, int(L.eegclip_head_gemm_workspace_floats(M, N, s))), device="cuda")
        tk = torch.zeros(((M + 63) // 64) * ((N + 63) // 64), dtype=torch.int32, device="cuda")
        d = _abi.HeadGemmDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), lda=K, ldb=K, M=M, N=N, K=K, slices=s,
                              slabs=ws.data_ptr() if s > 1 else None, tickets=tk.data_ptr() if s > 1 else None, bias=bias.data_ptr(), Cpre=cpre.data_ptr(), ldcpre=N,
                              act=_abi.ACT_GELU, C=c.data_ptr(), ldc=N, p_hi=ph.data_ptr(), p_lo=plo.data_ptr(), ldp=N)
        assert L.eegclip_head_gemm(d, st) == 0
        torch.cuda.synchronize()
        err = float((cpre - ref).abs().max())
        assert err < 2e-4, (M, N, K, s, err)
        res[f"head_s{s}_us"] = round(ev_us(lambda: L.eegclip_head_gemm(d, st)), 2)
        assert int(tk.abs().sum()) == 0
        assert float((cpre - ref).abs().max()) < 2e-4                      # after 210 calls through the same counters
    res["auto_slices"] = int(L.eegclip_head_gemm_slices(M, N, K))
    if N % 64 == 0:
        d = _abi.GemmPlanesDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), lda=K, ldb=K, M=M, N=N, K=K, C=c.data_ptr(), ldc=N,
                                bias=bias.data_ptr(), p_hi=ph.data_ptr(), p_lo=plo.data_ptr(), ldp=N, planes_of=1)
        assert L.eegclip_gemm_planes(d, st) == 0
        res["planes_us"] = round(ev_us(lambda: L.eegclip_gemm_planes(d, st)), 2)
    for sk in (1, 8):
        g = _abi.GemmDesc(M=M, N=N, K=K, A=a.data_ptr(), Am=D(K), Ak=D(1), B=b.data_ptr(), Bk=D(1), Bn=D(K), C=c.data_ptr(), Cm=D(N), Cn=D(1), Rm=D(0), Rn=D(0),
                          alpha=1.0, split_k=sk, accumulate=int(sk > 1), precision=_abi.PREC_BF16X3)
        assert L.eegclip_gemm_f32(g, st) == 0
        res[f"x3_sk{sk}_us"] = round(ev_us(lambda: L.eegclip_gemm_f32(g, st)), 2)
    print(f"{M}x{N}x{K}", res, flush=True)
    out[f"{M}x{N}x{K}"] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r6_head_gemm_bench.json", "w"), indent=1)

# EXPERIMENT: slabs without the in-launch combine + an elementwise consumer (torch sum as a stand-in for the consumer's extra reads)
for M, N, K in ((256, 1024, 1440), (256, 1024, 1024), (256, 1440, 1024), (256, 1024, 512)):
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
    ah, al, bh, bl = (torch.empty(t.shape, dtype=torch.bfloat16, device="cuda") for t in (a, a, b, b))
    L.eegclip_split_bf16(a.data_ptr(), ah.data_ptr(), al.data_ptr(), a.numel(), st)
    L.eegclip_split_bf16(b.data_ptr(), bh.data_ptr(), bl.data_ptr(), b.numel(), st)
    ref = (a.double() @ b.double().T).float()
    res = {}
    for s in (2, 3, 4, 6, 8):
        ws = torch.empty(s, M, N, device="cuda")
        d = _abi.HeadGemmDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), lda=K, ldb=K, M=M, N=N, K=K, slices=s,
                              slabs=ws.data_ptr(), tickets=None, C=ws.data_ptr(), ldc=N)
        assert L.eegclip_head_gemm(d, st) == 0
        torch.cuda.synchronize()
        assert float((ws.sum(0) - ref).abs().max()) < 2e-4
        res[f"slabs_s{s}_us"] = round(ev_us(lambda: L.eegclip_head_gemm(d, st)), 2)
    print(f"{M}x{N}x{K} slabs-only", res, flush=True)
