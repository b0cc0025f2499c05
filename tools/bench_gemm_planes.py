"""csrc/gemm_planes.hip against the general split-bf16 GEMM (csrc/gemm_x3.hip, fp32 operands) on the diffusion prior's Linear shapes at batch 1024:
HIP-event timed back to back.   python tools/bench_gemm_planes.py"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib
from eeg_image_decode_amd.plan import D

L = lib()


def ev_us(fn, reps=100, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for M, N, K in ((1024, 1024, 1024), (1024, 2880, 1024), (1024, 2880, 512), (1024, 512, 1024), (1024, 512, 512), (1024, 256, 512), (1024, 256, 256),
                    (1024, 128, 256), (1024, 128, 128), (1024, 64, 128), (1024, 64, 64)):
        a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") / K ** 0.5
        ah, al, bh, bl = (torch.empty(t.shape, dtype=torch.bfloat16, device="cuda") for t in (a, a, b, b))
        L.eegclip_split_bf16(a.data_ptr(), ah.data_ptr(), al.data_ptr(), a.numel(), st)
        L.eegclip_split_bf16(b.data_ptr(), bh.data_ptr(), bl.data_ptr(), b.numel(), st)
        c = torch.empty(M, N, device="cuda")
        bias = torch.randn(N, device="cuda")
        d = _abi.GemmPlanesDesc(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), lda=K, ldb=K, M=M, N=N, K=K, C=c.data_ptr(), ldc=N,
                                bias=bias.data_ptr())
        assert L.eegclip_gemm_planes(d, st) == 0
        g = _abi.GemmDesc(M=M, N=N, K=K, A=a.data_ptr(), Am=D(K), Ak=D(1), B=b.data_ptr(), Bk=D(1), Bn=D(K), C=c.data_ptr(), Cm=D(N), Cn=D(1), bias_n=bias.data_ptr(),
                          Rm=D(0), Rn=D(0), alpha=1.0, split_k=1, precision=_abi.PREC_BF16X3)
        ref = torch.empty(M, N, device="cuda")
        g.C = ref.data_ptr()
        assert L.eegclip_gemm_f32(g, st) == 0
        torch.cuda.synchronize()
        err = float((c - ref).abs().max())
        us_p = ev_us(lambda: L.eegclip_gemm_planes(d, st))
        us_x = ev_us(lambda: L.eegclip_gemm_f32(g, st))
        flop = 2.0 * M * N * K
        out[f"{M}x{N}x{K}"] = {"planes_us": round(us_p, 2), "gemm_x3_us": round(us_x, 2), "planes_TF": round(flop / us_p / 1e6, 1), "max_abs_diff": err}
        print(f"{M}x{N}x{K}", out[f"{M}x{N}x{K}"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__" and "--tn" not in sys.argv:
    main()


def main_tn():
    """the weight-gradient form (eegclip_wgrad_planes) against eegclip_gemm_f32 with both operands k-strided, rows = 1024"""
    st = torch.cuda.current_stream().cuda_stream
    rows = 1024
    for M, N, slices in ((1024, 1024, 1), (1024, 1024, 2), (1024, 1024, 4), (2880, 512, 1), (2880, 512, 2), (2880, 1024, 1), (512, 1024, 2), (512, 1024, 4), (512, 512, 4), (512, 512, 8),
                         (256, 512, 8), (256, 256, 8), (128, 256, 8), (128, 128, 8), (64, 128, 8), (64, 64, 8)):
        pad = lambda c: (c + 127) // 128 * 128
        dy, x = torch.randn(rows, pad(M), device="cuda"), torch.randn(rows, pad(N), device="cuda")
        ah, al, bh, bl = (torch.empty(t.shape, dtype=torch.bfloat16, device="cuda") for t in (dy, dy, x, x))
        L.eegclip_split_bf16(dy.data_ptr(), ah.data_ptr(), al.data_ptr(), dy.numel(), st)
        L.eegclip_split_bf16(x.data_ptr(), bh.data_ptr(), bl.data_ptr(), x.numel(), st)
        out, bias = torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")
        p = (_abi.WgradPlanesProblem * 1)(_abi.WgradPlanesProblem(a_hi=ah.data_ptr(), a_lo=al.data_ptr(), lda=pad(M), b_hi=bh.data_ptr(), b_lo=bl.data_ptr(), ldb=pad(N),
                                                                 rows=rows, M=M, N=N, out=out.data_ptr(), ldo=N, bias_out=bias.data_ptr(), slices=slices))
        assert L.eegclip_wgrad_planes(p, 1, st) == 0
        ref = torch.zeros(M, N, device="cuda")
        g = _abi.GemmDesc(M=M, N=N, K=rows, A=dy.data_ptr(), Am=D(1), Ak=D(pad(M)), B=x.data_ptr(), Bk=D(pad(N)), Bn=D(1), C=ref.data_ptr(), Cm=D(N), Cn=D(1),
                          Rm=D(0), Rn=D(0), alpha=1.0, accumulate=1, split_k=1, precision=_abi.PREC_BF16X3)
        assert L.eegclip_gemm_f32(g, st) == 0
        torch.cuda.synchronize()
        err = float((out - ref).abs().max())
        us_p = ev_us(lambda: L.eegclip_wgrad_planes(p, 1, st))
        us_x = ev_us(lambda: L.eegclip_gemm_f32(g, st))
        print(f"TN {M}x{N} rows {rows} slices {slices}: planes {us_p:.2f} us, gemm_x3 {us_x:.2f} us, {2.0 * M * N * rows / us_p / 1e6:.1f} TF, max diff {err:.2e}", flush=True)


if __name__ == "__main__" and "--tn" in sys.argv:
    main_tn()
