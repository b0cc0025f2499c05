"""time the planes weight-gradient path (csrc/wgrad_planes.hip) against the plan GEMM on the transformer block's shapes (K = 16384 token rows)"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib

L = lib()
D = _abi.dim


def ev(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


def main():
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for (M, N, K) in ((250, 256, 16384), (256, 250, 16384), (250, 248, 16384), (744, 250, 16384)):
        dy, x = torch.randn(K, M, device="cuda"), torch.randn(K, N, device="cuda")
        pad = lambda v, m: (v + m - 1) // m * m
        for ldp in (K, K + 64):
          at = torch.empty(2, pad(M, 128), ldp, dtype=torch.bfloat16, device="cuda")
          bt = torch.empty(2, pad(N, 64), ldp, dtype=torch.bfloat16, device="cuda")
          t_sa = ev(lambda: L.eegclip_split_transpose(dy.data_ptr(), M, K, M, at.shape[1], at[0].data_ptr(), at[1].data_ptr(), ldp, st))
          wsx = torch.empty(int(L.eegclip_wgrad_planes_workspace_floats(M, N, K)), device="cuda")
          cx = torch.zeros(M, N, device="cuda")
          t_gx = ev(lambda: L.eegclip_wgrad_planes(at[0].data_ptr(), at[1].data_ptr(), bt[0].data_ptr(), bt[1].data_ptr(), ldp, M, N, K, cx.data_ptr(), N, None, wsx.data_ptr(), st))
          out.setdefault(f"{M}x{N}x{K}", {})[f"ld={'K' if ldp == K else 'K+64'}"] = {"split_A_us": round(t_sa, 1), "planes_gemm_no_bias_us": round(t_gx, 1)}
        ldp = K + 64
        ws = torch.empty(int(L.eegclip_wgrad_planes_workspace_floats(M, N, K)), device="cuda")
        c, bias = torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")
        t_sa = ev(lambda: L.eegclip_split_transpose(dy.data_ptr(), M, K, M, at.shape[1], at[0].data_ptr(), at[1].data_ptr(), ldp, st))
        t_sb = ev(lambda: L.eegclip_split_transpose(x.data_ptr(), N, K, N, bt.shape[1], bt[0].data_ptr(), bt[1].data_ptr(), ldp, st))
        t_g = ev(lambda: L.eegclip_wgrad_planes(at[0].data_ptr(), at[1].data_ptr(), bt[0].data_ptr(), bt[1].data_ptr(), ldp, M, N, K, c.data_ptr(), N, bias.data_ptr(),
                                                ws.data_ptr(), st))
        c.zero_()
        L.eegclip_wgrad_planes(at[0].data_ptr(), at[1].data_ptr(), bt[0].data_ptr(), bt[1].data_ptr(), ldp, M, N, K, c.data_ptr(), N, bias.data_ptr(), ws.data_ptr(), st)
        ref = dy.double().T @ x.double()
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        d = _abi.GemmDesc(M=M, N=N, K=K, A=dy.data_ptr(), Am=D(1), Ak=D(M), B=x.data_ptr(), Bk=D(N), Bn=D(1), C=c.data_ptr(), Cm=D(N), Cn=D(1), Cpre=None, bias_n=None,
                          bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=1, act=0, drop_p=0.0, seed=0, drop_site=0, split_k=32, rowsum_a=bias.data_ptr(),
                          precision=_abi.PREC_BF16X3)
        t_old = ev(lambda: L.eegclip_gemm_f32(ctypes.byref(d), st))
        out[f"{M}x{N}x{K}"]["final"] = {"split_A_us": round(t_sa, 1), "split_B_us": round(t_sb, 1), "planes_gemm_us": round(t_g, 1), "plan_gemm_x3_us": round(t_old, 1),
                               "planes_TFLOPs": round(2.0 * M * N * K / t_g / 1e6, 1), "rel_err": err}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
