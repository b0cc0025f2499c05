"""InfoNCE logits GEMM at the large-batch configuration (BASELINE.json north_star: global batch 2048, D = 1024): bf16 MFMA kernel vs
the fp32-exact GEMM, HIP-event timed; prints one JSON line (fraction of the dense bf16 MFMA roofline, 2.5 PFLOP/s)."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib

D = _abi.dim
L = lib()
st = torch.cuda.current_stream().cuda_stream
PEAK_BF16_TF, PEAK_F32_TF = 2500.0, 157.3


def timed(fn, reps):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main(N=2048, Dm=1024, reps=200):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.nn.functional.layer_norm(torch.randn(N, Dm, device="cuda", generator=g), (Dm,))        # EEG embeddings leave a LayerNorm
    b = torch.nn.functional.normalize(torch.randn(N, Dm, device="cuda", generator=g), dim=1)          # CLIP image embeddings are unit norm
    sc = torch.tensor([2.6593], device="cuda")
    a16 = torch.empty(N, Dm, dtype=torch.bfloat16, device="cuda")
    b16 = torch.empty(N, Dm, dtype=torch.bfloat16, device="cuda")
    c16 = torch.empty(N, N, device="cuda")
    c32 = torch.empty(N, N, device="cuda")
    assert L.eegclip_cast_bf16(a.data_ptr(), a16.data_ptr(), N * Dm, st) == 0
    assert L.eegclip_cast_bf16(b.data_ptr(), b16.data_ptr(), N * Dm, st) == 0
    run16 = lambda: L.eegclip_logits_bf16(a16.data_ptr(), b16.data_ptr(), c16.data_ptr(), N, N, Dm, N, sc.data_ptr(), st)
    d = _abi.GemmDesc(M=N, N=N, K=Dm, A=a.data_ptr(), Am=D(Dm), Ak=D(1), B=b.data_ptr(), Bk=D(1), Bn=D(Dm), C=c32.data_ptr(), Cm=D(N), Cn=D(1),
                      Cpre=None, bias_n=None, bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=2.6593, accumulate=0, act=0, drop_p=0.0, seed=0,
                      drop_site=0, split_k=1, rowsum_a=None)
    run32 = lambda: L.eegclip_gemm_f32(ctypes.byref(d), st)
    assert run16() == 0 and run32() == 0
    torch.cuda.synchronize()
    ref = 2.6593 * a.double() @ b.double().T
    e16, e32 = float((c16.double() - ref).abs().max()), float((c32.double() - ref).abs().max())
    ms16, ms32, msc = timed(run16, reps), timed(run32, reps // 4), timed(lambda: L.eegclip_cast_bf16(a.data_ptr(), a16.data_ptr(), N * Dm, st), reps)
    flop = 2.0 * N * N * Dm
    print(json.dumps({"workload": f"InfoNCE logits {N}x{N}x{Dm} (configs[2]: global batch {N})", "bf16_kernel_us": round(ms16 * 1e3, 2),
                      "bf16_TFLOPs": round(flop / ms16 / 1e9, 1), "frac_of_bf16_mfma_peak": round(flop / ms16 / 1e9 / PEAK_BF16_TF, 4),
                      "cast_one_operand_us": round(msc * 1e3, 2), "f32_kernel_us": round(ms32 * 1e3, 2), "f32_TFLOPs": round(flop / ms32 / 1e9, 1),
                      "frac_of_f32_mfma_peak": round(flop / ms32 / 1e9 / PEAK_F32_TF, 4), "max_abs_logit_error_bf16": round(e16, 5),
                      "max_abs_logit_error_f32": round(e32, 7), "max_abs_logit": round(float(ref.abs().max()), 2)}))


if __name__ == "__main__":
    main(*(int(x) for x in sys.argv[1:]))
