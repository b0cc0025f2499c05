"""Split-bf16 GEMM (csrc/gemm_x3.hip) against the exact-f32 kernel on the shapes of one training step at B = 256: every tile
configuration, interleaved rounds in one process (guide 5.4 rule 24), median HIP-event time per launch.

    python tools/bench_gemm_x3.py [--rounds 7] [--reps 10] [--out gpurun_out/x3_sweep.json]
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib

D = _abi.dim

# (name, M, N, K, kind, split_k): nt = X W^T (forward Linear), nn = dY W (input gradient), tn = dY^T X (weight gradient, split-K)
SHAPES = [
    ("embed 16128x250x250", 16128, 250, 250, "nt", 1), ("qkv 16384x744x250", 16384, 744, 250, "nt", 1),
    ("attn_out 16384x250x248", 16384, 250, 248, "nt", 1), ("ffn1 16384x256x250", 16384, 256, 250, "nt", 1),
    ("ffn2 16384x250x256", 16384, 250, 256, "nt", 1), ("head0 256x1024x1440/sk8", 256, 1024, 1440, "nt", 8),
    ("head1 256x1024x1024/sk8", 256, 1024, 1024, "nt", 8),
    ("d_qkv 16384x250x744", 16384, 250, 744, "nn", 1), ("d_ffn2 16384x256x250", 16384, 256, 250, "nn", 1),
    ("d_ffn1 16384x250x256", 16384, 250, 256, "nn", 1), ("d_out 16384x248x250", 16384, 248, 250, "nn", 1),
    ("d_head1 256x1024x1024/sk8", 256, 1024, 1024, "nn", 8), ("d_head0 256x1440x1024/sk8", 256, 1440, 1024, "nn", 8),
    ("w_qkv 744x250x16384/sk32", 744, 250, 16384, "tn", 32), ("w_ffn1 256x250x16384/sk32", 256, 250, 16384, "tn", 32),
    ("w_ffn2 250x256x16384/sk32", 250, 256, 16384, "tn", 32), ("w_out 250x248x16384/sk32", 250, 248, 16384, "tn", 32),
    ("w_head0 1024x1440x256", 1024, 1440, 256, "tn", 1), ("w_head1 1024x1024x256", 1024, 1024, 256, "tn", 1),
    ("prior 1024x1024x1024", 1024, 1024, 1024, "nt", 1), ("sq 4096^3", 4096, 4096, 4096, "nt", 1),
]
CFGS = [("f32", 0), ("x3 auto", 1), ("x3 planes", "planes")] + [(f"x3 cfg{c}", 1 | ((c + 1) << 8)) for c in range(6)]


def make(M, N, K, kind, split):
    if kind == "nt":
        A = torch.randn(M, K, device="cuda"); B = torch.randn(N, K, device="cuda"); Am, Ak, Bk, Bn = D(K), D(1), D(1), D(K)
    elif kind == "nn":
        A = torch.randn(M, K, device="cuda"); B = torch.randn(K, N, device="cuda"); Am, Ak, Bk, Bn = D(K), D(1), D(N), D(1)
    else:
        A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); Am, Ak, Bk, Bn = D(1), D(M), D(N), D(1)
    C = torch.zeros(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    d = _abi.GemmDesc(M=M, N=N, K=K, A=A.data_ptr(), Am=Am, Ak=Ak, B=B.data_ptr(), Bk=Bk, Bn=Bn, C=C.data_ptr(), Cm=D(N), Cn=D(1), Cpre=None,
                      bias_n=bias.data_ptr(), bias_m=None, R=None, Rm=D(0), Rn=D(0), alpha=1.0, accumulate=int(split > 1), act=0, drop_p=0.0, seed=1,
                      drop_site=0, split_k=split)
    planes = None
    if kind in ("nt", "nn"):                                      # weights pre-split into bf16 planes (rows = output columns)
        Wm = B if kind == "nt" else B.t().contiguous()            # (N, K)
        ld = (K + 63) // 64 * 64
        hi = torch.zeros(N, ld, dtype=torch.bfloat16, device="cuda"); lo = torch.zeros(N, ld, dtype=torch.bfloat16, device="cuda")
        it = (_abi.SplitItem * 1)(_abi.SplitItem(src=Wm.data_ptr(), hi=hi.data_ptr(), lo=lo.data_ptr(), rows=N, cols=K, ld_src=K, ld_out=ld, transpose=0))
        assert lib().eegclip_split_rows(it, 1, torch.cuda.current_stream().cuda_stream) == 0
        planes = (hi, lo, ld, Wm)
    return d, (A, B, C, bias, planes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    L = lib()
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for name, M, N, K, kind, split in SHAPES:
        d, keep = make(M, N, K, kind, split)
        reps = max(2, args.reps // 4) if M * N * K > 1e10 else args.reps
        times = {c: [] for c, _ in CFGS if _ != "planes" or keep[4] is not None}
        planes = keep[4]

        def setp(p):
            if p == "planes":
                d.precision = 1
                d.B_hi, d.B_lo, d.ldb_planes = planes[0].data_ptr(), planes[1].data_ptr(), planes[2]
            else:
                d.precision = p
                d.B_hi, d.B_lo, d.ldb_planes = None, None, 0
        cfgs = [c for c in CFGS if c[1] != "planes" or planes is not None]
        for c, p in cfgs:                                             # warm-up: code objects, clocks
            setp(p)
            for _ in range(2):
                assert L.eegclip_gemm_f32(ctypes.byref(d), st) == 0
        for _ in range(args.rounds):
            for c, p in cfgs:
                setp(p)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    L.eegclip_gemm_f32(ctypes.byref(d), st)
                e1.record()
                torch.cuda.synchronize()
                times[c].append(e0.elapsed_time(e1) / reps * 1e3)
        row = {c: round(float(np.median(v)), 2) for c, v in times.items()}
        out[name] = row
        best = min((v, c) for c, v in row.items() if c != "f32")
        print(f"{name:28s} " + "  ".join(f"{c}={v:7.1f}" for c, v in row.items()) + f"   | best {best[1]} {row['f32'] / best[0]:.2f}x  "
              f"{2.0 * M * N * K / best[0] / 1e6:.0f} TF", flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump({"unit": "us per launch (median of rounds)", "shapes": out}, f, indent=1)


if __name__ == "__main__":
    main()
