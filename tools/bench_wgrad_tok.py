"""time csrc/wgrad_tok.hip stand-alone on the backward plan's three launches (B = 256): main kernel and slab reduction, per variant and slice count"""
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib

L = lib()
B = 256
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16


def planes(groups=1):
    return (torch.randn(groups, B, 2, 64, 256, device="cuda") * 0.5).to(bf)


def ev(f, n=20):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


LAUNCHES = {"ffn_out": [(250, 256, 1, 0, 0, 1), (256, 250, 1, 0, 0, 0), (250, 248, 1, 0, 1, 0)], "qkv": [(744, 250, 3, 1, 0, 0)], "embed": [(250, 250, 1, 0, 0, 0)]}
res = {}
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None          # e.g. qkv/16/0: one launch shape, slice count and variant (PMC runs)
for name, probs in LAUNCHES.items():
    if ONLY and name != ONLY[0]:
        continue
    arr = (_abi.WgradTokProblem * len(probs))()
    keep = []
    for i, (M, N, mg, hm, hn, bm) in enumerate(probs):
        a, b = planes(mg), planes()
        out, bias = torch.zeros(M, N, device="cuda"), torch.zeros(M, device="cuda")
        keep += [a, b, out, bias]
        arr[i] = _abi.WgradTokProblem(a=a.data_ptr(), b=b.data_ptr(), a_group_stride=B * 65536 if mg > 1 else 0, m_groups=mg, heads_m=hm, heads_n=hn, M=M, N=N,
                                      out=out.data_ptr(), ldo=N, bias_out=bias.data_ptr(), bias_mfma=bm)
    groups = sum(p[2] for p in probs)
    flops = sum(2.0 * p[0] * p[1] * 64 * B for p in probs)
    for slices in sorted({8, 16, 20, 21, 24, 32, 40, 42, 64, int(L.eegclip_wgrad_tok_slices(groups, B))}):
        if 4 * groups * slices > 1024 or (ONLY and slices != int(ONLY[1])):
            continue
        ws = torch.empty(int(L.eegclip_wgrad_tok_workspace_floats(arr, len(probs), B, slices)), device="cuda")
        row = {}
        for v in ((int(ONLY[2]),) if ONLY else (0, 2, 3)):          # 512 threads | 256 threads | 8 MFMA + 4 producer waves
            t = ev(lambda: L.eegclip_wgrad_tok(arr, len(probs), B, slices, ws.data_ptr(), v, st))
            row[f"kernel_v{v}_us"] = round(t, 2)
        row["reduce_us"] = round(ev(lambda: L.eegclip_wgrad_tok_reduce(arr, len(probs), B, slices, ws.data_ptr(), st)), 2)
        best = min(val for k_, val in row.items() if k_.startswith("kernel_"))
        row["TFLOPs_best_kernel_plus_reduce"] = round(flops / (best + row["reduce_us"]) * 1e-6, 1)
        res[f"{name}/s{slices}"] = row
        print(name, slices, row, flush=True)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wgrad_tok_bench.json", "w"), indent=1)
