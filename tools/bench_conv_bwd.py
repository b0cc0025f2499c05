"""time csrc/conv.hip: conv_bwd_fused_kernel stand-alone at B = 256 against the three kernels it replaces (sconv_bwd_x_apply, tsconv_bwd_w, tsconv_bwd_x)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeg_image_decode_amd import _abi
from eeg_image_decode_amd._lib import lib

L = lib()
B, H, C, W = int(os.environ.get("CB_B", "256")), 63, 40, 36
st = torch.cuda.current_stream().cuda_stream
dev = "cuda"
y1 = torch.randn(B, C, H, W, device=dev) * 1.3 + 0.2
g1, b1 = 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
Ws = torch.randn(C, C, H, device=dev) / (C * H) ** 0.5
dy2 = torch.randn(B, C, W, device=dev)
w25 = torch.randn(40, 25, device=dev) * 0.2
x = torch.randn(B, 64, 250, device=dev)
mu, rs = y1.mean((0, 2, 3)).contiguous(), (1 / (y1.var((0, 2, 3), unbiased=False) + 1e-5).sqrt()).contiguous()
K = C * H
wh, wl = torch.zeros(K, 64, dtype=torch.bfloat16, device=dev), torch.zeros(K, 64, dtype=torch.bfloat16, device=dev)
it = (_abi.SplitItem * 1)(_abi.SplitItem(src=Ws.data_ptr(), hi=wh.data_ptr(), lo=wl.data_ptr(), rows=C, cols=K, ld_src=K, ld_out=64, transpose=1))
assert L.eegclip_split_rows(it, 1, st) == 0
sums = torch.zeros(80, dtype=torch.float64, device=dev)
assert L.eegclip_sconv_bwd_x_stats(dy2.data_ptr(), Ws.data_ptr(), wh.data_ptr(), wl.data_ptr(), y1.data_ptr(), mu.data_ptr(), rs.data_ptr(), g1.data_ptr(), b1.data_ptr(),
                                   sums.data_ptr(), None, B, H, st) == 0
count = float(B * H * W)
dy1 = torch.empty(B, C, H, W, device=dev)
dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dw, dx = torch.zeros(40, 25, device=dev), torch.zeros(B, 64, 250, device=dev)
wsw = torch.empty(int(L.eegclip_tsconv_bwd_w_workspace_floats(B, H)), device=dev)
wsf = torch.empty(int(L.eegclip_conv_bwd_fused_workspace_floats(B, H)), device=dev)


def ev(f, n=10):
    for _ in range(2):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / n * 1e3, 2)


def fused(cap=0):
    assert L.eegclip_conv_bwd_fused(dy2.data_ptr(), wh.data_ptr(), wl.data_ptr(), y1.data_ptr(), mu.data_ptr(), rs.data_ptr(), g1.data_ptr(), b1.data_ptr(), sums.data_ptr(),
                                    None, count, dg.data_ptr(), db.data_ptr(), x.data_ptr(), 64 * 250, 250, w25.data_ptr(), dx.data_ptr(), dw.data_ptr(), wsf.data_ptr(), B, H, cap,
                                    st) == 0


res = {}
only = sys.argv[2] if len(sys.argv) > 2 else None
if only != "fused":
    res["apply_us"] = ev(lambda: L.eegclip_sconv_bwd_x_apply(dy2.data_ptr(), Ws.data_ptr(), wh.data_ptr(), wl.data_ptr(), y1.data_ptr(), mu.data_ptr(), rs.data_ptr(), g1.data_ptr(),
                                                             b1.data_ptr(), sums.data_ptr(), None, count, dy1.data_ptr(), dg.data_ptr(), db.data_ptr(), B, H, st))
    res["tsconv_bwd_w_us"] = ev(lambda: L.eegclip_tsconv_bwd_w(x.data_ptr(), 64 * 250, 250, dy1.data_ptr(), dw.data_ptr(), wsw.data_ptr(), B, H, 250, 40, st))
    res["tsconv_bwd_x_us"] = ev(lambda: L.eegclip_tsconv_bwd_x(dy1.data_ptr(), w25.data_ptr(), dx.data_ptr(), 64 * 250, 250, B, H, 250, 40, st))
for cap in ([0] if only else [0, 256, 1024]):
    res[f"fused_cap{cap}_us"] = ev(lambda: fused(cap))
print(res)
json.dump(res, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/conv_bwd_bench.json", "w"), indent=1)
