"""micro-benchmark of the three tsconv kernels at B = 256 (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd._lib import lib
L = lib(); st = torch.cuda.current_stream().cuda_stream
B, H = 256, 63
x = torch.randn(B, 64, 250, device="cuda"); w25 = torch.randn(40, 25, device="cuda") * 0.1; bias = torch.randn(40, device="cuda")
y = torch.empty(B, 40, H, 36, device="cuda"); dy = torch.randn_like(y); dx = torch.zeros(B, 64, 250, device="cuda")
dw25 = torch.zeros(40, 25, device="cuda"); ws = torch.empty(int(L.eegclip_tsconv_bwd_w_workspace_floats(B, H)), device="cuda")
sums = torch.zeros(80, dtype=torch.float64, device="cuda")
def t(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
print("fwd   %.1f us" % t(lambda: L.eegclip_tsconv_fwd(x.data_ptr(), 16000, 250, w25.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, 250, 40, sums.data_ptr(), None, st)))
print("bwd_w %.1f us" % t(lambda: L.eegclip_tsconv_bwd_w(x.data_ptr(), 16000, 250, dy.data_ptr(), dw25.data_ptr(), ws.data_ptr(), B, H, 250, 40, st)))
print("bwd_x %.1f us" % t(lambda: L.eegclip_tsconv_bwd_x(dy.data_ptr(), w25.data_ptr(), dx.data_ptr(), 16000, 250, B, H, 250, 40, st)))
