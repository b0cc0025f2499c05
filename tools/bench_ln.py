"""LayerNorm kernels at the step's shape (16384 x 250): forward, dx, parameter gradients (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd._lib import lib
L = lib(); st = torch.cuda.current_stream().cuda_stream
R, C = 16384, 250
x = torch.randn(R, C, device="cuda"); dy = torch.randn(R, C, device="cuda"); g = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
y = torch.empty_like(x); mu = torch.empty(R, device="cuda"); rs = torch.empty(R, device="cuda"); dx = torch.empty_like(x); dxd = torch.empty_like(x)
dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
P = lambda a: a.data_ptr()
print("fwd        %.1f us" % t(lambda: L.eegclip_layernorm_fwd(P(x), P(g), P(b), P(y), P(mu), P(rs), R, C, 1e-5, st)))
print("bwd dx     %.1f us" % t(lambda: L.eegclip_layernorm_bwd(P(dy), P(x), P(g), P(mu), P(rs), P(dx), None, None, R, C, 0, None, 0.0, 0, 0, st)))
print("bwd dx+drop %.1f us" % t(lambda: L.eegclip_layernorm_bwd(P(dy), P(x), P(g), P(mu), P(rs), P(dx), None, None, R, C, 0, P(dxd), 0.25, 1, 2, st)))
print("bwd param  %.1f us" % t(lambda: L.eegclip_layernorm_bwd(P(dy), P(x), None, P(mu), P(rs), None, P(dg), P(db), R, C, 0, None, 0.0, 0, 0, st)))
ws = torch.empty(int(L.eegclip_layernorm_bwd_params_workspace_floats(R, C)), device="cuda")
print("bwd param (partials + reduce) %.1f us" % t(lambda: L.eegclip_layernorm_bwd_params(P(dy), P(x), P(mu), P(rs), P(dg), P(db), R, C, P(ws), st)))
