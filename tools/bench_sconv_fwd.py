"""sconv_fwd at B = 256 with pieces switched off (EEGCLIP_SCF_DEBUG bits; timing ablation, results are wrong with any bit set)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd._lib import lib
L = lib(); st = torch.cuda.current_stream().cuda_stream
B, H, C, W = 256, 63, 40, 36
y1 = torch.randn(B, C, H, W, device="cuda"); Ws = torch.randn(C, C, H, device="cuda") * 0.02; bs = torch.randn(C, device="cuda")
mu, rs, g, be = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
y2 = torch.zeros(B, C, W, device="cuda"); sums = torch.zeros(80, dtype=torch.float64, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
names = {0: "full", 1: "no MFMA", 2: "no ELU", 32: "no LDS stores", 12: "no loads at all", 4 | 8 | 32 | 2: "MFMA + LDS reads only"}
ws = torch.empty(int(L.eegclip_sconv_fwd_workspace_floats(B)), device="cuda")
for use_ws in (False, True):
    for bits, name in names.items():
        os.environ["EEGCLIP_SCF_DEBUG"] = str(bits)
        us = t(lambda: L.eegclip_sconv_fwd(y1.data_ptr(), mu.data_ptr(), rs.data_ptr(), g.data_ptr(), be.data_ptr(), Ws.data_ptr(), None, None, 0, bs.data_ptr(),
                                             y2.data_ptr(), sums.data_ptr(), B, H, 1, ws.data_ptr() if use_ws else None, st))
        print(f"{'slabs ' if use_ws else 'atomics'} {bits:3d} {name:28s} {us:7.1f} us  (incl. the statistics kernel)", flush=True)
        if use_ws and bits == 0:
            break
