"""split-K GEMMs of the step (weight gradients dW = dY^T X over 16384 token rows, the M = 256 head GEMMs): time against split_k, atomics vs
the workspace reduction (eegclip_gemm_desc.workspace)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import bench_gemm_x3 as b
from eeg_image_decode_amd._lib import lib

L = lib()
st = torch.cuda.current_stream().cuda_stream


def t_of(d):
    for _ in range(3):
        assert L.eegclip_gemm_f32(ctypes.byref(d), st) == 0
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.eegclip_gemm_f32(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    return float(np.median(ts))


for name, M, N, K, kind, sks in [("w_ffn1", 256, 250, 16384, "tn", (8, 16, 32, 64, 128)), ("w_qkv", 744, 250, 16384, "tn", (8, 16, 32, 64, 128)),
                                 ("head0", 256, 1024, 1440, "nt", (1, 2, 4, 8, 16)), ("d_head0", 256, 1440, 1024, "nn", (1, 2, 4, 8, 16)),
                                 ("w_head0", 1024, 1440, 256, "tn", (1, 2, 4))]:
    for cfgname, prec in (("cfg2", 1), ("cfg0", 1 | (1 << 8)), ("cfg3", 1 | (4 << 8))):
        row = []
        for sk in sks:
            d, keep = b.make(M, N, K, kind, sk)
            d.precision = prec
            ta = t_of(d)
            need = L.eegclip_gemm_workspace_bytes(ctypes.byref(d))
            tw = float("nan")
            if need > 0 and os.environ.get("PROBE_WS"):
                ws = torch.zeros(need // 4, device="cuda")
                d.workspace, d.workspace_bytes = ws.data_ptr(), need
                tw = t_of(d)
            row.append(f"sk{sk}: {ta:5.1f}/{tw:5.1f}")
        print(f"{name:8s} {cfgname}  (atomics/workspace us)  " + "   ".join(row), flush=True)
