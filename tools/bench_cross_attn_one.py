"""one SDXL cross-attention shape, N launches -- target of rocprofv3 --pmc runs:  python tools/bench_cross_attn_one.py [HW heads images reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg_image_decode_amd.sdxl import cross_attention
HW, heads, images, reps = (int(x) for x in (sys.argv[1:] + ["4096", "10", "8", "10"])[:4])
B, C = 2 * images, heads * 64
q = torch.randn(B, HW, C, device="cuda", dtype=torch.float16)
k, v = torch.randn(B, 77, C, device="cuda", dtype=torch.float16), torch.randn(B, 77, C, device="cuda", dtype=torch.float16)
ki, vi = torch.randn(B, 4, C, device="cuda", dtype=torch.float16), torch.randn(B, 4, C, device="cuda", dtype=torch.float16)
for _ in range(reps):
    cross_attention(q, k, v, heads, ki, vi, 1.0)
torch.cuda.synchronize()
