"""one GEMM shape, N launches -- target of rocprofv3 --pmc runs:  python tools/x3_one.py M N K kind split precision reps"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_gemm_x3 as b
from eeg_image_decode_amd._lib import lib
M, N, K, kind, split, prec, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
d, keep = b.make(M, N, K, kind, split)
d.precision = prec
st = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    assert lib().eegclip_gemm_f32(ctypes.byref(d), st) == 0
torch.cuda.synchronize()
