"""one GEMM shape, N launches -- target of rocprofv3 --pmc runs:  python tools/bench_gemm_one.py M N K kind [split] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_gemm
M, N, K, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
split = int(sys.argv[5]) if len(sys.argv) > 5 else 1
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
bench_gemm.run(M, N, K, kind, split, reps)
