"""configs[3] alone: the diffusion-prior training step at batch 1024 (bench.py's secondary line) -- run under rocprofv3 --kernel-trace for the launch list"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench._sec_prior_train()))
