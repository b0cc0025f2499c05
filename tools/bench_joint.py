"""SURVEY 8f row 1 alone: the joint-subject training step (bench.py's secondary line bench_joint)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench._sec_joint()))
