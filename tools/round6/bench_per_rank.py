"""round 6: the row-sharded loss section of bench.py alone (secondary.infonce_per_rank_block)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
r = bench._sec_infonce_per_rank()
print(json.dumps({m: r[m] for m in ("parity_mode", "throughput_mode")}))
