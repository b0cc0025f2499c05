"""round 6: the SDXL stand-in sampling loop section of bench.py alone (secondary.sdxl_sampling_loop), for A/B runs through tools/with_lib.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
r = bench._sec_sdxl_loop()
print(json.dumps({k: r[k] for k in ("ms_per_step", "attention_stack_TFLOPs", "images_per_s")}))
