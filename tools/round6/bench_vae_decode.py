"""round 6: the VAE decode section of bench.py alone (secondary.vae_decode_1024px), for A/B runs through tools/with_lib.py and kernel traces"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print(json.dumps(bench._sec_vae_decode()))
