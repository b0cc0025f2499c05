"""time the fused transformer-block forward (csrc/token_block.hip) alone: train mode (Philox dropout on) and evaluation mode, B = 256.
   python tools/bench_token_block.py [iters]          (run it under rocprofv3 --pmc ... for counters)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eeg_image_decode_amd import synthetic as syn
from eeg_image_decode_amd.atms import ATMS


def main(iters=50, B=256):
    m = ATMS().cuda()
    x = torch.from_numpy(syn.eeg_batch(3, B)).cuda()
    out = {}
    for mode in ("train", "eval"):
        m.train(mode == "train")
        with torch.no_grad():
            for _ in range(3):
                m(x, 1)
            eng = m._engine()
            pl = eng.plans[next(k for k in eng.plans if k[0] == "f" and k[2] == (mode == "train"))]
            names = pl.op_names()
            idx = [i for i, n in enumerate(names) if n in ("eegclip_token_block_fwd", "eegclip_token_block_pack")]
            pl.use_c_executor = False
            pl.time_ops(idx)
            for _ in range(iters):
                m(x, 1)
            t = pl.timings_ms()
            pl.timed = {}
        out[mode] = {names[i]: round(1e3 * float(np.mean(v)), 2) for i, v in t.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
