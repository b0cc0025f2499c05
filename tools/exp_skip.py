"""Diagnosis only: the bench step with one family of plan ops left out, to see what that family costs on the step's critical path
(kernels of the two plan streams overlap, so a family's summed kernel time is not what removing it would save).  The results of such a
run are wrong by construction; only ms_per_step is read.      usage: python tools/exp_skip.py <family>[,<family>...] [bench args]
families: wgrad (side-stream weight-gradient GEMMs), convbwd, attnbwd, tb_fwd, tb_bwd, none"""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eeg_image_decode_amd import plan as _plan          # noqa: E402

fams = set(sys.argv[1].split(","))
sys.argv = ["bench.py"] + sys.argv[2:]
CALLS = {"convbwd": ("eegclip_sconv_bwd_w", "eegclip_sconv_bwd_x_stats", "eegclip_sconv_bwd_x_apply", "eegclip_tsconv_bwd_w", "eegclip_tsconv_bwd_x"),
         "convfwd": ("eegclip_sconv_fwd", "eegclip_tsconv_fwd"),
         "attnbwd": ("eegclip_attention_bwd", "eegclip_attention_bwd_x3"), "tb_fwd": ("eegclip_token_block_fwd",), "tb_bwd": ("eegclip_token_block_bwd",)}
skip_calls = {n for f in fams for n in CALLS.get(f, ())}
_gemm, _call, _call_desc = _plan.Plan.gemm, _plan.Plan.call, _plan.Plan.call_desc


def gemm(self, *a, side=False, **k):
    if "wgrad" in fams and k.get("accumulate") == 1 and k.get("split_k", 1) > 1 and side:
        return self.desc(*a, **k)
    return _gemm(self, *a, side=side, **k)


def call(self, fname, *args, **k):
    if fname in skip_calls:
        return None
    return _call(self, fname, *args, **k)


def call_desc(self, fname, d, *a, **k):
    if fname in skip_calls:
        return d
    return _call_desc(self, fname, d, *a, **k)


_plan.Plan.gemm, _plan.Plan.call, _plan.Plan.call_desc = gemm, call, call_desc
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
