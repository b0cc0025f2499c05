"""Build the kernel sources for x86 under the lane emulator (TEST ONLY; see hipemu.h)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "eeg_image_decode_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libeegclip_emu.so")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-DEEG_EMU", "-I", HERE, "-ffp-contract=off",
         "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed"]


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from eeg_image_decode_amd import build as product_build
    product_build.generate_plan_dispatch()                       # csrc/plan_exec_gen.inc (generated from the ctypes prototypes)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip")) + [os.path.join(HERE, "hipemu.cpp")]
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))] + [os.path.join(HERE, "hipemu.h"), os.path.join(ROOT, "include", "eegclip.h")]
    hm = max(os.path.getmtime(h) for h in hdrs)
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hm):
            jobs.append((s, o))

    def cc(j):
        cmd = [CXX, *FLAGS, "-c", j[0], "-o", j[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=8) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(LIB):
        subprocess.check_call([CXX, "-shared", "-fPIC", *objs, "-o", LIB, "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
