"""Build libeegclip_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain C ABI.

    python -m eeg_image_decode_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libeegclip_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "eegclip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    hdr_m = _deps_mtime()
    for src in sources():
        obj = src[:-4] + ".o"
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
