"""The hand-counted inline-asm load rings (tb_gemm in csrc/token_block.hip, the register ring of csrc/wgrad_planes.hip) are only correct if the
compiler never touches a ring register between its load and the `s_waitcnt vmcnt(N)` that guards it -- it believes an asm output is valid at once,
so a spill or a copy of an in-flight register would read stale data without any test on small shapes necessarily noticing.  This compiles the two
sources to gfx950 assembly and checks exactly that (tools/check_asm_ring.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,min_loads", [("token_block.hip", 500), ("wgrad_planes.hip", 12)])
def test_no_instruction_touches_an_in_flight_ring_register(tmp_path, src, min_loads):
    import check_asm_ring
    out = tmp_path / (src + ".s")
    csrc = os.path.join(ROOT, "eeg_image_decode_amd", "csrc")
    from eeg_image_decode_amd import build                            # the library's own compiler flags: the check is about THAT code generation
    subprocess.run([HIPCC, *build.FLAGS, "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", str(out), os.path.join(csrc, src)],
                   check=True, cwd=csrc, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    assert text.count("global_load_dwordx4") >= min_loads             # the rings are really in this build
    assert check_asm_ring.check(str(out)) == 0
