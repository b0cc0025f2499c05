"""The hand-counted inline-asm load rings (tb_gemm in csrc/token_block.hip) are only correct if the
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
@pytest.mark.parametrize("src,min_loads", [("token_block.hip", 500)])
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


GOOD = """
_Z4kernv:
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v[2:3], off
	;;#ASMEND
	;;#ASMSTART
	global_load_dwordx4 v[8:11], v[2:3], off offset:1024
	;;#ASMEND
	v_add_f32_e32 v20, v21, v22
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_mfma_f32_16x16x32_bf16 v[12:15], v[4:7], v[16:19], v[12:15]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mfma_f32_16x16x32_bf16 v[12:15], v[8:11], v[16:19], v[12:15]
	s_endpgm
"""


def test_the_checker_itself_flags_a_copy_of_an_in_flight_register(tmp_path):
    """the static check is not vacuous: the well-formed ring passes; a register copy ahead of the guarding wait, a use behind a wait that is too
    weak, and a load that is never guarded are each reported"""
    import check_asm_ring

    def run(text):
        f = tmp_path / "k.s"
        f.write_text(text)
        return check_asm_ring.check(str(f))

    assert run(GOOD) == 0
    assert run(GOOD.replace("\tv_add_f32_e32 v20, v21, v22", "\tv_mov_b32_e32 v30, v5")) == 1          # copies v5 while its load is in flight
    assert run(GOOD.replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)")) == 1                          # vmcnt(2) does not cover the first load
    assert run(GOOD.replace("\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n", "")) >= 1           # the second load is used unguarded
