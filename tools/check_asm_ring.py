"""Static check of the hand-counted inline-asm load rings (csrc/token_block.hip tb_gemm, csrc/wgrad_planes.hip): in the compiler's assembly, no
instruction may touch the destination registers of an inline-asm global_load between the load and the inline-asm `s_waitcnt vmcnt(N)` that
guarantees it has landed (loads retire in order: after vmcnt(N) only the N youngest may be outstanding, so a load with at least N inline-asm
loads issued after it is complete) -- the compiler believes an asm output is valid immediately, so a spill
or copy of an in-flight register would silently read stale data.      usage: python tools/check_asm_ring.py file.s"""
import re
import sys


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    lines = open(path).read().split("\n")
    ins = []            # (line no, text, in_asm)
    in_asm, kernel = False, None
    bad = 0
    for no, l in enumerate(lines, 1):
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if re.match(r"^_Z\w+:", s):
            kernel = s[:-1]
            ins.append((no, "@kernel " + kernel, False))
            continue
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        ins.append((no, s.split(";")[0].strip(), in_asm))
    n_loads = 0
    for i, (no, text, asm) in enumerate(ins):
        if text.startswith("@kernel"):
            kernel = text[8:]
        if not (asm and text.startswith("global_load_dwordx4")):
            continue
        n_loads += 1
        dest = regs_of(text.split(",")[0])
        younger = 0
        for no2, t2, asm2 in ins[i + 1:]:
            if t2.startswith("@kernel") or t2.startswith("s_endpgm"):
                print(f"{path}:{no}: load {text.split(',')[0]} never guarded before the end of {kernel}")
                bad += 1
                break
            if asm2 and t2.startswith("global_load_dwordx4"):
                if regs_of(t2.split(",")[0]) & dest:
                    print(f"{path}:{no2}: ring register reloaded before its previous load was guarded (issued at line {no})")
                    bad += 1
                    break
                younger += 1
                continue
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t2)
            if asm2 and m:
                if younger >= int(m.group(1)):
                    break
                continue
            if regs_of(t2) & dest:
                print(f"{path}:{no2}: `{t2}` touches {sorted(regs_of(t2) & dest)} of the in-flight load at line {no} ({kernel})")
                bad += 1
                break
    print(f"{path}: {n_loads} inline-asm loads checked, {bad} violations")
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
