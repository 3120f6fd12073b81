#!/usr/bin/env python
"""Checks the hand-written DPP blocks of als_kernels.h on the generated ISA: a DPP source register must
not have been written by a VALU instruction in the two instruction slots before its read (gfx9 data
hazard, 2 wait states); hipcc does not track that into inline asm.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s mals_api.hip; check_dpp_hazards.py k.s"""
import re
import sys


def regs(op):
    """v7 -> {v7}; v[10:13] -> {v10..v13}; anything else -> {}"""
    m = re.fullmatch(r"v(\d+)", op)
    if m:
        return {op}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", op)
    if m:
        return {"v%d" % i for i in range(int(m.group(1)), int(m.group(2)) + 1)}
    return set()


def written(instr):
    """vector registers a VALU instruction writes: its first operand (a register or a range); v_permlane*_swap and
    v_swap write both of theirs"""
    parts = instr.split(None, 1)
    if len(parts) < 2:
        return set()
    ops = [o.strip() for o in parts[1].split(",")]
    out = regs(ops[0])
    if "swap" in parts[0] and len(ops) > 1:
        out |= regs(ops[1].split()[0])
    return out


def main():
    lines = open(sys.argv[1]).read().splitlines()
    instr = []                                     # (text, wait states it provides)
    bad = blocks = 0
    active = [True]
    for ln in lines:
        t = ln.strip()
        if t.startswith(".if "):
            expr = t[4:].replace("&&", " and ").replace("||", " or ")
            active.append(active[-1] and bool(eval(expr)))
            continue
        if t == ".endif":
            active.pop()
            continue
        if not active[-1] or not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        if "_dpp" in t:
            blocks += 1
            m = re.match(r"(\S+)\s+(v\d+),\s*(v\d+)", t)
            src = m.group(3)
            slots = 0
            for prev in reversed(instr[-4:]):
                if slots >= 2:
                    break
                pm = re.match(r"s_nop\s+(\d+)", prev)
                if pm:
                    slots += int(pm.group(1)) + 1
                    continue
                if prev.startswith("v_") and "readlane" not in prev and src in written(prev):
                    bad += 1
                    print("HAZARD:", prev, "->", t)
                slots += 1
        instr.append(t)
    print("%d DPP instructions checked, %d hazards" % (blocks, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
