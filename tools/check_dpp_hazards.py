#!/usr/bin/env python
"""Checks the hand-written DPP blocks of als_kernels.h on the generated ISA: a DPP source register must
not have been written by a VALU instruction in the two instruction slots before its read (gfx9 data
hazard, 2 wait states); hipcc does not track that into inline asm.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s mals_api.hip; check_dpp_hazards.py k.s"""
import re
import sys


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
                wm = re.match(r"v_\S+\s+(v\d+)(?:,|$)", prev)
                if wm and wm.group(1) == src and "readlane" not in prev:
                    bad += 1
                    print("HAZARD:", prev, "->", t)
                slots += 1
        instr.append(t)
    print("%d DPP instructions checked, %d hazards" % (blocks, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
