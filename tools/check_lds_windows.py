#!/usr/bin/env python
"""Checks on the generated ISA that no instruction reads the destination of an LDS instruction (ds_bpermute_b32,
ds_read_*) before an s_waitcnt has covered it.  hipcc's own waitcnt pass guarantees that for the LDS instructions it
emits; the split column fetch of the gather (bperm2_i_start / bperm2_i_land in als_kernels.h) issues its two
ds_bpermute in one asm statement and waits in another, and a register copy or a spill of the destinations placed
between the two by the compiler would read them before they land.
LDS results return in issue order: `s_waitcnt lgkmcnt(k)` leaves at most the k youngest lgkm operations outstanding
(scalar loads share the counter; with one outstanding hipcc only ever waits for 0).  Labels reset the state (linear scan).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only -o k.s mals_api.hip; check_lds_windows.py k.s"""
import re
import sys


def regs(tok):
    """VGPR numbers named by one operand token (v12, v[4:7], -v3, |v5|)."""
    m = re.search(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.search(r"\bv(\d+)\b", tok)
    return {int(m.group(1))} if m else set()


def main():
    pending = []          # lgkm operations in issue order: set of destination VGPRs (empty for stores / scalar loads)
    bad = checked = 0
    active = [True]
    for ln in open(sys.argv[1]).read().splitlines():
        t = ln.strip()
        if t.startswith(".if "):
            active.append(active[-1] and bool(eval(t[4:].replace("&&", " and ").replace("||", " or "))))
            continue
        if t == ".endif":
            active.pop()
            continue
        if not active[-1] or not t:
            continue
        if t.split()[0].endswith(":"):   # a label (with or without a trailing comment): other paths join here
            pending = []
            continue
        if t[0] in ";./":
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        op, _, rest = t.partition(" ")
        ops = [o.strip() for o in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                k = int(m.group(1))
                pending = pending[len(pending) - k:] if k else []
            elif "lgkmcnt" not in t and re.fullmatch(r"s_waitcnt\s+\d+", t):
                pending = []  # raw immediate form: treat as a full wait
            continue
        # sources: every operand except the destination of instructions that have one
        has_dst = op.startswith(("v_", "ds_read", "ds_bpermute", "global_load", "scratch_load", "buffer_load", "flat_load"))
        srcs = set()
        for o in (ops[1:] if has_dst else ops):
            srcs |= regs(o)
        if op in ("v_fmac_f32", "v_fmac_f32_e32", "v_fmac_f32_e64", "v_fmac_f32_dpp") or op.startswith("v_mfma") and len(ops) >= 4:
            srcs |= regs(ops[0]) if op.startswith("v_fmac") else set()
        live = set().union(*pending) if pending else set()
        if srcs & live:
            bad += 1
            print("READ BEFORE WAIT:", t, "reads", sorted(srcs & live))
        if op.startswith("ds_"):
            dst = regs(ops[0]) if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) else set()
            pending.append(dst)
            checked += 1
        elif op.startswith(("s_load", "s_buffer_load")):
            pending.append(set())
    print("%d LDS instructions checked, %d reads before their wait" % (checked, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
