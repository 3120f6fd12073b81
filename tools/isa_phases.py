#!/usr/bin/env python3
"""Per-phase instruction histogram of one kernel in hipcc's -S output.  Phases are delimited by `; MARK name` comments
(asm volatile("; MARK name" ::: "memory") in a scratch copy of the source); assembler conditionals (.if / .endif, as
factor_diag uses them) are evaluated, so only instructions that are really assembled count.
usage: isa_phases.py file.s mangled_kernel_name [top_n]"""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 14
lines = open(path).read().split("\n")
s = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
e = next(i for i in range(s, len(lines)) if "s_endpgm" in lines[i])
phase = "start"
counts = collections.OrderedDict()
skip = []
for ln in lines[s:e]:
    t = ln.strip()
    m = re.match(r"\.if (.*)", t)
    if m:
        expr = m.group(1).replace("&&", " and ").replace("||", " or ")
        skip.append(not eval(expr))
        continue
    if t.startswith(".endif"):
        skip.pop()
        continue
    if any(skip):
        continue
    m = re.match(r"; MARK (\w+)", t)
    if m:
        phase = m.group(1)
        continue
    if not t or t[0] in ";." or t.split()[0].endswith(":"):   # comments, directives, labels (with or without a comment)
        continue
    op = t.split()[0].replace("_e32", "").replace("_e64", "")
    counts.setdefault(phase, collections.Counter())[op] += 1
total = 0
for ph, d in counts.items():
    n = sum(d.values())
    total += n
    print("%-8s %5d | %s" % (ph, n, ", ".join("%s %d" % kv for kv in d.most_common(top))))
print("total", total)
