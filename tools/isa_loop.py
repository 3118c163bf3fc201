#!/usr/bin/env python3
"""Instruction mix of the innermost hot region of a kernel: every basic block of kernel `name` that contains a
`marker` instruction (default buffer_load_dwordx2), from an ISA listing made with hipcc -S.
    python tools/isa_loop.py /tmp/res.s k_level_residentILi32 [marker]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().splitlines()
name = sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "buffer_load_dwordx2"
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and re.match(r"^_Z\w+:", l)][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end")][0]
blocks, cur = [], None
for i in range(start, end):
    l = lines[i]
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = [l, []]
        blocks.append(cur)
    elif cur is not None and l.startswith("\t") and not l.strip().startswith((".", ";")):
        cur[1].append(l.strip())
tot = Counter()
n = 0
for nm, ins in blocks:
    if any(x.startswith(marker) for x in ins):
        n += len(ins)
        tot.update(x.split()[0] for x in ins)
        print(nm.split(";")[0].strip(), len(ins), (nm.split(";")[1].strip() if ";" in nm else ""))
print("total", n)
cls = Counter()
for op, c in tot.items():
    k = ("valu" if op.startswith("v_") else "nop" if op == "s_nop" else "waitcnt" if op == "s_waitcnt" else
         "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("buffer_", "global_")) else
         "scratch" if op.startswith("scratch_") else op)
    cls[k] += c
print(dict(cls))
print(tot.most_common(24))
