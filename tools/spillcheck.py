#!/usr/bin/env python3
"""Where are a kernel's scratch (spill) accesses? Lists every scratch_load / scratch_store of a translation unit's ISA
with the loop depth of its basic block -- a spill in a level prologue is harmless, one in the iteration loop is not.
    python tools/spillcheck.py ictr_resident.hip [kernel-name-substring] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, HIPCC_FLAGS  # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    path = src if os.path.exists(src) else os.path.join(CSRC, src)
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + extra + ["-S", "--cuda-device-only", "-o", out, path],
                       stderr=subprocess.DEVNULL, check=True)
        lines = open(out).read().splitlines()
    kern, depth, counts = None, 0, {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kern, depth = m.group(1), 0
        m = re.search(r"Loop Header: Depth=(\d+)", ln) or re.search(r"in Loop: Header=\S+ Depth=(\d+)", ln)
        if m:
            depth = int(m.group(1))
        elif re.match(r"^\.LBB\d+_\d+:\s*$", ln) or (re.match(r"^\.LBB", ln) and "Loop" not in ln and "Depth" not in ln):
            depth = 0
        if "scratch_" in ln and kern and flt in kern:
            kind = "load " if "scratch_load" in ln else "store"
            counts.setdefault((kern, depth, kind), 0)
            counts[(kern, depth, kind)] += 1
    for (k, d, kind), n in sorted(counts.items()):
        print(f"{k[:60]:60s} loop depth {d}: {n:3d} scratch {kind}")
    if not counts:
        print("no scratch accesses")


if __name__ == "__main__":
    main()
