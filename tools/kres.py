#!/usr/bin/env python3
"""Kernel resource usage of one translation unit: VGPRs, spills, occupancy, LDS per kernel (gfx950), from
`hipcc -Rpass-analysis=kernel-resource-usage`.   python tools/kres.py ictr_resident.hip [filter] [extra flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import CSRC, HIPCC_FLAGS  # noqa: E402


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    path = src if os.path.exists(src) else os.path.join(CSRC, src)
    flags = [f for f in HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    cmd = ["/opt/rocm/bin/hipcc"] + flags + extra + ["-c", path, "-o", "/dev/null",
                                                      "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    cur = {}
    rows = []
    for ln in err.splitlines():
        m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+)", ln)
        if not m:
            if "error" in ln:
                print(ln)
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        else:
            cur[k] = v
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), stdout=subprocess.PIPE,
                           text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        n = re.sub(r"ictr::|\(ictr::EngineDev.*", "", n)
        if flt and flt not in n:
            continue
        print(f"{n[:70]:70s} vgpr {r.get('VGPRs', '?'):>4} agpr {r.get('AGPRs', '?'):>3} spill {r.get('VGPRs Spill', '?'):>3} "
              f"sgpr {r.get('TotalSGPRs', '?'):>4} sspill {r.get('SGPRs Spill', '?'):>3} occ {r.get('Occupancy', '?'):>2} "
              f"lds {r.get('LDS Size', '?'):>6} scratch {r.get('ScratchSize', '?')}")


if __name__ == "__main__":
    main()
