#!/usr/bin/env python3
"""Static check of rocprofv3 counter files (`pmc:` lines) against the gfx950 per-block counter-slot budget, and a
guarded profiler launch.

Why: a `pmc:` line that needs more slots of one block than the hardware has makes rocprofv3 abort with signal 6 in
rocprofiler_create_counter_config (error 38, "Request exceeds the capabilities of the hardware") and then hang in
its finaliser until the lease watchdog kills the box after 7 silent minutes (round 1 lost ~14 GPU-minutes to this,
profiles/r01_notes.md). Budget per pass (MI355X_MICROARCH.md, "rocprofv3 PMC slots"): SQ 8, TCC 4 (FETCH_SIZE costs
3 of them, WRITE_SIZE 2), GRBM 2; the other blocks (TCP, TA, TD, ...) are held to 4, the most any committed
pass has used successfully.

  python tools/check_pmc.py profiles/pmc_*.txt            # exit 1 when a line is over budget
  python tools/check_pmc.py --run profiles/pmc_x.txt -d gpurun_out/pmc -- python3 bench.py --steps 3
      checks first, then runs `rocprofv3 -i FILE --kernel-trace -d DIR -- CMD` under a timeout and exits non-zero
      when the profiler aborted (signal, "exceeds the capabilities", missing output) instead of hanging.
"""
from __future__ import annotations

import argparse
import os
import re
import signal
import subprocess
import sys

SLOTS = {"SQ": 8, "TCC": 4, "GRBM": 2, "TCP": 4, "TA": 4, "TD": 4, "SPI": 4, "CPC": 2, "CPF": 2, "GDS": 4}
# derived / wide counters that occupy several slots of their block
COST = {"FETCH_SIZE": ("TCC", 3), "WRITE_SIZE": ("TCC", 2)}


def block_of(counter: str):
    if counter in COST:
        return COST[counter]
    m = re.match(r"([A-Z]+)_", counter)
    if not m:
        return None, 1
    return m.group(1), 1


def check_line(line: str):
    """Return (usage dict, list of problems) for one `pmc:` line."""
    names = line.split(":", 1)[1].split()
    use, problems = {}, []
    for n in names:
        blk, cost = block_of(n)
        if blk is None or blk not in SLOTS:
            problems.append(f"unknown block for counter {n!r} (add it to SLOTS before trusting this line)")
            continue
        use[blk] = use.get(blk, 0) + cost
    for blk, u in use.items():
        if u > SLOTS[blk]:
            problems.append(f"{blk}: {u} slots needed, {SLOTS[blk]} available")
    if len(set(names)) != len(names):
        problems.append("duplicate counter in one pass")
    return use, problems


def check_file(path: str):
    bad = []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            s = line.strip()
            if not s or s.startswith("#"):
                continue
            if not s.startswith("pmc:"):
                bad.append((ln, s, [f"not a `pmc:` line"]))
                continue
            _, problems = check_line(s)
            if problems:
                bad.append((ln, s, problems))
    return bad


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--run", metavar="PMCFILE", help="check PMCFILE, then run rocprofv3 with it (command after --)")
    ap.add_argument("-d", "--outdir", default="gpurun_out/pmc")
    ap.add_argument("--timeout", type=int, default=300)
    argv = sys.argv[1:] if argv is None else argv
    cmd = []
    if "--" in argv:
        i = argv.index("--")
        argv, cmd = argv[:i], argv[i + 1:]
    a = ap.parse_args(argv)
    files = list(a.files) + ([a.run] if a.run else [])
    if not files:
        ap.error("no counter file given")
    rc = 0
    for p in files:
        for ln, s, problems in check_file(p):
            rc = 1
            print(f"{p}:{ln}: {s}\n    -> " + "; ".join(problems))
    if rc:
        print("check_pmc: over-budget or unknown counters; not handing this to rocprofv3", file=sys.stderr)
        return rc
    print(f"check_pmc: {len(files)} file(s) fit the gfx950 slot budget")
    if not a.run:
        return 0
    if not cmd:
        ap.error("--run needs a command after --")
    os.makedirs(a.outdir, exist_ok=True)
    full = ["rocprofv3", "-i", a.run, "--kernel-trace", "--output-format", "csv", "-d", a.outdir, "--"] + cmd
    print("[check_pmc]", " ".join(full), flush=True)
    # own process group: on a profiler abort ("finalizing after signal 6" never returns) kill the whole group
    p = subprocess.Popen(full, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    aborted = None
    try:
        out, _ = p.communicate(timeout=a.timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, _ = p.communicate()
        aborted = f"timeout after {a.timeout} s"
    sys.stdout.write(out[-4000:])
    if aborted is None and p.returncode != 0:
        aborted = f"exit code {p.returncode}"
    if aborted is None and re.search(r"exceeds the capabilities|signal 6|Aborted", out):
        aborted = "profiler abort message in the output"
    if aborted:
        print(f"[check_pmc] profiler run FAILED: {aborted}", file=sys.stderr)
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
