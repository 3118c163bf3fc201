#!/usr/bin/env python3
"""Recompute bench.py's `roofline.frac` from committed rocprofv3 output -- no GPU needed.

    python profiles/recompute_roofline.py profiles/r03_kernel_trace_big.csv          # kernel trace (preferred)
    python profiles/recompute_roofline.py profiles/r03_streams1_kernel_stats.csv     # --stats summary of `--streams 1`
    python profiles/recompute_roofline.py <rocprof dir>/*_kernel_trace.csv --reduce profiles/rNN_kernel_trace_big.csv

roofline.frac of the driver line = algorithmic bytes of one k_iter8 launch / mean duration of the launches that ran
ALONE / 8 TB/s. Algorithmic bytes: 16 B per patch pixel = T, Gx, Gy and one current-frame
texel, x 64 pixels x points x frame pairs of the launch (frame pairs = the dispatch's grid size in y).

* From a kernel TRACE of the default command (`rocprofv3 --kernel-trace --stats -- python3 bench.py`): the two engines'
  launches overlap during the timed steps; bench.py then launches a marker (k_stream_read) and runs the same engines
  once more on ONE stream. The k_iter8 dispatches behind the LAST marker with the headline's frame pairs per launch
  are that leg; their mean duration is what bench.py measured with HIP events. Also printed: all k_iter8 dispatches
  (what --stats averages) and those whose interval overlaps no other big dispatch, wherever they ran.
  `--reduce` writes the few columns this needs for the big kernels only (small enough to commit).
* From a --stats SUMMARY: only meaningful for a run without overlap (`bench.py --streams 1`): sum(calls x average) over
  the k_iter8 instantiations / sum(calls); pass --pairs for the frame pairs per launch (32 with one engine).
"""
from __future__ import annotations

import argparse
import csv
import sys

PEAK = 8000.0e9
BIG = ("k_iter8", "k_ref8", "k_level_res", "k_iter4", "k_ref4", "k_stream_read")


def short(name):
    for k in BIG:
        if k in name:
            return k
    return None


def from_trace(path, points, reduce_to=None, pairs=None, res_pairs=32, maxiter=10):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, int(r["Grid_Size_Y"]),
                         r["Kernel_Name"].split("(")[0]))
    rows.sort()
    if pairs is None:
        pairs = 16  # bench.py's default: --batch 32 over --streams 2 engines
    if reduce_to:
        # keep the headline's dispatches (the secondary records launch thousands of small ones) and the markers
        rows = [r for r in rows if r[2] in ("k_stream_read", "k_level_res") or r[3] == pairs]
        with open(reduce_to, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_Y"])
            for s, e, k, gy, full in rows:
                w.writerow([full, s, e, gy])
        print(f"wrote {len(rows)} dispatches to {reduce_to}")
    # overlap test by a sweep: a dispatch is "alone" when no other big dispatch's interval intersects its own
    alone = [True] * len(rows)
    active = []  # indices of dispatches that have started and may still be running
    for i, (s, e, k, gy, _) in enumerate(rows):
        active = [j for j in active if rows[j][1] > s]
        for j in active:
            alone[i] = alone[j] = False
        active.append(i)
    it_all = [(e - s, gy) for (s, e, k, gy, _) in rows if k == "k_iter8" and gy == pairs]
    it_solo = [(e - s, gy) for a, (s, e, k, gy, _) in zip(alone, rows) if a and k == "k_iter8" and gy == pairs]
    if not it_solo:
        print("no un-overlapped k_iter8 dispatch in this trace")
        return 1
    frac = 0.0

    def report(tag, lst):
        byts = sum(16.0 * 64 * points * gy for _, gy in lst)
        ns = sum(d for d, _ in lst)
        rate = byts / (ns * 1e-9)
        print(f"{tag:28s} {len(lst):6d} launches, mean {ns / len(lst) / 1e3:8.2f} us, pairs/launch "
              f"{sum(gy for _, gy in lst) / len(lst):5.1f}: {rate / 1e9:8.1f} GB/s = {rate / PEAK:.4f} of 8 TB/s")
        return rate / PEAK

    report("k_iter8, all dispatches", it_all)
    frac = report("k_iter8, running alone", it_solo)
    marks = [i for i, r in enumerate(rows) if r[2] == "k_stream_read"]
    if marks:
        tail = [(e - s, gy) for (s, e, k, gy, _) in rows[marks[-1]:] if k == "k_iter8"]
        if tail:
            leg = [(d, gy) for d, gy in tail if gy == pairs]
            frac = report("k_iter8, solo leg (marker)", leg)
    else:
        print("(no k_stream_read marker in this trace: the figure is that of the dispatches running alone)")
    for k in ("k_ref8",):
        lst = [(e - s, gy) for a, (s, e, kk, gy, _) in zip(alone, rows) if a and kk == k]
        if lst:
            ns = sum(d for d, _ in lst)
            print(f"{k + ', running alone':28s} {len(lst):6d} launches, mean {ns / len(lst) / 1e3:8.2f} us")
    # resident-iteration headline (r03): the k_level_resident dispatches between the FIRST group of markers (the
    # streaming-read yardstick before the run) and the next one (the marker of the streaming leg behind the timed steps)
    groups, prev = [], None
    for i, r in enumerate(rows):
        if r[2] == "k_stream_read":
            if prev is None or i != prev + 1:
                groups.append([i, i])
            groups[-1][1] = i
            prev = i
    res = [(e - s) for (s, e, k, gy, _) in (rows[groups[0][1]:groups[1][0]] if len(groups) >= 2 else rows)
           if k == "k_level_res"]
    if res:
        mean = sum(res) / len(res)
        byts = 16.0 * 64 * points * res_pairs * maxiter
        rate = byts / (mean * 1e-9)
        print(f"{'k_level_resident, headline':28s} {len(res):6d} launches, mean {mean / 1e3:8.2f} us, {byts / 1e9:.2f} GB "
              f"algorithmic per launch ({res_pairs} pairs x {maxiter} iterations): {rate / 1e9:8.1f} GB/s equivalent = "
              f"{rate / PEAK:.4f} of 8 TB/s")
        print(f"roofline.frac = {rate / PEAK:.4f}   (headline kernel k_level_resident)")
        print(f"roofline.streaming_kernel.frac = {frac:.4f}   (k_iter8, solo leg)")
    else:
        print(f"roofline.frac = {frac:.4f}")
    return 0


def from_stats(path, points, pairs):
    calls = tot = 0.0
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if "k_iter8" in r["Name"]:
                c = float(r["Calls"])
                calls += c
                tot += c * float(r["AverageNs"])
                print(f"  {r['Name'].split('(')[0]:60s} calls {int(c):6d} avg {float(r['AverageNs']) / 1e3:8.2f} us")
    if calls == 0:
        print("no k_iter8 rows")
        return 1
    mean = tot / calls
    byts = 16.0 * 64 * points * pairs
    rate = byts / (mean * 1e-9)
    print(f"k_iter8: {int(calls)} launches, mean {mean / 1e3:.2f} us, {byts / 1e6:.1f} MB per launch ({pairs} pairs): "
          f"{rate / 1e9:.1f} GB/s")
    print(f"roofline.frac = {rate / PEAK:.4f}   (valid only if nothing overlapped: bench.py --streams 1)")
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--points", type=int, default=32400, help="8x8 patches per frame pair (bench default 240 x 135)")
    ap.add_argument("--pairs", type=int, default=None,
                    help="frame pairs per launch = the dispatch's grid size in y (trace input: default 16, bench.py's "
                         "--batch 32 over two engines; --stats input: default 32, one engine)")
    ap.add_argument("--res-pairs", type=int, default=32, help="frame pairs per k_level_resident launch (bench default 32)")
    ap.add_argument("--maxiter", type=int, default=10)
    ap.add_argument("--reduce", default=None, help="trace input: also write the reduced trace here")
    a = ap.parse_args()
    with open(a.csv, newline="") as f:
        header = f.readline()
    if "Start_Timestamp" in header:
        return from_trace(a.csv, a.points, a.reduce, a.pairs, a.res_pairs, a.maxiter)
    return from_stats(a.csv, a.points, a.pairs or 32)


if __name__ == "__main__":
    sys.exit(main())
