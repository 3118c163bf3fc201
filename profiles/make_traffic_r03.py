"""rocprofv3 --pmc passes of `bench.py` (tools/prof_r03.sh) -> profiles/traffic_r03.json: HBM-side bytes per launch of
the resident-iteration kernel and of the setup kernel, per pyramid level.
    python profiles/make_traffic_r03.py gpurun_out/pmc 32 32400 [OUT.json]
FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read
(MI355X_MICROARCH.md, HBM section): doubled here as prescribed; the kernels' 4- and 8-byte-per-lane gathers are
"uncalibrated" there, so the doubled figure is an upper bound and the raw one is kept beside it. FETCH_SIZE counts what
the L2s requested from the fabric -- Infinity-Cache hits included -- not DRAM bytes alone. Levels are told apart by
dispatch order: per tracking one launch per level, level 2 first (under --pmc rocprofv3 serialises the kernels)."""
import collections
import csv
import glob
import json
import sys

root, batch, points = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
levels = 3
out = {"batch": batch, "points": points, "kernel": "k_level_resident",
       "source": "rocprofv3 -i profiles/pmc_traffic.txt --kernel-trace (separate passes; tools/prof_r03.sh)"}
for key, sub in (("k_level_resident", "k_level_res"), ("k_ref8", "k_ref8")):
    rows = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    rec = {}
    for c, v in rows.items():
        v.sort()
        per = [[] for _ in range(levels)]
        for k, (_, val) in enumerate(v):
            per[levels - 1 - (k % levels)].append(val)
        rec[c + "_mean_per_level"] = [sum(x) / max(len(x), 1) for x in per]
        rec["launches_counted"] = len(v)
    if "FETCH_SIZE_mean_per_level" in rec:
        fm, wm = rec["FETCH_SIZE_mean_per_level"], rec.get("WRITE_SIZE_mean_per_level", [0.0] * levels)
        rec["hbm_bytes_per_launch_per_level"] = [f * 2048 + w * 1024 for f, w in zip(fm, wm)]
        rec["hbm_bytes_per_launch_per_level_raw_fetch"] = [f * 1024 + w * 1024 for f, w in zip(fm, wm)]
        rec["hbm_bytes_per_launch_mean"] = sum(rec["hbm_bytes_per_launch_per_level"]) / levels
    out[key] = rec
out["hbm_bytes_per_launch_mean"] = out["k_level_resident"].get("hbm_bytes_per_launch_mean")
px = batch * points * 64
out["algorithmic_bytes_per_launch_16B_per_px_per_iteration_x10"] = 160.0 * px
out["minimal_bytes_per_launch_templates_once_plus_frame_once"] = [(12.0 + 4.0 / 4 ** l) * px for l in range(levels)]
out["note"] = ("k_level_resident: T / Gx / Gy cross HBM once per level (12 B per patch pixel = %.0f MB per launch); the "
               "rest is the current frame's windows, re-fetched by every iteration where the frames in flight exceed the "
               "L2s (level 0: 4 pairs x 8.5 MB per XCD against 4 MB of L2; served by the Infinity Cache)." % (12.0 * px / 1e6))
json.dump(out, open(sys.argv[4] if len(sys.argv) > 4 else "profiles/traffic_r03.json", "w"), indent=1)
print(json.dumps(out, indent=1))
