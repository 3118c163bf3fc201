"""Turn rocprofv3 --pmc output of `bench.py` into profiles/traffic_rNN.json (HBM bytes per launch of the iteration
kernel): python profiles/make_traffic_json.py DIR PAIRS_PER_LAUNCH POINTS VARIANT [OUT.json]. FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced
streaming read (MI355X_MICROARCH.md, HBM section) and is doubled here. Levels are told apart by dispatch order:
per tracking (one engine) the iteration kernel runs maxiter times at level 2, then 1, then 0; with two engines on two
streams each engine's launches are still consecutive dispatch ids (ids are assigned at enqueue time, and an engine
enqueues its whole tracking at once), and under --pmc rocprofv3 serialises the kernels, so every dispatch is counted
alone."""
import csv, glob, json, sys, collections
root, batch, points, variant, maxiter, levels = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), 10, 3
# rocprofv3 names the pass directories pmc_1, pmc_2, ...; every pass re-runs the whole command
rows = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_iter8" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
out = {"batch": batch, "points": points, "variant": variant, "source": "rocprofv3 -i profiles/pmc_traffic.txt (separate passes)"}
for c, v in rows.items():
    v.sort()
    per_level = [[], [], []]
    for k, (_, val) in enumerate(v):
        per_level[levels - 1 - (k % (maxiter * levels)) // maxiter].append(val)
    out[c + "_mean_per_level"] = [sum(x) / max(len(x), 1) for x in per_level]
if "FETCH_SIZE" in rows:
    fetch0 = out["FETCH_SIZE_mean_per_level"][0] * 1024 * 2   # KiB -> B, gfx950 x2 correction
    write0 = out.get("WRITE_SIZE_mean_per_level", [0])[0] * 1024
    out["hbm_bytes_per_launch_level0"] = fetch0 + write0
    fm = out["FETCH_SIZE_mean_per_level"]
    wm = out.get("WRITE_SIZE_mean_per_level", [0, 0, 0])
    out["hbm_bytes_per_launch_mean"] = sum(f * 2048 + w * 1024 for f, w in zip(fm, wm)) / len(fm)
    out["note"] = "FETCH_SIZE x 1024 x 2 (gfx950 coalesced-read correction) + WRITE_SIZE x 1024; 4-byte-per-lane loads are 'uncalibrated' per the guide, so read this as an upper bound of ~2x the raw counter"
json.dump(out, open(sys.argv[5] if len(sys.argv) > 5 else "profiles/traffic_r01.json", "w"), indent=1)
print(json.dumps(out, indent=1))
