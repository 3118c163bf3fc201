"""Per-level means of PMC counters for the regular k_iter8 launches (dispatch order: 10 launches per level, levels
lv_f..0, first launch of a level is the WH instantiation and is listed separately)."""
import collections, csv, glob, sys
root = sys.argv[1]
maxiter, levels = 10, 3
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_iter8" in r["Kernel_Name"]]
    per_counter = collections.defaultdict(list)
    for r in rows:
        per_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), "true>" in r["Kernel_Name"].split("k_iter8")[1][:24].replace(" ", "") and r["Kernel_Name"].split("k_iter8<")[1].split(">")[0].replace(" ", "").endswith("true,true")))
    for c, v in per_counter.items():
        v.sort()
        for k, (_, val, first) in enumerate(v):
            lvl = levels - 1 - (k % (maxiter * levels)) // maxiter
            acc[c][(lvl, "first" if (k % maxiter) == 0 else "regular")].append(val)
for c in sorted(acc):
    print(c)
    for key in sorted(acc[c]):
        v = acc[c][key]
        print(f"   level {key[0]} {key[1]:8s} n={len(v):3d} mean={sum(v)/len(v):14.1f}")
