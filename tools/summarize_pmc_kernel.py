"""Mean PMC counter values per kernel name from rocprofv3 counter_collection CSVs: python tools/summarize_pmc_kernel.py DIR SUBSTR"""
import collections, csv, glob, sys
root, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    v.sort()
    print(f"{k:60s} {c:22s} n={len(v):4d} mean={sum(v)/len(v):14.1f} max={v[-1]:14.1f} p90={v[int(0.9*(len(v)-1))]:14.1f}")
