"""Summarise rocprofv3 --pmc output (counter_collection csv files): mean counter value per dispatch, per kernel."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ictr::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not (k.startswith("k_iter") or k.startswith("k_ref")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:42s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
