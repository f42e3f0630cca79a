"""average each PMC counter over the dispatches of one kernel (substring match)"""
import csv, sys, glob, collections
root, pat = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(root + "/p*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
