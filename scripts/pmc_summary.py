"""summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of each counter for kernels matching a substring"""
import csv, glob, sys, collections
root, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:36s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
