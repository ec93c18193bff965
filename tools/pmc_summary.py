"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs.  usage: pmc_summary.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if sub in k:
            acc[k.split("(")[0][-60:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
