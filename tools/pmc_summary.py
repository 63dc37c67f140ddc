"""Sums rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_summary.py DIR [kernel substring ...]
Prints counter totals and per-launch averages for the kernels whose name holds one of the substrings."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
subs = sys.argv[2:] or [""]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        for sub in subs:
            if sub in name:
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
                acc[short][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[short][r["Counter_Name"]].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        n = max(1, len(launches[k][c]))
        print("   %-26s total %18.0f   launches %5d   per launch %16.1f" % (c, acc[k][c], n, acc[k][c] / n))
