"""Timeline of one step out of a rocprofv3 --kernel-trace csv: every kernel from the N-th launch of the anchor
kernel to the next launch of it (start offset, duration, gap to the previous kernel's end).
python tools/trace_window.py kernel_trace.csv ANCHOR_SUBSTRING [nth=1]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor, nth = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a = idx[nth]
b = idx[nth + 1] if nth + 1 < len(idx) else len(rows)
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("smh::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
    print("%10.1f us  %9.1f us  gap %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = e
    tot += e - s
print("window %.1f us, kernels %.1f us" % ((prev_end - t0) / 1e3, tot / 1e3))
