"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
# rocprofv3 emits one row per (dispatch, counter[, dimension]); sum dimensions per dispatch is not
# distinguishable here, so report total / number of dispatches from the kernel trace
disp = defaultdict(int)
dur = defaultdict(float)
for f in glob.glob(root + "/sq1/**/*kernel_trace.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0]
            disp[name] += 1
            dur[name] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
for name in sorted(acc):
    n = max(disp.get(name, 0), 1)
    print(f"== {name[:110]}  dispatches={n} avg_dur_us={dur.get(name,0)/n/1e3:.1f}")
    for c in sorted(acc[name]):
        v = acc[name][c]
        print(f"   {c:28s} per-dispatch {sum(v)/n:16.1f}")
