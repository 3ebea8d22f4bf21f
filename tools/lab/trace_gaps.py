"""Kernel (and copy) timeline of a rocprofv3 --kernel-trace [--memory-copy-trace] run: the last `count` operations with their
durations and the gaps between them.  usage: trace_gaps.py <dir> [count]"""
import csv
import glob
import sys

d = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ops.sort()
ops = ops[-count:]
prev_end = ops[0][0]
for s, e, name in ops:
    print(f"{(e - s) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:8.1f} us  {name}")
    prev_end = max(prev_end, e)
