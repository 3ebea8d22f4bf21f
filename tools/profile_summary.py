"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh into the small files committed under profiles/."""
import csv
import glob
import json
import sys
from collections import defaultdict

out, rnd = sys.argv[1], sys.argv[2]


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


# kernel stats from the trace
dur = defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = short(row["Kernel_Name"])
        us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        # the one-call sort enqueues its candidate first passes before the plan is known; the one the plan disarms leaves
        # at once (all its workgroups read one word and exit): listed apart, it is not a pass over the keys
        if "onesweep_scatter_kernel" in name and us < 10.0:
            name += " [speculative launch the plan disarmed: left at once]"
        dur[name].append(us)
total = sum(sum(v) for v in dur.values())
lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    lines.append(f'"{k}",{len(v)},{sum(v):.1f},{sum(v)/len(v):.2f},{min(v):.2f},{max(v):.2f},{100*sum(v)/total:.2f}')  # names hold commas
open(f"{out}/{rnd}_bench_kernel_stats.csv", "w").write("\n".join(lines) + "\n")

# rocprofv3's own stats file, verbatim
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    open(f"{out}/{rnd}_rocprofv3_kernel_stats.csv", "w").write(open(f).read())


def counter_per_dispatch(sub, counter):
    acc = defaultdict(float)
    disp = defaultdict(set)
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    return {k: acc[k] / max(len(disp[k]), 1) for k in acc}


fetch = counter_per_dispatch("pmc_fetch", "FETCH_SIZE")
write = counter_per_dispatch("pmc_write", "WRITE_SIZE")
rows = {}
for k in sorted(set(fetch) | set(write)):
    # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
    # i.e. reports half of the bytes a coalesced stream fetches (MI355X_MICROARCH.md, HBM section) -> x2.
    # Calibration on this access pattern: the scatter's corrected read bytes = N*4 keys + W*1 KiB offsets.
    fb = fetch.get(k, 0.0) * 1024 * 2
    wb = write.get(k, 0.0) * 1024
    rows[k] = {"fetch_bytes_corrected_x2": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
               "raw_FETCH_SIZE": fetch.get(k, 0.0), "raw_WRITE_SIZE": write.get(k, 0.0)}
json.dump(rows, open(f"{out}/{rnd}_hbm_traffic_per_launch.json", "w"), indent=1)
src = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 10 --warmup 2 --no-cpu-baseline`"
corr = "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1; both x1024 B"
# bench.py reads <dominant kernel>_traffic.json: the contract path's scatter and the one-call path's look-back scatter
pairs = "pairs" in rnd
for fname, kname, pick in (("scatter_traffic.json", "scatter_kernel", lambda k: "scatter_kernel" in k and "onesweep" not in k),
                           ("lookback_scatter_pairs_traffic.json" if pairs else "lookback_scatter_traffic.json", "onesweep_scatter_kernel",
                            lambda k: "onesweep_scatter_kernel" in k)):
    hit = [v for k, v in rows.items() if pick(k)]
    if hit:
        json.dump({"kernel": kname, "round": rnd, "source": src, "correction": corr, **hit[0]}, open(f"{out}/{fname}", "w"), indent=1)
print(open(f"{out}/{rnd}_bench_kernel_stats.csv").read())
print(json.dumps(rows, indent=1))
