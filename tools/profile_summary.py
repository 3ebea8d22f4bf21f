"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh into the small files committed under profiles/."""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(name):
    # (kernels in an unnamed namespace: "vrs::(anonymous namespace)::pool_pass_a_kernel(...)")
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


out = None  # set by the command line below


def counter_per_dispatch(sub, counter, root=None):
    """per kernel name: (counter value per dispatch, dispatches)"""
    acc = defaultdict(float)
    disp = defaultdict(set)
    for f in glob.glob(f"{root or out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    return {k: (acc[k] / max(len(disp[k]), 1), len(disp[k])) for k in acc}


def traffic_rows(fetch, write):
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        # FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B,
        # i.e. reports half of the bytes a coalesced stream fetches (MI355X_MICROARCH.md, HBM section) -> x2.
        # Calibration on this access pattern: the scatter's corrected read bytes = N*4 keys + W*1 KiB offsets.
        f, fd = fetch.get(k, (0.0, 0))
        w, wd = write.get(k, (0.0, 0))
        fb, wb = f * 1024 * 2, w * 1024
        rows[k] = {"fetch_bytes_corrected_x2": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb, "dispatches": max(fd, wd),
                   "raw_FETCH_SIZE": f, "raw_WRITE_SIZE": w}
    return rows


# The dominant kernels bench.py reports traffic for.  A kernel name matches several template instantiations (the one-call
# sort enqueues a candidate first pass that the plan disarms: ONE dispatch that reads a word and leaves): the record is the
# instantiation that moved the most bytes in total (bytes per launch x dispatches), never just the first name.
def pick_dominant(rows, match):
    hit = [(k, v) for k, v in rows.items() if match(k)]
    if not hit:
        return None
    k, v = max(hit, key=lambda kv: kv[1]["hbm_bytes_per_launch"] * max(kv[1]["dispatches"], 1))
    return dict(v, instantiation=k)


def dominant_records(rows, pairs=False):
    """file name -> record; the first MSD pass and the second are separate records of one file (bench.py reports both)"""
    src = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 10 --warmup 2 --no-cpu-baseline`"
    corr = "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1; both x1024 B"
    files = {}
    rec = pick_dominant(rows, lambda k: "scatter_kernel" in k and "onesweep" not in k)
    if rec:
        files["scatter_traffic.json"] = {"kernel": "scatter_kernel", "source": src, "correction": corr, **rec}
    first = pick_dominant(rows, lambda k: "onesweep_scatter_kernel" in k)
    second = pick_dominant(rows, lambda k: "msd_pass_b_kernel" in k)
    if first:
        name = "lookback_scatter_pairs_traffic.json" if pairs else "lookback_scatter_traffic.json"
        files[name] = {"kernel": "the MSD / look-back scatter passes of the one-call sort's counted forms", "source": src, "correction": corr, **first,
                       "passes": [r for r in (first, second) if r]}
    # one file per profile name of bench.py (KERNEL_BYTES_PER_KEY): whichever its `roofline` block names finds its traffic
    for fname, what, match in (("local_sort_traffic.json", "the LDS-local sort of every bucket", lambda k: "local_sort" in k),
                               ("pool_pass_a_traffic.json", "pool_pass_a_kernel", lambda k: "pool_pass_a_kernel" in k),
                               ("pool_pass_b_traffic.json", "pool_pass_b_kernel", lambda k: "pool_pass_b_kernel" in k),
                               ("histogram_traffic.json", "histogram_kernel", lambda k: "histogram_kernel" in k),
                               ("digit_tables_traffic.json", "digit_tables_kernel", lambda k: "digit_tables_kernel" in k)):
        rec = pick_dominant(rows, match)
        if rec:  # (a --pairs run: <kernel>_pairs_traffic.json -- the kernels' payload-carrying instantiations)
            files[fname.replace("_traffic", "_pairs_traffic") if pairs else fname] = {"kernel": what, "source": src, "correction": corr, **rec}
    return files



if __name__ == "__main__":

    out, rnd = sys.argv[1], sys.argv[2]
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



    fetch = counter_per_dispatch("pmc_fetch", "FETCH_SIZE")
    write = counter_per_dispatch("pmc_write", "WRITE_SIZE")
    rows = traffic_rows(fetch, write)
    json.dump(rows, open(f"{out}/{rnd}_hbm_traffic_per_launch.json", "w"), indent=1)
    # every traffic file says which code and which box it came from (VRS_COMMIT: the caller's `git rev-parse --short HEAD` -- the GPU box has no
    # .git --; the box: host name + the device's unique id): round 5's files were overwritten in place with neither
    import os
    import re
    import socket
    import subprocess
    try:
        uid = subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r"Unique ID:\s*(0x[0-9a-fA-F]+|[0-9a-fA-F]{8,})", uid)  # (not the table's "==== Unique ID ====" header line)
        uid = m.group(1) if m else ""
    except Exception:  # noqa: BLE001
        uid = ""
    stamp = {"round": rnd, "commit": os.environ.get("VRS_COMMIT", "unknown"), "box": f"{socket.gethostname()} {uid}".strip()}
    for fname, rec in dominant_records(rows, pairs="pairs" in rnd).items():
        json.dump(dict(rec, **stamp), open(f"{out}/{fname}", "w"), indent=1)
    json.dump(stamp, open(f"{out}/{rnd}_stamp.json", "w"), indent=1)
    print(open(f"{out}/{rnd}_bench_kernel_stats.csv").read())
    print(json.dumps(rows, indent=1))

