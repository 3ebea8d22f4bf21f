"""prints the parts of a bench.py line a builder looks at first: usage show_bench.py FILE"""
import json
import sys

r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", r["value"], "ms/step", r["ms_per_step"], "individually", r.get("ms_per_step_individually_timed"))
rf = r["roofline"]
print("roofline:", rf.get("kernel"), rf.get("avg_launch_us"), "us frac", rf.get("frac"), "traffic", rf.get("traffic"), rf.get("instrumented_us_by_kernel"))
print("sort_roofline", r.get("sort_roofline"))
print("kernels (all instrumented)", r.get("kernels_all_instrumented_rerun"))
cp = r.get("contract_path")
if cp:
    print("contract", cp["ms_per_step"], cp["frac_of_peak"], cp["kernels"])
for k, v in (r.get("configs") or {}).items():
    print(k, v.get("ms_per_step"), v.get("frac_of_peak"))
