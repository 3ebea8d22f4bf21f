#!/bin/bash
# LDS counters of digit_tables_kernel on uniform vs sorted keys
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/dt_pmc; rm -rf $OUT; mkdir -p $OUT
for d in uniform sorted; do
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/$d -o p -- python tools/one_call_time.py 3e7 1 $d > $OUT/$d.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for d in ("uniform","sorted"):
    acc=collections.defaultdict(float); disp=set()
    for f in glob.glob(f"gpurun_out/dt_pmc/{d}/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if "digit_tables" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n=max(len(disp),1)
    print(d, {k: round(v/n) for k,v in sorted(acc.items())})
PY
