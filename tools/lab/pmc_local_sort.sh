#!/bin/bash
# SQ / LDS counters of the local-sort lab kernels (tools/lab/local_sort_lab2 1e8 lean): two rocprofv3 --pmc passes
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_local; rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/a -o p -- $OLDPWD/tools/lab/local_sort_lab2 1e8 lean > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -o p -- $OLDPWD/tools/lab/local_sort_lab2 1e8 lean > $OUT/b.log 2>&1
cd $OLDPWD
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_local/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); disp[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
for k in acc:
    print(k, {c: round(v/max(len(disp[(k,c)]),1)) for c,v in sorted(acc[k].items())})
PY
