#!/bin/bash
# HBM traffic of the one-call sort's kernels for a lab build of the library: tools/lab/pmc_variant.sh <variant> [N]
export TMPDIR=/tmp
V=$1; N=${2:-1e8}
OUT=$PWD/gpurun_out/pmcv_$V; rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && VRS_LIB=$OLDPWD/tools/lab/libs/libvrs_$V.so rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o p -- python $OLDPWD/tools/lab/ab_bench.py $V $N 4 > $OUT/$C.log 2>&1)
done
python - <<PY
import csv,glob,collections
for C,mult in (("FETCH_SIZE",2048),("WRITE_SIZE",1024)):
    acc=collections.defaultdict(float); disp=collections.defaultdict(set)
    for f in glob.glob("$OUT/"+C+"/**/*counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]!=C: continue
            k=r["Kernel_Name"].split("(")[0][-50:]
            acc[k]+=float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k in acc:
        if "local_sort" in k or "digit_tables" in k: print("$V", C, k, round(acc[k]/len(disp[k])*mult/1e6,1), "MB per launch")
PY
