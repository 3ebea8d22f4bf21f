#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call15.txt
: > $O
for rep in 1 2; do
timeout 300 python tools/lab/ab_bench.py items16 1e8 12 >> $O 2>&1
for v in items12 items20 items24; do VRS_LIB=tools/lab/libs/libvrs_$v.so timeout 300 python tools/lab/ab_bench.py $v 1e8 12 >> $O 2>&1; done
done
timeout 300 python tools/lab/ab_bench.py items16 1e7 40 >> $O 2>&1
for v in items12 items20 items24; do VRS_LIB=tools/lab/libs/libvrs_$v.so timeout 300 python tools/lab/ab_bench.py $v 1e7 40 >> $O 2>&1; done
cat $O
