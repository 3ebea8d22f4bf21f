#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call6.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_one_call.py -x -q -m gpu 2>&1 | tail -3 >> $O
for rep in 1 2; do
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e8 12 >> $O 2>&1
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_prev.so timeout 300 python tools/lab/ab_bench.py prev 1e8 12 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py oneload 1e8 12 >> $O 2>&1
for v in late b2 late_b2 late_b3 late_both; do
VRS_LIB=tools/lab/libs/libvrs_$v.so timeout 300 python tools/lab/ab_bench.py oneload-$v 1e8 12 >> $O 2>&1
done
done
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e7 40 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py oneload-g8 1e7 40 8 >> $O 2>&1
for v in late late_b2 late_both; do
VRS_LIB=tools/lab/libs/libvrs_$v.so timeout 300 python tools/lab/ab_bench.py oneload-$v-g8 1e7 40 8 >> $O 2>&1
done
cat $O
