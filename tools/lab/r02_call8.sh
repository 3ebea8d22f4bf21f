#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call8.txt
: > $O
for rep in 1 2 3; do
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e8 12 >> $O 2>&1
VRS_FUSED_PLAN=0 timeout 300 python tools/lab/ab_bench.py new 1e8 12 32 >> $O 2>&1
VRS_LAB_EXACT_GRID=1 VRS_FUSED_PLAN=0 timeout 300 python tools/lab/ab_bench.py new-exactgrid 1e8 12 32 >> $O 2>&1
VRS_LAB_STATUS_PER_PASS=1 VRS_FUSED_PLAN=0 timeout 300 python tools/lab/ab_bench.py new-statusperpass 1e8 12 32 >> $O 2>&1
VRS_LAB_EXACT_GRID=1 VRS_LAB_STATUS_PER_PASS=1 VRS_FUSED_PLAN=0 timeout 300 python tools/lab/ab_bench.py new-both 1e8 12 32 >> $O 2>&1
done
cat $O
