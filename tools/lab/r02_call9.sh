#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call9.txt
: > $O
timeout 1200 python -m pytest tests/test_gpu_one_call.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 >> $O
for rep in 1 2 3; do
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e8 12 >> $O 2>&1
VRS_FUSED_PLAN=0 timeout 300 python tools/lab/ab_bench.py new-separate-g32 1e8 12 32 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py new-fused-g32 1e8 12 32 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py new-fused-g16 1e8 12 16 >> $O 2>&1
done
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e7 40 >> $O 2>&1
VRS_FUSED_PLAN=0 timeout 300 python tools/lab/ab_bench.py new-separate-g8 1e7 40 8 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py new-fused-g8 1e7 40 8 >> $O 2>&1
cat $O
