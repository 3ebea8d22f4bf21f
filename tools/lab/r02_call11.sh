#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call11.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_one_call.py -x -q -m gpu 2>&1 | grep -E "passed|failed" >> $O
for rep in 1 2 3; do
VRS_LIB=tools/lab/libs/libvrs_prev.so timeout 300 python tools/lab/ab_bench.py prev-waits 1e8 12 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py rolling-u8 1e8 12 >> $O 2>&1
for v in u4 u6 u12; do VRS_LIB=tools/lab/libs/libvrs_$v.so timeout 300 python tools/lab/ab_bench.py rolling-$v 1e8 12 >> $O 2>&1; done
done
VRS_LIB=tools/lab/libs/libvrs_prev.so timeout 300 python tools/lab/ab_bench.py prev-waits 1e7 40 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py rolling-u8 1e7 40 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_u4.so timeout 300 python tools/lab/ab_bench.py rolling-u4 1e7 40 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_prev.so timeout 300 python tools/lab/ab_bench.py prev-waits 3e7 20 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py rolling-u8 3e7 20 >> $O 2>&1
cat $O
