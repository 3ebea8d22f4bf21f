#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call5.txt
: > $O
echo "== one_call tests with the cooperative look-back" >> $O
timeout 900 python -m pytest tests/test_gpu_one_call.py -x -q -m gpu 2>&1 | tail -5 >> $O
echo "== A/B under bench conditions" >> $O
for rep in 1 2; do
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e8 12 >> $O 2>&1
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_prev.so timeout 300 python tools/lab/ab_bench.py prev-perthread 1e8 12 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py coop 1e8 12 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_2wg.so timeout 300 python tools/lab/ab_bench.py coop-2wg 1e8 12 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_late.so timeout 300 python tools/lab/ab_bench.py coop-late 1e8 12 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_late2wg.so timeout 300 python tools/lab/ab_bench.py coop-late-2wg 1e8 12 >> $O 2>&1
done
timeout 300 python tools/lab/ab_bench.py coop-g8 1e8 12 8 >> $O 2>&1
for lib in r01 prev; do VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_$lib.so timeout 300 python tools/lab/ab_bench.py $lib 1e7 40 >> $O 2>&1; done
timeout 300 python tools/lab/ab_bench.py coop 1e7 40 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py coop-g8 1e7 40 8 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_2wg.so timeout 300 python tools/lab/ab_bench.py coop-2wg-g8 1e7 40 8 >> $O 2>&1
VRS_LIB=tools/lab/libs/libvrs_late.so timeout 300 python tools/lab/ab_bench.py coop-late-g8 1e7 40 8 >> $O 2>&1
echo "== lookback lab, cooperative" >> $O
timeout 300 tools/lab/lookback_lab 1e8 32 5 2>&1 | grep -v "^C" >> $O
timeout 300 tools/lab/lookback_lab_2wg 1e8 32 5 2>&1 | grep -E "^n=|^A|look-back|^B|^D|longest|lifetime" >> $O
timeout 300 tools/lab/lookback_lab 1e7 8 10 2>&1 | grep -E "^n=|^A|look-back|^B|lifetime|longest" >> $O
cat $O
