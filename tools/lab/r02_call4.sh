#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call4.txt
: > $O
echo "== full gpu test suite" >> $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> $O
echo "== bench default" >> $O
timeout 600 python bench.py > gpurun_out/r02_bench_a.json 2>> $O
cat gpurun_out/r02_bench_a.json >> $O
echo "== bench 1e7" >> $O
timeout 600 python bench.py --n 1e7 --steps 50 --warmup 5 > gpurun_out/r02_bench_1e7_a.json 2>> $O
cat gpurun_out/r02_bench_1e7_a.json >> $O
cat $O
