#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call14.txt
: > $O
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu.txt 2>&1
grep -E "passed|failed|error|Error" gpurun_out/r02_pytest_gpu.txt | tail -5 >> $O
timeout 300 python tools/small_n_sweep.py gpurun_out/r02_small_n_crossover.csv >> $O 2>&1
cat $O
