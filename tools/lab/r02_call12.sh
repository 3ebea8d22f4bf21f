#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call12.txt
: > $O
bash tools/collect_profiles.sh r02 > gpurun_out/r02_collect.log 2>&1
tail -40 gpurun_out/r02_collect.log >> $O
timeout 1200 python tools/config_matrix.py 8 > gpurun_out/r02_config_matrix.json 2>> $O
cat gpurun_out/r02_config_matrix.json >> $O
timeout 900 python tools/one_call_sweep.py > gpurun_out/r02_one_call_crossover.csv 2>> $O
cat gpurun_out/r02_one_call_crossover.csv >> $O
cat $O
