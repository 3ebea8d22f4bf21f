#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call13.txt
: > $O
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu.txt 2>&1
grep -E "passed|failed|error|Error" gpurun_out/r02_pytest_gpu.txt | tail -5 >> $O
timeout 600 python tools/one_call_sweep.py 5e3,1e4,2e4,3e4,5e4,7e4,1e5,2e5,5e5 >> $O 2>&1
for R in 1 2 4; do
VRS_BENCH_FORCE_MULTI=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python bench.py --rounds $R --rounds-forced --steps 10 --warmup 3 >> $O 2>&1
done
cat $O
