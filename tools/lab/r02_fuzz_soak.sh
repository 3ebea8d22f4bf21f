#!/bin/bash
# long randomised runs of the final code: fuzz (all paths and knobs vs numpy) and soak (back-to-back large one-call sorts)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_fuzz_soak.txt
: > $O
for seed in 201 202 203; do timeout 400 python tools/fuzz_gpu.py 240 $seed >> $O 2>&1; done
timeout 400 python tools/soak_one_call.py 240 11 keys >> $O 2>&1
timeout 300 python tools/soak_one_call.py 150 12 pairs >> $O 2>&1
timeout 300 python tools/soak_one_call.py 150 13 u64 >> $O 2>&1
timeout 300 python tools/soak_one_call.py 90 14 misplaced >> $O 2>&1
cat $O
