#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_hybrid_val.txt
: > $O
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu.txt 2>&1
grep -E "passed|failed|error|Error" gpurun_out/r02_pytest_gpu.txt | tail -3 >> $O
for seed in 301 302; do timeout 300 python tools/fuzz_gpu.py 200 $seed >> $O 2>&1; done
timeout 300 python tools/soak_one_call.py 200 21 keys >> $O 2>&1
timeout 600 python bench.py > gpurun_out/r02_bench_hybrid.json 2>> $O
python - >> $O <<'PY'
import json
d = json.load(open("gpurun_out/r02_bench_hybrid.json"))
print(d["value"], d["ms_per_step"], d["ms_per_step_uninstrumented_rerun"], d["ms_per_step_individually_timed"], d["roofline"], d["sort_roofline"], d["kernels_all_instrumented_rerun"], d["contract_path"]["ms_per_step"], d["config"]["path"][:60], d["verified"])
PY
for rep in 1 2 3; do
VRS_LIB_LENIENT=1 VRS_LIB=tools/lab/libs/libvrs_r01.so timeout 300 python tools/lab/ab_bench.py r01 1e8 12 >> $O 2>&1
timeout 300 python tools/lab/ab_bench.py hybrid 1e8 12 >> $O 2>&1
done
cat $O
