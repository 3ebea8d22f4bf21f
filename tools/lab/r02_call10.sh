#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call10.txt
: > $O
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r02_pytest_gpu.txt | tail -3 >> $O
python -c "import __graft_entry__ as g; g.smoke()" >> $O 2>&1
timeout 600 python bench.py > gpurun_out/r02_bench_b.json 2>> $O
timeout 600 python bench.py --n 1e7 --steps 50 --warmup 5 > gpurun_out/r02_bench_1e7_b.json 2>> $O
python - >> $O <<'PY'
import json
for f in ("gpurun_out/r02_bench_b.json", "gpurun_out/r02_bench_1e7_b.json"):
    d = json.load(open(f))
    print(f, d["value"], d["ms_per_step"], d["ms_per_step_individually_timed"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["measured_d2d_copy_GBps"], d["sort_roofline"]["frac_of_peak"], d["kernels_all_instrumented_rerun"], d["contract_path"]["ms_per_step"])
PY
cat $O
