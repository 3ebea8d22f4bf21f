#!/bin/bash
# round-2 evidence with the final code: GPU tests, smoke, bench lines, rocprofv3 summaries, config matrix, sweeps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_final.txt
: > $O
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r02_pytest_gpu.txt 2>&1
grep -E "passed|failed|error|Error" gpurun_out/r02_pytest_gpu.txt | tail -3 >> $O
python -c "import __graft_entry__ as g; g.smoke()" >> $O 2>&1
timeout 600 python bench.py > gpurun_out/r02_bench_line.json 2>> $O
timeout 600 python bench.py --n 1e7 --steps 50 --warmup 5 > gpurun_out/r02_bench_line_1e7.json 2>> $O
bash tools/collect_profiles.sh r02 > gpurun_out/r02_collect.log 2>&1
timeout 1200 python tools/config_matrix.py 8 > gpurun_out/r02_config_matrix.json 2>> $O
timeout 900 python tools/one_call_sweep.py 5e3,1e4,2e4,5e4,1e5,2e5,5e5,1e6,2e6,5e6,1e7,2e7,5e7,1e8,2e8 > gpurun_out/r02_one_call_crossover.csv 2>> $O
python - >> $O <<'PY'
import json
for f in ("gpurun_out/r02_bench_line.json", "gpurun_out/r02_bench_line_1e7.json"):
    d = json.load(open(f))
    print(f, d["value"], d["ms_per_step"], d["ms_per_step_uninstrumented_rerun"], d["ms_per_step_individually_timed"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["measured_d2d_copy_GBps"], d["sort_roofline"]["frac_of_peak"], d["kernels_all_instrumented_rerun"], d["contract_path"]["ms_per_step"], d["contract_path"]["kernels"])
PY
cat gpurun_out/profiles_r02/r02_bench_kernel_stats.csv >> $O
cat gpurun_out/r02_config_matrix.json | python -c "import json,sys; d=json.load(sys.stdin); [print(k, v) for k,v in d.items()]" >> $O
cat gpurun_out/r02_one_call_crossover.csv >> $O
cat $O
