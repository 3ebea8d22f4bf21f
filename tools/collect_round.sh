#!/bin/bash
# Everything a round commits under profiles/, from ONE box:  gpurun -- "VRS_COMMIT=$(git rev-parse --short HEAD) tools/collect_round.sh r06"
# (GPU tests' tail, rocprofv3 kernel stats + PMC traffic of the three bench commands, the bench lines, the config matrix, the block sweep, fuzz + soak)
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/round_$TAG; rm -rf $OUT; mkdir -p $OUT
filter() { grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | filter | tail -6 > $OUT/${TAG}_pytest_gpu_tail.txt
tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
tools/collect_profiles.sh ${TAG}_pairs --pairs > $OUT/collect_pairs.log 2>&1
tools/collect_profiles.sh ${TAG}_1e7 --n 1e7 --steps 50 --warmup 5 > $OUT/collect_1e7.log 2>&1
tools/collect_bench_lines.sh $TAG > $OUT/lines.log 2>&1
python tools/config_matrix.py > $OUT/${TAG}_config_matrix.json 2> $OUT/matrix.err
python tools/sweep_blocks.py > $OUT/${TAG}_block_sweep.csv 2> $OUT/sweep.err
{
python tools/fuzz_gpu.py 150 9601
python tools/soak_one_call.py 120 31 keys
python tools/soak_one_call.py 150 32 pairs
python tools/soak_one_call.py 60 33 u64
} 2>&1 | filter > $OUT/${TAG}_fuzz_soak.txt
cat $OUT/${TAG}_pytest_gpu_tail.txt $OUT/${TAG}_fuzz_soak.txt
tail -12 $OUT/lines.log
