#!/bin/bash
# Collects the rocprofv3 evidence bench.py's numbers are judged against.  Run on the GPU box:
#   gpurun -- "VRS_COMMIT=$(git rev-parse --short HEAD) tools/collect_profiles.sh r06"      the default bench command (configs[2], 10^8 keys);
#                                                                   VRS_COMMIT stamps the traffic files (the GPU box has no .git)
#   gpurun -- 'tools/collect_profiles.sh r03_pairs --pairs'         configs[3]
#   gpurun -- 'tools/collect_profiles.sh r03_1e7 --n 1e7 --steps 50 --warmup 5'   configs[1]
# then copy gpurun_out/profiles_<tag>/<tag>_* (and the *_traffic.json files) into profiles/ (tracked).
set -u
ROUND=${1:-r01}
shift || true
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/profiles_$ROUND
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs $*"
# 1. per-kernel time: kernel trace + stats of the bench command
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
# 2. HBM traffic: separate --pmc passes (TCC: FETCH_SIZE costs 3 slots, WRITE_SIZE 2 -- not both in one pass)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o bench -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o bench -- $CMD > /dev/null 2> $OUT/pmc_write.log
python tools/profile_summary.py $OUT $ROUND
ls -la $OUT
