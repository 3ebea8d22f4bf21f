#!/bin/bash
# The bench lines committed under profiles/ (unprofiled, as printed), all on ONE box:  gpurun -- 'tools/collect_bench_lines.sh r05'
TAG=${1:-r05}
OUT=$PWD/gpurun_out/lines_$TAG
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench.err
python bench.py --n 1e7 --steps 50 --warmup 5 > $OUT/${TAG}_bench_line_1e7.json 2> $OUT/bench_1e7.err
python bench.py --pairs > $OUT/${TAG}_bench_line_pairs.json 2> $OUT/bench_pairs.err
for R in 1 4; do
  VRS_DIST_SHAPE=hybrid VRS_BENCH_FORCE_MULTI=1 python bench.py --rounds $R --rounds-forced --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_multi_world1_hybrid_shape_R$R.json 2> $OUT/multi_R$R.err
  VRS_BENCH_FORCE_MULTI=1 python bench.py --rounds $R --rounds-forced --steps 10 --warmup 2 > $OUT/${TAG}_bench_multi_world1_byte_shape_R$R.json 2> $OUT/multi_byte_R$R.err
  VRS_DIST_COPY_OWN=1 VRS_DIST_SHAPE=byte VRS_BENCH_FORCE_MULTI=1 python bench.py --rounds $R --rounds-forced --steps 10 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_bench_multi_world1_byte_shape_copy_own_R$R.json 2> $OUT/multi_byte_copy_R$R.err
done
python bench.py --loopback-ranks 8 --n 2e6 --steps 5 > $OUT/${TAG}_bench_multi_loopback_world8.json 2> $OUT/loop8.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "UNPARSABLE", e); continue
    print(f.split("/")[-1], d["value"], d["unit"], d["ms_per_step"], "roofline", d["roofline"].get("frac"), "sort", d.get("sort_roofline", d.get("step_roofline", {})).get("frac_of_peak"))
PY
