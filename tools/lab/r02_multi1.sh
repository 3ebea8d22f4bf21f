#!/bin/bash
# world-size-1 run of the multi-GPU step (partition pass + exchange with itself + sub-range sorts): R = 1, 2, 4
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for R in 1 2 4; do
  VRS_BENCH_FORCE_MULTI=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 \
    timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --rounds $R --rounds-forced --no-cpu-baseline > gpurun_out/r02_multi1_R$R.json 2> gpurun_out/r02_multi1_R$R.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r02_multi1_R$R.json"))
print("R=$R", d["value"], d["ms_per_step"], d.get("roofline", {}).get("avg_launch_us"), {k: v for k, v in d.items() if k in ("phase_ms", "phases", "breakdown")})
PY
done
