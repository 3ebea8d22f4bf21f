#!/bin/bash
# round-2 lab call 1: new one-call flow -- correctness, group/unroll sweep, kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02_call1.txt
: > $O
timeout 900 python -m pytest tests/test_gpu_one_call.py -x -q -m gpu 2>&1 | tail -15 >> $O
export VRS_ONLY_ONE_READ=1
for g in 32 16 8; do
  VRS_TAG="g$g u8" VRS_GROUPS=$g timeout 300 python tools/one_call_time.py 1e8 10 uniform >> $O 2>&1
done
for lib in u4 u12; do
  for g in 32 8; do
    VRS_TAG="g$g $lib" VRS_GROUPS=$g VRS_LIB=tools/lab/libs/libvrs_$lib.so timeout 300 python tools/one_call_time.py 1e8 10 uniform >> $O 2>&1
  done
done
for g in 32 8; do
  VRS_TAG="g$g u8" VRS_GROUPS=$g timeout 300 python tools/one_call_time.py 1e7 20 uniform >> $O 2>&1
done
unset VRS_ONLY_ONE_READ
VRS_TAG="both" timeout 300 python tools/one_call_time.py 1e8 10 uniform >> $O 2>&1
VRS_TAG="sorted" VRS_ONLY_ONE_READ=1 timeout 300 python tools/one_call_time.py 1e8 5 sorted >> $O 2>&1
VRS_TAG="mult256" VRS_ONLY_ONE_READ=1 timeout 300 python tools/one_call_time.py 1e8 5 mult256 >> $O 2>&1
# kernel trace of a few sorts (start/end timestamps: gaps)
cd /tmp && VRS_ONLY_ONE_READ=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/trace1 -o t -- python $GRAFT_REPO_ROOT/tools/one_call_time.py 1e8 4 uniform > /tmp/trace1.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/trace1 -name "*kernel_trace.csv" | head -1)
echo "trace file: $f" >> $O
python - "$f" >> $O 2>&1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 14 kernels: name, duration, gap to previous end
tail = rows[-16:]
prev_end = None
for r in tail:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{r['Kernel_Name'][:60]:60s} dur={(e-s)/1e3:8.1f}us gap={gap:7.1f}us grid={r.get('Grid_Size_X','?')}")
    prev_end = e
PY
cat $O
