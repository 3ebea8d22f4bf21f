#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call3.txt
: > $O
R=$PWD
echo "== lookback lab (contention-free stats)" >> $O
timeout 300 tools/lab/lookback_lab 1e8 32 5 >> $O 2>&1
timeout 300 tools/lab/lookback_lab_b2 1e8 32 5 2>&1 | grep -E "^n=|^A|look-back|^D|longest" >> $O
timeout 300 tools/lab/lookback_lab_b8 1e8 32 5 2>&1 | grep -E "^n=|^A|look-back|^D|longest" >> $O
timeout 300 tools/lab/lookback_lab 1e7 8 10 2>&1 | grep -E "^n=|^A|look-back|^B|lifetime|longest" >> $O
echo "== local sort feasibility" >> $O
timeout 600 tools/lab/local_sort_lab 1e8 >> $O 2>&1
echo "== kernel trace gaps, no events attached" >> $O
for N in 1e8 1e7; do
rm -rf /tmp/trace1; cd /tmp && VRS_NO_PROFILE=1 VRS_ONLY_ONE_READ=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace1 -o t -- python $R/tools/one_call_time.py $N 6 uniform > /tmp/trace1.log 2>&1
cd $R
f=$(find /tmp/trace1 -name "*kernel_trace.csv" | head -1)
python - "$f" $N >> $O 2>&1 <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "copyBuffer" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-13:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{sys.argv[2]} {r['Kernel_Name'][:60]:60s} dur={(e-s)/1e3:8.1f}us gap={gap:7.1f}us")
    prev_end = e
PY
done
cat $O
