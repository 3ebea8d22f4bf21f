#!/bin/bash
# round-2 lab call 2: look-back lab, kernel gaps, digit_tables counters, small-N crossover, tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02_call2.txt
: > $O
echo "== pytest one_call + cpp examples" >> $O
timeout 900 python -m pytest tests/test_gpu_one_call.py "tests/test_gpu_parity.py::test_cpp_host_examples_run_and_verify" -x -q -m gpu 2>&1 | tail -8 >> $O
echo "== lookback lab" >> $O
timeout 300 tools/lab/lookback_lab 1e8 32 5 >> $O 2>&1
timeout 300 tools/lab/lookback_lab_b2 1e8 32 5 2>&1 | grep -E "^n=|^A|look-back|^D" >> $O
timeout 300 tools/lab/lookback_lab_b8 1e8 32 5 2>&1 | grep -E "^n=|^A|look-back|^D" >> $O
timeout 300 tools/lab/lookback_lab 1e8 8 5 2>&1 | grep -E "^n=|^A|look-back" >> $O
timeout 300 tools/lab/lookback_lab 1e7 8 10 2>&1 | grep -E "^n=|^A|look-back|^B|lifetime" >> $O
echo "== kernel trace gaps" >> $O
R=$PWD
cd /tmp && VRS_ONLY_ONE_READ=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace1 -o t -- python $R/tools/one_call_time.py 1e8 4 uniform > /tmp/trace1.log 2>&1
cd $R
f=$(find /tmp/trace1 -name "*kernel_trace.csv" | head -1)
python - "$f" >> $O 2>&1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-14:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} dur={(e-s)/1e3:8.1f}us gap={gap:7.1f}us")
    prev_end = e
PY
cd /tmp && VRS_ONLY_ONE_READ=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace2 -o t -- python $R/tools/one_call_time.py 1e7 6 uniform > /tmp/trace2.log 2>&1
cd $R
f=$(find /tmp/trace2 -name "*kernel_trace.csv" | head -1)
python - "$f" >> $O 2>&1 <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev_end = None
for r in rows[-14:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"1e7 {r['Kernel_Name'][:66]:66s} dur={(e-s)/1e3:8.1f}us gap={gap:7.1f}us")
    prev_end = e
PY
echo "== digit_tables counters" >> $O
for cset in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc; cd /tmp
  VRS_ONLY_ONE_READ=1 timeout 300 rocprofv3 --pmc $cset --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $R/tools/one_call_time.py 1e8 1 uniform > /tmp/pmc.log 2>&1
  cd $R
  python - >> $O 2>&1 <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "digit_tables" if "digit_tables" in r["Kernel_Name"] else "onesweep" if "onesweep" in r["Kernel_Name"] else None
        if k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k in acc:
    n = max(len(disp[k]), 1)
    print(k, "per launch:", {c: round(v / n) for c, v in sorted(acc[k].items())})
PY
done
echo "== small N crossover" >> $O
timeout 600 python tools/small_n_sweep.py gpurun_out/r02_small_n_crossover.csv >> $O 2>&1
cat $O
