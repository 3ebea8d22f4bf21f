export TMPDIR=/tmp
# usage: tools/lab/lds_pmc.sh [TAG]   (VRS_PMC_CMD: the command to profile instead of the bench, e.g. a lab build through tools/lab/ab_bench.py)
TAG=${1:-lds_pmc}
OUT=$PWD/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
CMD=${VRS_PMC_CMD:-"python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs"}
for set in "SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o b -- $CMD > /dev/null 2> $OUT/$tag.log || echo "failed $set"
done
python - $OUT <<'PY'
import csv, glob, collections, sys
for f in sorted(glob.glob(sys.argv[1] + '/*/*counter_collection.csv')+glob.glob(sys.argv[1] + '/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        import re
        m=re.search(r'(\w+_kernel(?:<[^>(]*>)?)', r['Kernel_Name'])  # (names in an anonymous namespace carry a '(' before the kernel's own)
        k=m.group(1) if m else r['Kernel_Name'][:50]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'pool_local' in k or 'pass_b' in k or 'pass_a' in k or 'lean' in k or 'msd_local' in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, len(list(v.values())[0]))
PY
