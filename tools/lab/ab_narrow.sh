# round 6: the 8 + 8 cut with 16-bit slack slots (20 B/key) against the default 7 + 7 cut (24 B/key) and the 8 + 8 cut with 32-bit slots
mkdir -p gpurun_out/r06d
{
for rep in 1 2; do
python tools/lab/ab_bench.py base7+7 1e8 12
VRS_LAB_POOL_WIDE=1 VRS_TOP_BITS=8 VRS_SUB_BITS=8 python tools/lab/ab_bench.py cut8+8wide 1e8 12
VRS_TOP_BITS=8 VRS_SUB_BITS=8 python tools/lab/ab_bench.py cut8+8narrow 1e8 12
done
for n in 1e7 3e7 6e7; do
python tools/lab/ab_bench.py base $n 12
VRS_TOP_BITS=8 VRS_SUB_BITS=8 python tools/lab/ab_bench.py narrow $n 12
done
} > gpurun_out/r06d/ab_narrow.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06d/ab_narrow.txt | cut -c1-400
