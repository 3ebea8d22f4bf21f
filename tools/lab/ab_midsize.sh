# round 6: keys between 2.6e7 and 1e8 -- 16384 buckets in 256-thread workgroups (the default) against 32768 buckets (VRS_TUNE_MSD_POOL_SUB_BITS 7)
mkdir -p gpurun_out/r06f
{
for n in 2.6e7 3e7 4e7 5e7 6e7 8e7; do
python tools/lab/ab_bench.py default $n 12
VRS_SUB_BITS=7 python tools/lab/ab_bench.py sub7 $n 12
done
} > gpurun_out/r06f/ab_midsize.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06f/ab_midsize.txt
