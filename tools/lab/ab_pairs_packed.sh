# round 6: the pairs' local sort on packed words (key's low bits | place; payloads staged once, in the words' own LDS array: 3 workgroups per CU)
# against the one that carries the payloads through both passes (2 per CU) -- by size (VRS_PACKED -> VRS_TUNE_MSD_POOL_PAIRS_PACKED)
mkdir -p gpurun_out/r06c
{
for n in 2.6e7 4e7 6e7 8e7 9e7 1e8 1.5e8 2e8; do
VRS_PACKED=0 python tools/lab/pairs_ab.py never $n 6
VRS_PACKED=1 VRS_CHECK=1 python tools/lab/pairs_ab.py always $n 6
python tools/lab/pairs_ab.py default $n 6
done
} > gpurun_out/r06c/ab_pairs_sizes.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06c/ab_pairs_sizes.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_pool.py -x -q -k "pairs" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5
