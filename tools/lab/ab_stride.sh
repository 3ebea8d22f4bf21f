# round 6: slack regions of ONE size (bucket b at b * stride) -- the local sort asks for its keys before it knows anything about its bucket
mkdir -p gpurun_out/r06e
{
for rep in 1 2 3; do
VRS_LIB=tools/lab/libs/libvrs_clean.so python tools/lab/ab_bench.py base 1e8 12
VRS_LIB=tools/lab/libs/libvrs_stride.so VRS_LAB_POOL_STRIDE=1 python tools/lab/ab_bench.py stride 1e8 12
done
} > gpurun_out/r06e/ab_stride.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06e/ab_stride.txt
