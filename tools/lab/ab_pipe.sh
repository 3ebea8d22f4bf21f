# round 6: the pipelined local sort with the next bucket's keys asked for BETWEEN this bucket's two passes (a thread holds nothing there)
mkdir -p gpurun_out/r06b
{
for rep in 1 2; do
VRS_LIB=tools/lab/libs/libvrs_pipe.so VRS_LAB_LOCAL_PIPE=0 python tools/lab/ab_bench.py base 1e8 12
VRS_LIB=tools/lab/libs/libvrs_pipe.so python tools/lab/ab_bench.py free_top 1e8 12
VRS_LIB=tools/lab/libs/libvrs_freemid.so python tools/lab/ab_bench.py free_mid 1e8 12
VRS_LIB=tools/lab/libs/libvrs_rankmid.so python tools/lab/ab_bench.py rank_mid 1e8 12
VRS_LIB=tools/lab/libs/libvrs_rankmid3.so VRS_LAB_LOCAL_PIPE=768 python tools/lab/ab_bench.py rank_mid3 1e8 12
done
} > gpurun_out/r06b/ab_mid.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06b/ab_mid.txt
