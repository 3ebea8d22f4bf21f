# timing only: the pool form's local sort with its LDS passes left out (reads every bucket from its slack region, writes it to its place)
mkdir -p gpurun_out/r06c
{
for rep in 1 2; do
python tools/lab/ab_bench.py base 1e8 12
VRS_LIB=tools/lab/libs/libvrs_nosort.so python tools/lab/ab_bench.py nosort 1e8 12
done
} > gpurun_out/r06c/ab_nosort.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" gpurun_out/r06c/ab_nosort.txt
