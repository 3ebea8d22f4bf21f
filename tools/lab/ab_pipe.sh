# round 6, verdict item 1(c): the pipelined local sort (tools/lab/patches/r06_pipelined_local_sort.patch) against the product's, LDS counters of both
mkdir -p gpurun_out/r06b
VRS_PMC_CMD="env VRS_LIB=tools/lab/libs/libvrs_free4.so python tools/lab/ab_bench.py free4 1e8 6" bash tools/lab/lds_pmc.sh pmc_free4 > gpurun_out/r06b/pmc_free4.txt 2>&1
VRS_PMC_CMD="env VRS_LIB=tools/lab/libs/libvrs_rank3.so VRS_LAB_LOCAL_PIPE=768 python tools/lab/ab_bench.py rank3 1e8 6" bash tools/lab/lds_pmc.sh pmc_rank3 > gpurun_out/r06b/pmc_rank3.txt 2>&1
VRS_PMC_CMD="python tools/lab/ab_bench.py base 1e8 6" bash tools/lab/lds_pmc.sh pmc_base > gpurun_out/r06b/pmc_base.txt 2>&1
rm -rf gpurun_out/pmc_free4 gpurun_out/pmc_rank3 gpurun_out/pmc_base
tail -5 gpurun_out/r06b/pmc_*.txt
