mkdir -p gpurun_out/r06c
VRS_PMC_CMD="python tools/lab/pairs_ab.py packed 1e8 3" bash tools/lab/lds_pmc.sh pmc_packed > gpurun_out/r06c/pmc_packed.txt 2>&1
VRS_PMC_CMD="env VRS_LIB=tools/lab/libs/libvrs_pairs_unpacked.so python tools/lab/pairs_ab.py unpacked 1e8 3" bash tools/lab/lds_pmc.sh pmc_unpacked > gpurun_out/r06c/pmc_unpacked.txt 2>&1
rm -rf gpurun_out/pmc_packed gpurun_out/pmc_unpacked
grep -h local_sort gpurun_out/r06c/pmc_packed.txt gpurun_out/r06c/pmc_unpacked.txt
