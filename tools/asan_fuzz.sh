#!/bin/bash
# The GPU fuzz, the soak and the multi-rank tests with the library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer
# (device code unchanged): the asynchronous settle, buffer release, the ticket ring and the loopback hub's threads are where a stale
# pointer would hide.  Builds the instrumented library on this box (vkradixsort_amd/_build does not travel), preloads the sanitizer
# runtime into python.   tools/asan_fuzz.sh [seconds of fuzz] -> gpurun_out/asan_fuzz.txt
set -u
cd "$(dirname "$0")/.."
SECS=${1:-120}
OUT=gpurun_out/asan_fuzz.txt; mkdir -p gpurun_out; : > $OUT
python - <<'PY' >> $OUT 2>&1
from vkradixsort_amd import build as b
b.build_sanitized("asan")
print("built", b.san_dir("asan") / "libvkradixsort_amd.so")
PY
RT=$(python -c "from vkradixsort_amd import build as b; print(b.sanitizer_runtime('asan'))")
export VRS_LIB=$PWD/vkradixsort_amd/_build/san_asan/libvkradixsort_amd.so
# (max_malloc_fill_size=0: the HSA runtime -- not instrumented -- reads heap words it never wrote when the process exits, and the 0xbe
#  pattern ASan fills fresh memory with makes that a wild pointer; allocator_may_return_null: ROCm's ASan runtime also intercepts
#  hsa_amd_memory_pool_allocate, an allocation it cannot serve must fail the hipMalloc, not abort the process)
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:halt_on_error=1:abort_on_error=0:protect_shadow_gap=0:max_malloc_fill_size=0:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
run() { echo "== $*" >> $OUT; LD_PRELOAD=$RT timeout 900 "$@" >> $OUT 2>&1; echo "== exit $?" >> $OUT; }
run python tools/fuzz_gpu.py $SECS 4242
run python tools/fuzz_gpu.py 60 999   # (tools/soak_one_call.py verifies with torch, whose allocations the preloaded runtime cannot serve)
run python tools/lab/pool_soak.py 300 3e7
run python -m pytest tests/test_gpu_dist.py tests/test_gpu_one_call.py tests/test_gpu_pool.py -q -k "async or settle or release or loopback or ranks or ticket or unbalanced or over_capacity or no_room or stale or kept"
grep -c "ERROR: AddressSanitizer\|runtime error:" $OUT | sed 's/^/sanitizer reports: /' >> $OUT
tail -5 $OUT
