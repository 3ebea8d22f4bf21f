#!/bin/bash
# lab builds of the library with extra -D flags: tools/lab/build_variant.sh NAME [-DVRS_X=..]...  -> tools/lab/libs/libvrs_NAME.so
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Ivkradixsort_amd/csrc "$@" \
    vkradixsort_amd/csrc/vrs_contract.hip vkradixsort_amd/csrc/vrs_one_call.hip vkradixsort_amd/csrc/vrs_msd_hybrid.hip vkradixsort_amd/csrc/vrs_msd_pool.hip vkradixsort_amd/csrc/vrs_msd_pool_local.hip vkradixsort_amd/csrc/vrs_pool_shape.hip vkradixsort_amd/csrc/vrs_capi.hip vkradixsort_amd/csrc/vrs_capi_contract.hip vkradixsort_amd/csrc/vrs_capi_sort.hip vkradixsort_amd/csrc/vrs_capi_pool.hip vkradixsort_amd/csrc/vrs_capi_msd.hip vkradixsort_amd/csrc/vrs_dist.hip -ldl -o tools/lab/libs/libvrs_$name.so
echo built tools/lab/libs/libvrs_$name.so "$@"
