#!/bin/bash
# lab build that differs from the product library in ONE translation unit: tools/lab/build_local_variant.sh NAME FILE.hip [-DVRS_X=..]...
# (the other objects come from vkradixsort_amd/_build, i.e. build the product library first)  -> tools/lab/libs/libvrs_NAME.so
set -e
cd "$(dirname "$0")/../.."
name=$1; unit=$2; shift; shift
mkdir -p tools/lab/libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Iinclude -Ivkradixsort_amd/csrc "$@" vkradixsort_amd/csrc/$unit.hip -o tools/lab/libs/${unit}_$name.o
objs=$(ls vkradixsort_amd/_build/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs tools/lab/libs/${unit}_$name.o -ldl -o tools/lab/libs/libvrs_$name.so
rm -f tools/lab/libs/${unit}_$name.o
echo built tools/lab/libs/libvrs_$name.so "$@"
