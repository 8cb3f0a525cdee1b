#!/bin/bash
# A/B builds of the library that differ only in how pt_kernels_extend.hip is compiled: ab/lib_<name>.so (selected with PT_AMD_LIB).
# Usage: tools/build_variant.sh <name> "<extra flags>" [noilp]       (noilp: without -amdgpu-sched-strategy=max-ilp)
set -e
name=$1; extra=$2; ilp="-mllvm -amdgpu-sched-strategy=max-ilp"; [ "$3" = noilp ] && ilp=""
cd "$(dirname "$0")/../path-tracer_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function \
  -I../../include $ilp $extra -c pt_kernels_extend.hip -o ../../ab/ext_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/lib_$name.so pt_api.o pt_api_render.o pt_kernels_build.o pt_kernels_render.o ../../ab/ext_$name.o pt_sort.o pt_comm.o -ldl
echo built ab/lib_$name.so
