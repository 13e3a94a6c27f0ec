#!/bin/bash
# A build of the library with extra compiler flags on one kernel file, for A/B timing through DRAY_HIP_LIB (never
# the shipped library).  usage: tools/build_variant_lib.sh <name> <render_ref|render_f64|render_f32> "<flags>"
set -e
name=$1; unit=$2; flags=$3
cd "$(dirname "$0")/../distributed-raytracer_amd/csrc"
mkdir -p ../../tools/ab_libs
exact="-ffp-contract=off"; [ "$unit" = render_f32 ] && exact="-mllvm -phi-node-folding-threshold=0"   # (the Makefile's flags per file)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $exact $flags -c $unit.hip -o /tmp/variant_$name.o
others=""
for u in dray_api render_ref render_f64 render_f32; do [ $u = $unit ] || others="$others $u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab_libs/$name.so $others /tmp/variant_$name.o
echo built tools/ab_libs/$name.so
