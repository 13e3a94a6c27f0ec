#!/bin/bash
# Diagnostic build of the library with the main-loop section counters (never the shipped library).
set -e
cd "$(dirname "$0")/../distributed-raytracer_amd/csrc"
mkdir -p ../../tools/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -phi-node-folding-threshold=0 -DDRAY_SECTION_COUNTERS -c render_f32.hip -o /tmp/render_f32_sections.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab_libs/sections.so dray_api.o render_ref.o render_f64.o /tmp/render_f32_sections.o
