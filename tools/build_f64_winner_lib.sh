#!/bin/bash
# Experimental build: the f32 kernel re-evaluates the winning narrow sphere in f64 (tools/ab_libs/f64winner.so).
set -e
cd "$(dirname "$0")/../distributed-raytracer_amd/csrc"
mkdir -p ../../tools/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDRAY_F64_WINNER -c render_f32.hip -o /tmp/render_f32_f64winner.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ab_libs/f64winner.so dray_api.o render_ref.o /tmp/render_f32_f64winner.o
