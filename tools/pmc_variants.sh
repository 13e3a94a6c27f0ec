#!/bin/bash
# Collects instruction-mix counters for each scan variant of the f32 kernel (one rocprofv3 pass per variant).
# usage: tools/pmc_variants.sh <outdir> "<counters>" <variants...>
set -e
out=$1; shift
ctrs=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out" -o v$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --spp 100 --variant $v --no-cpu-baseline > "$out/bench_v$v.json" 2> "$out/err_v$v.txt"
done
