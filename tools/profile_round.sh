#!/bin/bash
# The round's evidence in one go: kernel trace of the default bench, HBM traffic passes, counter passes for the
# grid and the hybrid kernels, then the plain bench line.  Everything lands under gpurun_out/<tag>/.
# usage: tools/profile_round.sh <tag>
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/kt_err.txt" || { tail -5 "$out/kt_err.txt"; exit 1; }
cd $root
tools/ab_traffic.sh $tag || exit 1
tools/pmc_final.sh "$out/pmc" 3 > "$out/pmc3.txt" || exit 1
tools/pmc_final.sh "$out/pmc" 2 > "$out/pmc2.txt" || exit 1
cd $root && timeout -k 10 300 python3 bench.py > "$out/bench.json" 2> "$out/bench_err.txt" || exit 1
tail -1 "$out/bench.json"
