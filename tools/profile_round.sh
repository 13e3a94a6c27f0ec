#!/bin/bash
# The round's evidence in one go: kernel trace of the default bench command, the counter passes of the headline
# configuration (tools/kernel_counters.sh), then the plain bench line.  Everything lands under gpurun_out/<tag>/
# and gpurun_out/counters/; fold the counters into profiles/counters.json with tools/merge_counters.py --into.
# usage: tools/profile_round.sh <tag>
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 $root/bench.py --steps 5 --warmup 1 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/kt_err.txt" || { tail -5 "$out/kt_err.txt"; exit 1; }
cd $root
tools/kernel_counters.sh 3 || exit 1
cd $root && timeout -k 10 300 python3 bench.py > "$out/bench.json" 2> "$out/bench_err.txt" || exit 1
tail -1 "$out/bench.json"
