#!/bin/bash
# usage: tl.sh <name> <lib or ""> : kernel trace of a short bench, timeline of the last kernels
name=$1; lib=$2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/tl_$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
if [ -n "$lib" ]; then export DRAY_HIP_LIB=$root/$lib; fi
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 $root/bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extra-legs > $out/bench.json 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
python3 $root/tools/trace_timeline.py $(ls $out/*kernel_trace.csv | head -1) 24 > $out/timeline.txt
tail -16 $out/timeline.txt
python3 -c "import json;b=json.loads(open('$out/bench.json').read().strip().splitlines()[-1]);print('ms_per_step',b['ms_per_step'])"
