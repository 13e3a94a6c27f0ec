#!/bin/bash
# Everything the round's records need, on the sources as they are: counters for the four GPU configurations, kernel
# traces, bench lines, the compaction on/off pair, the one-call timings.  usage: tools/final_round.sh
root=$GRAFT_REPO_ROOT
cd $root
tools/profile_round.sh final || exit 1
tools/config_records.sh 2 4 5 || exit 1
timeout -k 10 200 python3 tools/context_timing.py > gpurun_out/final/context_timing.json || exit 1
DRAY_HIP_LIB=tools/ab_libs/sections.so timeout -k 10 200 python3 tools/section_counts.py --config 3 > gpurun_out/final/sections_c3.json || exit 1
echo done
