#!/bin/bash
# One GPU call's worth of checks, each step under its own time limit; stops at the first step that fails.
# usage: tools/gpu_check.sh <tag> [pytest -k expression]
tag=$1; kexpr=$2
out=gpurun_out/$tag
mkdir -p $out
if [ -n "$kexpr" ]; then
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -k "$kexpr" > $out/tests.log 2>&1
else
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $out/tests.log 2>&1
fi
rc=$?; tail -4 $out/tests.log; if [ $rc -ne 0 ]; then echo "tests rc=$rc"; exit $rc; fi
if [ -f tests/wip_no_alloc.py ]; then
  timeout -k 10 200 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > $out/noalloc.log 2>&1
  rc=$?; tail -2 $out/noalloc.log; if [ $rc -ne 0 ]; then echo "noalloc rc=$rc"; exit $rc; fi
fi
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err
rc=$?; tail -c 2500 $out/bench.json; if [ $rc -ne 0 ]; then tail -5 $out/bench.err; exit $rc; fi
