#!/bin/bash
# usage (GPU box): tools/ab_step.sh REPEATS "ENV1=.." "ENV2=.." ...  -> ms/step (median of bench.py's windows) on the TOOLS build, interleaved
REPO=${GRAFT_REPO_ROOT:-/root/repo}
make -C $REPO/glow_tts_amd/csrc tools > /tmp/tools_build.log 2>&1 || { tail -20 /tmp/tools_build.log; exit 1; }
n=$1; shift
for i in $(seq $n); do
  for e in "$@"; do
    v=$(env $e GLOWTTS_LIB_PATH=$REPO/tools/_build/libglowtts_hip_tools.so timeout 200 python $REPO/bench.py --no-cpu-baseline --windows 4 $BENCH_ARGS 2>/dev/null </dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['windows']['ms_per_step_median'], d['windows']['ms_per_step_min'])")
    echo "[$e] $v"
  done
done
