#!/bin/bash
# usage: tools/ab.sh REPEATS "ENV1=.. ENV2=.." "ENV=.." ...   -> ms/step of bench.py for every environment, interleaved
n=$1; shift
for i in $(seq $n); do
  for e in "$@"; do
    v=$(env $e timeout 150 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null </dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$e: $v"
  done
done
