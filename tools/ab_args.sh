#!/bin/bash
# usage: tools/ab_args.sh "bench args" "ENV=.." "ENV=.." ...
a=$1; shift
for e in "$@"; do
  v=$(env $e timeout 150 python bench.py --no-cpu-baseline --steps 30 --warmup 8 $a 2>/dev/null </dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "[$a] $e: $v"
done
