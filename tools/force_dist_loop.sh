#!/bin/bash
# N consecutive `bench.py --force-dist` runs (the --gpus N path with one rank on RCCL) - VERDICT r4 item 1: 30 clean runs.
# Usage: tools/force_dist_loop.sh [N] [logfile]
N=${1:-30}
LOG=${2:-gpurun_out/r05_force_dist_loop.log}
: > "$LOG"
fail=0
for i in $(seq 1 "$N"); do
    t0=$(date +%s.%N)
    out=$(timeout 300 python bench.py --force-dist --steps 3 --warmup 2 --windows 0 --no-cpu-baseline --no-f32-key 2>gpurun_out/.fd_err)
    rc=$?
    t1=$(date +%s.%N)
    ms=$(echo "$out" | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "run $i rc $rc ms_per_step ${ms:-?} wall $(python -c "print(round($t1 - $t0, 1))")s" >> "$LOG"
    if [ $rc -ne 0 ]; then fail=$((fail+1)); tail -5 gpurun_out/.fd_err >> "$LOG"; fi
done
echo "force-dist loop: $((N-fail)) of $N runs clean" >> "$LOG"
tail -1 "$LOG"
