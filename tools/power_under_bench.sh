#!/bin/bash
# Package power and shader clock while bench.py replays the training step (rocm-smi sampled beside it).
#   usage (GPU box): tools/power_under_bench.sh [bench args]      e.g. --tune fused_wn_bwd=12
REPO=${GRAFT_REPO_ROOT:-/root/repo}
python $REPO/bench.py --no-f32-key --no-cpu-baseline --steps 300 --windows 6 "$@" > /tmp/pub.json 2>/dev/null &
PID=$!
sleep 14
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | sed 's/.*sclk clock level: [0-9]*: //; s/.*Power (W): /W /' | tr '\n' ' '; echo
  sleep 0.4
done
wait $PID
tail -1 /tmp/pub.json | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('ms/step', d['ms_per_step'], d['windows'])"
