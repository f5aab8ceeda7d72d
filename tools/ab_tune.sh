#!/bin/bash
# A/B of decoder.TUNE switches on ONE box: alternating runs of bench.py (whole Train_Step, config 2) with and without the given --tune entries.
#   usage (GPU box): tools/ab_tune.sh "prep_fused=0" [repeats]     -> prints ms/step of A (default) and B (with the entries), alternating
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="$1"; N=${2:-2}
targs=""; for kv in $B; do targs="$targs --tune $kv"; done
for i in $(seq 1 $N); do
  for arm in A B; do
    if [ $arm = A ]; then extra=""; else extra="$targs"; fi
    python $REPO/bench.py --no-cpu-baseline --no-f32-key --windows 4 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', '$extra', 'ms/step', d['ms_per_step'], 'median', d.get('ms_per_step_median'), 'fwd_bwd_only', (d.get('fwd_bwd_only') or {}).get('ms_per_step'))"
  done
done
