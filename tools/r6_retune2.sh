for i in 1 2 3; do
for arm in "" "--tune fused_wn_bwd=8" "--tune wgrad_tail_splits=1" "--tune fused_wn_fwd_skip=4"; do
  python bench.py --no-cpu-baseline --no-f32-key --windows 4 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- $arm', d['ms_per_step'], d['windows']['ms_per_step_median'], (d.get('fwd_bwd_only') or {}).get('ms_per_step'))"
done; done
for c in 3 5; do for arm in "" "--tune fused_wn_bwd=8"; do python bench.py --config $c --no-cpu-baseline --no-f32-key --windows 4 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c $arm', d['ms_per_step'], d['windows']['ms_per_step_median'])"; done; done
