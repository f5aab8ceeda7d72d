mkdir -p gpurun_out
bash tools/step_order.sh r06b > gpurun_out/r06b_order.log 2>&1; tail -2 gpurun_out/r06b_order.log
for c in 3 5; do python bench.py --config $c --no-cpu-baseline --no-f32-key --windows 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', d['ms_per_step'], d['windows']['ms_per_step_median'])"; done
