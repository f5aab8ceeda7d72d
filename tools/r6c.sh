mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; tail -3 gpurun_out/gpu_suite.log
for i in 1 2; do python bench.py --no-cpu-baseline --no-f32-key --windows 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('--', d['ms_per_step'], d['windows']['ms_per_step_median'], (d.get('fwd_bwd_only') or {}).get('ms_per_step'))"; done
bash tools/step_order.sh r06b > gpurun_out/r06b_order.log 2>&1; tail -1 gpurun_out/r06b_order.log; grep -c "at::native\|rocclr" gpurun_out/r06b_step_order.txt
