python -m pytest tests/test_gpu_wavenet_fused.py tests/test_gpu_conv.py tests/test_gpu_benchmarked_sizes.py -x -q 2>&1 | grep -v Warn | tail -6
for i in 1 2 3; do
for arm in "" "--tune wgrad_balance=0"; do
  python bench.py --no-cpu-baseline --no-f32-key --windows 4 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- $arm', d['ms_per_step'], d['windows']['ms_per_step_median'], (d.get('fwd_bwd_only') or {}).get('ms_per_step'))"
done; done
python bench.py --no-cpu-baseline --no-f32-key --windows 2 --timeline 2>&1 | grep timeline | tail -8 | cut -c1-80
