for c in 2 3 4 5; do python bench.py --config $c --no-cpu-baseline --no-f32-key --windows 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', d['ms_per_step'], d['windows']['ms_per_step_median'])"; done
python -m pytest tests/test_gpu_model.py tests/test_gpu_decoder.py tests/test_gpu_benchmarked_sizes.py -x -q 2>&1 | tail -2
