mkdir -p gpurun_out
run() { python bench.py --no-cpu-baseline --no-f32-key --windows 2 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err; python -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(' '.join(sys.argv[1:]), d['ms_per_step'], d['windows']['ms_per_step_median'], d['windows']['ms_per_step_min'])" -- "$@"; }
run --config 3
run --config 3 --tune tail_aside=0
run --config 3 --tune prior_loss=0
run --config 3 --tune prior_loss=0 --tune tail_aside=0
run --config 2 --tune prior_loss=0
run --config 2
python -X faulthandler -m pytest tests/test_gpu_model.py -x -q > gpurun_out/t2.log 2>&1; grep -n "Fatal\|Segmentation\|File \"/root\|passed\|failed" gpurun_out/t2.log | head -30
