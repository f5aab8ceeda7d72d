mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoder.py -x -q -k "prosody" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
run() { python bench.py --no-cpu-baseline --no-f32-key --windows 2 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err; python -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(' '.join(sys.argv[1:]), d['ms_per_step'], d['windows']['ms_per_step_median'], d['windows']['ms_per_step_min'])" -- "$@"; }
run --config 5
run --config 2
run --config 5
ORDER_ARGS="--config 5" bash tools/step_order.sh r06j_config5 > gpurun_out/r06j_config5_order.log 2>&1; tail -2 gpurun_out/r06j_config5_order.log
grep -n "c2d_wgrad\|c2d_first" gpurun_out/r06j_config5_step_order.txt | cut -c1-130
