mkdir -p gpurun_out
python -m pytest tests/test_gpu_benchmarked_sizes.py -x -q -k "config5 or config3 or config4 or conditioned or reproducib" > gpurun_out/t3.log 2>&1; tail -4 gpurun_out/t3.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_entrypoints.py -x -q > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
run() { python bench.py --no-cpu-baseline --no-f32-key --windows 4 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err; python -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(' '.join(sys.argv[1:]), d['ms_per_step'], d['windows']['ms_per_step_median'], d['windows']['ms_per_step_min'])" -- "$@"; }
run --config 2
run --config 3
run --config 4
run --config 5
run --config 2
run --config 3
