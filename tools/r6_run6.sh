mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_conv.py -x -q > gpurun_out/t1.log 2>&1; tail -4 gpurun_out/t1.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q > gpurun_out/t2.log 2>&1; tail -4 gpurun_out/t2.log
python -m pytest tests/test_gpu_benchmarked_sizes.py -x -q -k "config2 or config5 or config3" > gpurun_out/t3.log 2>&1; tail -4 gpurun_out/t3.log
python bench.py --config 2 --no-cpu-baseline --no-f32-key --windows 4 > gpurun_out/b2.json 2> gpurun_out/b2.err; python -c "
import json;d=json.loads(open('gpurun_out/b2.json').read().strip().splitlines()[-1]);print('config2', d['ms_per_step'], d['windows'], d.get('fwd_bwd_only'))"
python bench.py --config 2 --no-cpu-baseline --no-f32-key --windows 0 --timeline > gpurun_out/r06d_config2_tl.json 2> gpurun_out/r06d_config2_timeline.txt; grep timeline gpurun_out/r06d_config2_timeline.txt
