mkdir -p gpurun_out
python -m pytest tests/test_gpu_benchmarked_sizes.py -x -q -s -k "config5 or config3 or reproducib or conditioned" > gpurun_out/t4.log 2>&1; tail -5 gpurun_out/t4.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_entrypoints.py -x -q > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
python bench.py --config 5 --no-cpu-baseline --windows 2 > gpurun_out/b5.json 2> gpurun_out/b5.err; python -c "
import json;d=json.loads(open('gpurun_out/b5.json').read().strip().splitlines()[-1]);print('config5', d['ms_per_step'], d['windows'])"
python bench.py --config 5 --no-cpu-baseline --windows 0 --timeline > gpurun_out/r06c_config5_tl.json 2> gpurun_out/r06c_config5_timeline.txt; grep timeline gpurun_out/r06c_config5_timeline.txt | tail -12
