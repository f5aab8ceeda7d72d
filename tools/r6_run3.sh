set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoder.py -x -q -k "prosody" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
python -m pytest tests/test_gpu_benchmarked_sizes.py -x -q -s -k "config5 or config3 or reproducib" > gpurun_out/t4.log 2>&1; tail -5 gpurun_out/t4.log
python -m pytest tests/test_gpu_model.py -x -q > gpurun_out/t5.log 2>&1; tail -3 gpurun_out/t5.log
python tests/longform_check.py > gpurun_out/longform.log 2>&1; grep "BF16\|OK\|Error" gpurun_out/longform.log
for t in 1 0; do python bench.py --config 5 --no-cpu-baseline --windows 2 --tune tail_aside=$t > gpurun_out/b5_$t.json 2> gpurun_out/b5_$t.err; python -c "
import json;d=json.loads(open('gpurun_out/b5_$t.json').read().strip().splitlines()[-1]);print('config5 tail_aside=$t', d['ms_per_step'], d['windows'])"; done
for t in 1 0; do python bench.py --config 3 --no-cpu-baseline --windows 2 --tune tail_aside=$t > gpurun_out/b3_$t.json 2> gpurun_out/b3_$t.err; python -c "
import json;d=json.loads(open('gpurun_out/b3_$t.json').read().strip().splitlines()[-1]);print('config3 tail_aside=$t', d['ms_per_step'], d['windows'])"; done
python bench.py --config 2 --no-cpu-baseline --no-f32-key --windows 2 > gpurun_out/b2.json 2> gpurun_out/b2.err; python -c "
import json;d=json.loads(open('gpurun_out/b2.json').read().strip().splitlines()[-1]);print('config2', d['ms_per_step'], d['windows'])"
ORDER_ARGS="--config 5" bash tools/step_order.sh r06b_config5 > gpurun_out/r06b_config5_order.log 2>&1; tail -3 gpurun_out/r06b_config5_order.log
grep "c2d" gpurun_out/r06b_config5_step_order.txt | cut -c1-120
