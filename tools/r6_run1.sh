set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_encoder.py -x -q -k "prosody" > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
python -m pytest tests/test_gpu_modes.py tests/test_conditioning_encoders.py -x -q > gpurun_out/t3.log 2>&1; tail -5 gpurun_out/t3.log
python -m pytest tests/test_gpu_benchmarked_sizes.py -x -q -s -k "config5 or long_form or conditioned" > gpurun_out/t4.log 2>&1; tail -15 gpurun_out/t4.log
python bench.py --config 5 --no-cpu-baseline --windows 2 > gpurun_out/b5.json 2> gpurun_out/b5.err; tail -c 1500 gpurun_out/b5.json
python bench.py --config 2 --no-f32-key --windows 2 > gpurun_out/b2.json 2> gpurun_out/b2.err; tail -c 3000 gpurun_out/b2.json
