mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.log 2>&1; tail -15 gpurun_out/gpu_suite.log
