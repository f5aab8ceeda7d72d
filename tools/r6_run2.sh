set -x
mkdir -p gpurun_out
python tests/longform_check.py > gpurun_out/longform.log 2>&1; tail -12 gpurun_out/longform.log
bash tools/profile_step.sh r06a_config5 --config 5 > gpurun_out/r06a_config5_prof.log 2>&1; tail -20 gpurun_out/r06a_config5_prof.log
ORDER_ARGS="--config 5" bash tools/step_order.sh r06a_config5 > gpurun_out/r06a_config5_order.log 2>&1; tail -3 gpurun_out/r06a_config5_order.log
python bench.py --config 5 --no-cpu-baseline --windows 0 --timeline > gpurun_out/r06a_config5_tl.json 2> gpurun_out/r06a_config5_timeline.txt; grep timeline gpurun_out/r06a_config5_timeline.txt
