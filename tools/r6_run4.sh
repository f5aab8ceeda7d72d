mkdir -p gpurun_out
python bench.py --config 5 --no-cpu-baseline --windows 0 --timeline > gpurun_out/r06b_config5_tl.json 2> gpurun_out/r06b_config5_timeline.txt; grep timeline gpurun_out/r06b_config5_timeline.txt
python bench.py --config 2 --no-cpu-baseline --no-f32-key --windows 0 --timeline > gpurun_out/r06b_config2_tl.json 2> gpurun_out/r06b_config2_timeline.txt; grep timeline gpurun_out/r06b_config2_timeline.txt
