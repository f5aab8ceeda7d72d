# Round-6 evidence run (GPU box): test suite, profiles, PMC passes, bench lines.  Everything lands in gpurun_out/ and is copied to profiles/ by the builder.
mkdir -p gpurun_out profiles
python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1; tail -3 gpurun_out/r06_gpu_tests.log
bash tools/profile_step.sh r06_step > gpurun_out/r06_step_prof.log 2>&1; tail -4 gpurun_out/r06_step_prof.log | cut -c1-200
cp gpurun_out/r06_step_kernel_stats.csv profiles/r06_step_kernel_stats.csv
bash tools/profile_step.sh r06_config5 --config 5 > gpurun_out/r06_config5_prof.log 2>&1
bash tools/step_order.sh r06 > gpurun_out/r06_order.log 2>&1; tail -1 gpurun_out/r06_order.log
ORDER_ARGS="--config 5" bash tools/step_order.sh r06_config5 > gpurun_out/r06_config5_order.log 2>&1
bash tools/pmc_conv.sh gpurun_out/r06_conv_pmc.json > gpurun_out/r06_pmc_conv.log 2>&1; tail -3 gpurun_out/r06_pmc_conv.log
cp gpurun_out/r06_conv_pmc.json profiles/r06_conv_pmc.json
bash tools/pmc_mfma.sh gpurun_out/r06_conv_mfma_pmc.json > gpurun_out/r06_pmc_mfma.log 2>&1; tail -3 gpurun_out/r06_pmc_mfma.log
python bench.py --timeline --no-cpu-baseline --no-f32-key --windows 0 > gpurun_out/r06_tl2.json 2> gpurun_out/r06_step_timeline.txt
python bench.py --config 5 --timeline --no-cpu-baseline --windows 0 > gpurun_out/r06_tl5.json 2> gpurun_out/r06_config5_timeline.txt
python bench.py > gpurun_out/r06_bench_config2.json 2> gpurun_out/r06_bench_config2.err; tail -c 600 gpurun_out/r06_bench_config2.json
for c in 3 4 5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r06_bench_config$c.json 2> gpurun_out/r06_bench_config$c.err; done
python bench.py --batch 16 --no-cpu-baseline --no-f32-key > gpurun_out/r06_bench_b16.json 2>/dev/null
python bench.py --batch 64 --no-cpu-baseline --no-f32-key > gpurun_out/r06_bench_b64.json 2>/dev/null
python bench.py --ragged --no-cpu-baseline --no-f32-key > gpurun_out/r06_bench_ragged.json 2>/dev/null
python bench.py --force-dist --no-cpu-baseline --no-f32-key > gpurun_out/r06_bench_rccl_1rank.json 2> gpurun_out/r06_bench_rccl_1rank.err
python bench.py --force-dist --no-overlap --no-cpu-baseline --no-f32-key > gpurun_out/r06_bench_rccl_1rank_no_overlap.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["windows"]["ms_per_step_median"], d.get("fwd_bwd_only", {}).get("ms_per_step"), d.get("distributed"))
    except Exception as e:
        print(f, "ERR", e)
PY
