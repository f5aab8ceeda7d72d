# round 6: the final tree on one more box - the GPU suite and three default bench lines (box-to-box variance record)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite2.log 2>&1; tail -1 gpurun_out/gpu_suite2.log
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-f32-key 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 2', d['ms_per_step'], d['windows'], 'fwd_bwd_only', d['fwd_bwd_only']['ms_per_step'], 'wn_fwd alone', d['roofline']['us_per_launch'], 'clock', d['roofline']['sustained_mfma_clock_ghz'])"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
