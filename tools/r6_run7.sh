mkdir -p gpurun_out
run() { python bench.py --config 2 --no-cpu-baseline --no-f32-key --windows 4 "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err; python -c "
import json,sys;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(' '.join(sys.argv[1:]), d['ms_per_step'], d['windows']['ms_per_step_median'], d['windows']['ms_per_step_min'])" -- "$@"; }
for rep in 1 2; do
run
run --tune fused_wn_bwd=9
run --tune fused_wn_bwd=10
run --tune fused_wn_bwd=12
run --tune fused_wn_fwd_skip=2
run --tune fused_wn_fwd_skip=1
run --tune fused_wn_fwd_skip=2 --tune fused_wn_bwd=9
done
