#!/bin/bash
# round 6, the forward -> backward gap: the tests of the fused loss / expansion / duration-projection launches, then alternating bench runs (seeded loss nodes
# on / off) and a timeline
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py tests/test_gpu_model.py tests/test_gpu_encoder.py tests/test_gpu_entrypoints.py -x -q > gpurun_out/gap_tests.log 2>&1; tail -8 gpurun_out/gap_tests.log
for i in 1 2; do
  for arm in "" "--tune seeded=0"; do
    python bench.py --no-cpu-baseline --no-f32-key --windows 4 $arm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('-- $arm', d['ms_per_step'], d['windows']['ms_per_step_median'], (d.get('fwd_bwd_only') or {}).get('ms_per_step'))"
  done
done
python bench.py --no-cpu-baseline --no-f32-key --windows 2 --timeline > gpurun_out/gap_timeline.txt 2>&1; grep -n "dec_fwd_end\|dec_bwd_begin\|enc_fwd_project\|enc_dgrads\|wgrads_done\|launches" gpurun_out/gap_timeline.txt | head -20
