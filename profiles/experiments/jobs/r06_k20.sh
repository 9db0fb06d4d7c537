#!/bin/bash
OUT=gpurun_out/r06d; mkdir -p $OUT
python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_schedule.py tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log | cut -c1-200
python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06d/bench_k20.json'))
print('K=20: value', d['value'], 'k_region', d['ms_per_step_k_region'], 'mean15', d.get('ms_per_step_k_region_mean_of_15'))
PY
