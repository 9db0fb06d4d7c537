#!/bin/bash
OUT=gpurun_out/r06c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s > $OUT/gpu_tests.log 2>&1; tail -15 $OUT/gpu_tests.log | cut -c1-200
grep "^seed" $OUT/gpu_tests.log
