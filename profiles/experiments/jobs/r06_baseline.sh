#!/bin/bash
# round 6, first call: the chunk-ordered stream (fe_ord) against the bitwise tests, dispatch-to-dispatch of k_step_fe, and
# whether device-resident kernel arguments change it
OUT=gpurun_out/r06a; mkdir -p $OUT
python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_parity.py -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
python profiles/experiments/fe_time.py > $OUT/fe_time.log 2>&1; cat $OUT/fe_time.log
HIP_FORCE_DEV_KERNARG=1 python profiles/experiments/fe_time.py > $OUT/fe_time_devkernarg.log 2>&1; cat $OUT/fe_time_devkernarg.log
HIP_FORCE_DEV_KERNARG=0 python profiles/experiments/fe_time.py > $OUT/fe_time_hostkernarg.log 2>&1; cat $OUT/fe_time_hostkernarg.log
