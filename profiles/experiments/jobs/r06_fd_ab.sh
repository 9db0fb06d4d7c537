#!/bin/bash
# A/B of build variants of the deep kernel: fd_check timing lines only
OUT=gpurun_out/r06b; mkdir -p $OUT
for v in "$@"; do
  DPGO_HIP_LIB=profiles/experiments/build/$v/libdpgo_hip.so timeout 300 python profiles/experiments/fd_check.py 5 5 > $OUT/ab_$v.log 2>&1; echo "== $v"; grep -E "RESULT|k_step_fd|ms/iter" $OUT/ab_$v.log | tail -5
done
