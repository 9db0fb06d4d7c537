#!/bin/bash
OUT=gpurun_out/r06b; mkdir -p $OUT
for v in "$@"; do
  DPGO_HIP_LIB=profiles/experiments/build/$v/libdpgo_hip.so timeout 300 python profiles/experiments/fd_trace.py > $OUT/trace_$v.log 2>&1; echo "== $v"; tail -9 $OUT/trace_$v.log | grep -E "^wave [0467]|workgroups|k_step_fd"
done
