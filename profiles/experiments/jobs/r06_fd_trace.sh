#!/bin/bash
OUT=gpurun_out/r06b; mkdir -p $OUT
DPGO_HIP_LIB=profiles/experiments/build/fd/libdpgo_hip.so timeout 300 python profiles/experiments/fd_trace.py > $OUT/fd_trace.log 2>&1; tail -9 $OUT/fd_trace.log
