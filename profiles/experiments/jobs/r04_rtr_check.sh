# RTR solve: parity tests, per-iteration time, phase trace (run from the repo root on the GPU box)
python -m pytest tests/test_gpu_rtr_fused.py -m gpu -q 2>&1 | tail -30
echo "=== rtr_run"; python profiles/experiments/rtr_run.py 100
if [ -f profiles/experiments/build/rtrtrace/libdpgo_hip.so ]; then
echo "=== rtr_trace"; DPGO_HIP_LIB=profiles/experiments/build/rtrtrace/libdpgo_hip.so python profiles/experiments/rtr_trace.py 101
fi
