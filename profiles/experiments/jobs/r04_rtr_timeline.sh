cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_linesearch.py tests/test_gpu_tunnels_gnc_pin.py -m gpu -q -s 2>&1 | tail -12
echo "=== rtr_run"; python profiles/experiments/rtr_run.py 100
echo "=== rtr_trace"; DPGO_HIP_LIB=profiles/experiments/build/rtrtrace/libdpgo_hip.so python profiles/experiments/rtr_trace.py 101
echo "=== timeline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/rtr_tl -o rtr -- python $GRAFT_REPO_ROOT/profiles/experiments/rtr_run.py 100 > /tmp/rtr_tl.log 2>&1
python $GRAFT_REPO_ROOT/profiles/prof_query.py /tmp/rtr_tl/rtr_results.db 2000 40 | tail -42
