# RTR + Nesterov on sphere2500 / 5: counter-tree hand-off (default) against the flat one (-DDPGO_RTR_FLATBAR=1), then the
# dispatch timeline of the default build (gaps between the launches of an iteration)
cd $GRAFT_REPO_ROOT
echo "=== default"; python profiles/experiments/rtr_run.py 100
echo "=== flatbar"; DPGO_HIP_LIB=profiles/experiments/build/flatbar/libdpgo_hip.so python profiles/experiments/rtr_run.py 100
DPGO_HIP_LIB=profiles/experiments/build/flatbar/libdpgo_hip.so python -m pytest tests/test_gpu_rtr_fused.py -m gpu -q 2>&1 | tail -3
echo "=== timeline"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/rtr_tl -o rtr -- python $GRAFT_REPO_ROOT/profiles/experiments/rtr_run.py 100 > /tmp/rtr_tl.log 2>&1
python $GRAFT_REPO_ROOT/profiles/prof_query.py /tmp/rtr_tl/rtr_results.db 2000 40 | tail -42
