#!/bin/bash
# runtime knobs of the HIP runtime against the headline (k_step_fd, graphs of 256 launches) and the drop-in path:
# where do kernel arguments live, are graph packets pre-built?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/knobs
cd $R
python -c "import bench; bench.build_agent_api_bench()"
run() {
  echo "== $1"
  env $1 timeout 300 python profiles/experiments/fd_check.py 5 5 2>&1 | grep -E "RESULT|^ms/iter|k_step_fd" | tr '\n' ' '; echo
  env $1 tests/cpp/agent_api_bench data/sphere2500.g2o 5 1 1 400 | cut -c1-140
}
{
run "DPGO_NOP=1"
run "HIP_FORCE_DEV_KERNARG=1"
run "HIP_FORCE_DEV_KERNARG=0"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"
run "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"
run "GPU_MAX_HW_QUEUES=1"
run "HSA_KERNARG_POOL_SIZE=4194304"
} 2>&1 | tee $R/gpurun_out/knobs/knobs.log
