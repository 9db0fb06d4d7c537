#!/bin/bash
# drop-in path (tests/cpp/agent_api_bench), 5 single-agent teams in ONE process: do their five streams share hardware queues?
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/api
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench; print(bench.build_agent_api_bench())
PY
for q in default 4 8 16; do
  for rep in 1 2; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    echo "queues=$q rgd $($R/tests/cpp/agent_api_bench $R/data/sphere2500.g2o 5 1 1 400 | cut -c1-330)"
  done
  echo "queues=$q rtr $($R/tests/cpp/agent_api_bench $R/data/sphere2500.g2o 5 0 1 100 | cut -c1-330)"
done 2>&1 | tee $R/gpurun_out/api/queues.log
