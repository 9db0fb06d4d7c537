#!/bin/bash
# RTR iterate(true) of the drop-in path: report on the solve's launch (A), report kernel behind the same solve (B), the
# solve compiled without the report's code + report kernel (C: profiles/experiments/build/rtr_norep)
R=$GRAFT_REPO_ROOT; cd $R
python -c "import bench; bench.build_agent_api_bench()"
for rep in 1 2 3; do
  echo "A $(tests/cpp/agent_api_bench data/sphere2500.g2o 5 0 1 100 | cut -c1-120)"
  echo "B $(DPGO_REPORT_TAIL=0 tests/cpp/agent_api_bench data/sphere2500.g2o 5 0 1 100 | cut -c1-120)"
  echo "C $(DPGO_REPORT_TAIL=0 LD_LIBRARY_PATH=$R/profiles/experiments/build/rtr_norep:$LD_LIBRARY_PATH tests/cpp/agent_api_bench data/sphere2500.g2o 5 0 1 100 | cut -c1-120)"
done
ldd tests/cpp/agent_api_bench | grep dpgo
LD_LIBRARY_PATH=$R/profiles/experiments/build/rtr_norep:$LD_LIBRARY_PATH ldd tests/cpp/agent_api_bench | grep dpgo
