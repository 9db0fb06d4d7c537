#!/bin/bash
# end of round 6: the whole GPU suite, the profile collection of the default bench command, the driver's K = 20 invocation,
# the kernel timelines of the drop-in path
R=$GRAFT_REPO_ROOT; cd $R
OUT=gpurun_out/r06f; mkdir -p $OUT
timeout 1300 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4 > $OUT/gpu_tests.log; cat $OUT/gpu_tests.log
bash profiles/collect.sh r06f 2>&1 | tail -5
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench_k20.json 2> $OUT/bench_k20.err
bash profiles/experiments/jobs/r06_api_trace.sh > $OUT/api_trace.log 2>&1
cp gpurun_out/api/timeline_1.txt $OUT/api_timeline_rgd.txt; cp gpurun_out/api/timeline_0.txt $OUT/api_timeline_rtr.txt
python - <<'PY'
import json
for f in ("gpurun_out/r06f/bench.json", "gpurun_out/r06f/bench_k20.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "value", d["value"], "k_region", d.get("ms_per_step_k_region"), "frac", d["roofline"]["frac"], "api", {k: v.get("ms_per_iterate_cxx") for k, v in d["convergence"]["agent_api"].items() if isinstance(v, dict)})
PY
