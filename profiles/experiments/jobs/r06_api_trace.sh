#!/bin/bash
# kernel timeline of the drop-in path (tests/cpp/agent_api_bench): which launches an iterate(true) is made of and where the gaps are
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/api
python - <<PY
import sys; sys.path.insert(0, "$R")
import bench; print(bench.build_agent_api_bench())
PY
for m in 1 0; do
  $R/tests/cpp/agent_api_bench $R/data/sphere2500.g2o 5 $m 1 200 > $R/gpurun_out/api/plain_$m.json 2>&1
  cat $R/gpurun_out/api/plain_$m.json
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/api/trace_$m -- $R/tests/cpp/agent_api_bench $R/data/sphere2500.g2o 5 $m 1 40 > $R/gpurun_out/api/traced_$m.json 2>&1
  f=$(find $R/gpurun_out/api/trace_$m -name '*kernel_trace.csv' | head -1)
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-160:-40]
t0=int(rows[0]["Start_Timestamp"])
out=open("$R/gpurun_out/api/timeline_$m.txt","w")
prev=None
for r in rows:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    line="%9.2f %7.2f  q%-3s %s grid %s" % (s/1e3,(e-s)/1e3,r.get("Queue_Id","?"),r["Kernel_Name"][:60],r.get("Grid_Size_X", r.get("Grid_Size","?")))
    out.write(line+"\n")
out.close()
PY
  rm -rf $R/gpurun_out/api/trace_$m
done
