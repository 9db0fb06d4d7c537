#!/bin/bash
# message path in loopback: host-bound or GPU-bound?  + kernel timeline of a few iterations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/loop
python $R/profiles/experiments/loop_time.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee $R/gpurun_out/loop/time.log
LOOP_K=40 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/loop/trace -- python $R/profiles/experiments/loop_time.py > /dev/null 2>&1
f=$(find $R/gpurun_out/loop/trace -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the loopback runs come first: find the last run_ranks region = rows before the first k_step_fd / k_step_fe of the second team
idx=[i for i,r in enumerate(rows) if "k_xfer_multi" in r["Kernel_Name"]]
lo=idx[len(idx)//2]; rows=rows[lo:lo+70]
t0=int(rows[0]["Start_Timestamp"])
with open("$R/gpurun_out/loop/timeline.txt","w") as out:
    for r in rows:
        s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
        out.write("%9.2f %7.2f  q%-3s %s\n" % (s/1e3,(e-s)/1e3,r.get("Queue_Id","?"),r["Kernel_Name"][:90]))
PY
rm -rf $R/gpurun_out/loop/trace
