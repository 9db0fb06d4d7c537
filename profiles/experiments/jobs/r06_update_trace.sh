#!/bin/bash
# kernel timeline of one UPDATE_WEIGHT round (torus3D / 8 agents, two-level preconditioners rebuilt): launches, durations, gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/upd
DPGO_TIMING=1 python $R/profiles/experiments/${UPD_SCRIPT:-gnc_update_torus.py} 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -8 | tee $R/gpurun_out/upd/time.log
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/upd/trace -- python $R/profiles/experiments/${UPD_SCRIPT:-gnc_update_torus.py} > /dev/null 2>&1
f=$(find $R/gpurun_out/upd/trace -name '*kernel_trace.csv' | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last UPDATE_WEIGHT round: from the last k_residuals to the first k_nest/k_eval after the rebuild
idx=[i for i,r in enumerate(rows) if "k_residuals" in r["Kernel_Name"]]
lo=idx[-8] if len(idx)>=8 else idx[0]
seg=rows[lo:lo+400]
t0=int(seg[0]["Start_Timestamp"])
with open("$R/gpurun_out/upd/timeline.txt","w") as out:
    prev_end=None
    for r in seg:
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        gap=(s-prev_end)/1e3 if prev_end else 0.0
        out.write("%9.1f %7.1f gap %6.1f  %s grid %s\n" % ((s-t0)/1e3,(e-s)/1e3,gap,r["Kernel_Name"][:70],r.get("Grid_Size_X", r.get("Grid_Size","?"))))
        prev_end=e
        if "k_rtr_solve" in r["Kernel_Name"]: break
PY
rm -rf $R/gpurun_out/upd/trace
wc -l $R/gpurun_out/upd/timeline.txt
