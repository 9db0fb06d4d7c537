#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/upd
for w in 0 1; do
  WARM=$w python $R/profiles/experiments/gnc_update_warm.py 2>&1 | grep "update_weights ms"
  WARM=$w rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/upd/tracew$w -- python $R/profiles/experiments/gnc_update_warm.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/upd/tracew$w -name '*kernel_trace.csv' | head -1)
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_potrf_diag" in r["Kernel_Name"]]
lo=idx[-2]   # first chain of the last round
seg=[r for r in rows[lo:lo+40] if "k_syrk" in r["Kernel_Name"]][:9]
print("WARM=$w syrk us:", ["%.0f" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in seg])
PY
  rm -rf $R/gpurun_out/upd/tracew$w
done
