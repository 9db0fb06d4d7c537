#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for fold in 1 0; do
  DPGO_SYRK_FOLD=$fold python $R/profiles/experiments/gnc_update_torus.py 2>&1 | grep "update_weights ms"
  DPGO_SYRK_FOLD=$fold rocprofv3 --kernel-trace --output-format csv -d /tmp/trf$fold -- python $R/profiles/experiments/gnc_update_torus.py > /dev/null 2>&1
  f=$(find /tmp/trf$fold -name '*kernel_trace.csv' | head -1)
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_tl_gather" in r["Kernel_Name"]]
lo=idx[-2]
seg=rows[lo:lo+60]
out=[]
for r in seg:
    nm=r["Kernel_Name"]
    if "k_wtw" in nm: break
    for key in ("k_potrf","k_trsm","k_syrk"):
        if key in nm: out.append("%s %.0f" % (key[2:6], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print("FOLD=$fold:", " | ".join(out))
PY
done
