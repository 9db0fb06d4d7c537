python -m pytest tests/test_gpu_parity.py -m gpu -q -k "chordal" 2>&1 | tail -5
python - <<'P'
import time, os, numpy as np, subprocess, sys
from oracle import oracle as O
for ds in ("sphere2500","torus3D","smallGrid3D","cubicle","parking-garage"):
    m,n=O.read_g2o("data/%s.g2o"%ds)
    To=O.chordal_init(m,n)
    for mode in ("0","1"):
        os.environ["DPGO_CHORDAL_DENSE"]=mode
        code=("import time,numpy as np;from dpgo_ros_amd import capi;m,n=capi.read_g2o('data/%s.g2o');capi.chordal_init(m,n);"
              "ts=[]\nfor _ in range(3):\n t=time.perf_counter();T=capi.chordal_init(m,n);ts.append(time.perf_counter()-t)\n"
              "np.save('/tmp/T.npy',T);print(min(ts)*1e3)") % ds
        out=subprocess.check_output([sys.executable,"-c",code],text=True,env=dict(os.environ)).strip().splitlines()[-1]
        T=np.load('/tmp/T.npy')
        print(ds, "dense" if mode=="1" else "team ", "ms", out, "max|T-T_oracle|", np.abs(T-To).max(), flush=True)
P
