"""per-phase timestamps of the fused iteration kernel (build csrc with CXXFLAGS+=-DDPGO_ITER_TRACE first)"""
import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
team = capi.Team.from_measurements(mp, prm, device=0)
team.set_initial(T, Y)
names = ["start", "eval done", "lookahead done", "drained", "arrived", "poll done", "sync B", "la stored", "staged+sync", "reduced", "sync C", "end", "slab in regs", "fma done"]
for rep in range(4):
    team.run(37)
    buf = np.zeros(18 * 16 + 48, dtype=np.uint64)
    capi.lib().dpgo_team_read_handoff_state(team.h, capi._d(buf), len(buf))
    tr = buf[17 * 16 + 2:]
    for w, off in (("hw0 (eval tile)", 0), ("hw100", 16)):
        t = tr[off:off + 14].astype(np.int64)
        print(w, " ".join("%s=%.2f" % (names[k], (t[k] - t[0]) / 100.0) for k in (0,1,2,3,4,5,6,7,8,12,13,9,10,11) if t[k]))
