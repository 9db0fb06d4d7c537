# lockstep ASAPP on the 8-robot tunnels graph: cost after T ticks for a few step sizes, ms per tick
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
rows, seen = [], set()
for k in range(8):
    for e in capi.read_csv(os.path.join(ROOT, "data/tunnels/robot%d/measurements.csv" % k), capi.WEIGHT_WRAPPER):
        key = (int(e["r1"]), int(e["p1"]), int(e["r2"]), int(e["p2"]))
        if key not in seen:
            seen.add(key); rows.append(e)
m = np.array(rows, dtype=capi.MEAS_DTYPE)
N = 8
nk = [0] * N
for e in m:
    nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1); nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
Ts = []
for k in range(N):
    odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy(); odo["r1"] = 0; odo["r2"] = 0
    Ts.append(capi.odometry_init(odo, nk[k]))
T = np.concatenate(Ts); Y = capi.fixed_stiefel(5)
for step in (0.2, 0.1, 0.05):
    t = capi.Team.from_measurements(m, capi.default_params(r=5, num_robots=N, method=1, rgd_stepsize=step, acceleration=0))
    t.set_initial(T, Y)
    c0 = t.cost(); t.run_simultaneous(64); t.synchronize()
    t0 = time.perf_counter(); t.run_simultaneous(640); t.synchronize(); dt = time.perf_counter() - t0
    print("step", step, "cost0 %.4e" % c0, "after 704 ticks %.6e" % t.cost(), "ms/tick %.4f" % (dt / 640 * 1e3))
    t.close()
