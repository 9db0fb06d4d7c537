"""The one-launch iteration (k_step_fe) on agents below 257 poses (DPGO_FE_MIN_N): bitwise against the two-launch
sequence and time per iteration, per dataset / team."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from dpgo_ros_amd import capi
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
RGD = dict(method=1, acceleration=1, rgd_stepsize=0.2, rgd_use_preconditioner=1, restart_interval=20)
def team(ds, N, fused, r=5):
    os.environ["DPGO_FUSED_EVAL"] = "1" if fused else "0"
    m, n = capi.read_g2o(os.path.join(ROOT, "data", ds + ".g2o"))
    mp = capi.partition(m, n, N)
    t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=N, **RGD))
    t.set_initial(capi.odometry_init(m, n), capi.fixed_stiefel(r))
    return t
for ds, N, r in (("sphere2500", 5, 5), ("sphere2500", 6, 5), ("sphere2500", 7, 5), ("sphere2500", 8, 5), ("sphere2500", 6, 4), ("sphere2500", 5, 3),
                 ("torus3D", 10, 5), ("parking-garage", 4, 5), ("parking-garage", 6, 5), ("smallGrid3D", 3, 5)):
    ta, tb = team(ds, N, False, r), team(ds, N, True, r)
    ok = True
    for iters in (23, 300, 64, 7, 129):
        ta.run(iters); ta.synchronize(); tb.run(iters); tb.synchronize()
        for k in ta.ids:
            ok = ok and np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X())
    ms = []
    for t in (ta, tb):
        t.run(512); t.synchronize()
        t0 = time.perf_counter(); t.run(2048); t.synchronize(); ms.append((time.perf_counter() - t0) / 2048 * 1e3)
    print("%-16s N=%d r=%d poses/agent %d: bitwise %s, one-launch iterations %d, ms/iter two-launch %.4f one-launch %.4f" % (
        ds, N, r, ta.agents[ta.ids[0]].n if hasattr(ta.agents[ta.ids[0]], "n") else -1, ok, tb.counters()[7], ms[0], ms[1]))
    ta.close(); tb.close()
