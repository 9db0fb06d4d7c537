"""deep-carried one-launch iteration (csrc/step_deep.hip) against the two-launch sequence, bit for bit, then its
dispatch-to-dispatch time.  usage: fd_check.py [robots] [r]"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load

robots = int(sys.argv[1]) if len(sys.argv) > 1 else 5
r = int(sys.argv[2]) if len(sys.argv) > 2 else 5
RGD = dict(method=1, acceleration=1, rgd_stepsize=0.2, rgd_use_preconditioner=1, restart_interval=20)


def team(fused, deep=True, persist=False):
    os.environ["DPGO_FE_PERSIST"] = "1" if persist else "0"
    os.environ["DPGO_FUSED_EVAL"] = "1" if fused else "0"
    os.environ["DPGO_FE_DEEP"] = "1" if deep else "0"
    os.environ["DPGO_FE_MIN_N"] = "32"
    m, mp, n = load("sphere2500", robots)
    t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=r, num_robots=robots, **RGD))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(r))
    return t


PERSIST = os.environ.get("FD_CHECK_PERSIST") == "1"
ta, tb = team(False), team(True, persist=PERSIST)
ok = True
for iters in (23, 300, 64, 7, 129, 41):
    ta.run(iters); ta.synchronize(); tb.run(iters); tb.synchronize()
    worst = 0.0
    for k in ta.ids:
        xa, xb = ta.agents[k].get_X(), tb.agents[k].get_X()
        if not np.array_equal(xa, xb):
            ok = False
            worst = max(worst, float(np.abs(xa - xb).max()))
    print("iters %4d: %s (max diff %.3e)  counters fe %d carried %d deep %d" % (iters, "bitwise" if worst == 0 else "DIFFERENT", worst, tb.counters()[7], tb.counters()[8], tb.counters()[9]), flush=True)
print("cost two-launch %.15g deep %.15g" % (ta.cost(), tb.cost()))
print("RESULT", "OK" if ok else "MISMATCH")
if True:
    tc = team(True, deep=False)
    for nm, tt in (("deep", tb), ("round-5 form", tc)):
        tt.prepare(4000); tt.synchronize()
        for rep in range(2):
            a0 = time.perf_counter(); tt.run(4000); tt.synchronize(); print("%s ms/iter %.5f (deep-carried launches %d)" % (nm, (time.perf_counter() - a0) / 4000 * 1e3, tt.counters()[9]))
if robots == 5 and r == 5:
    tb.prepare(4000); tb.synchronize()
    for rep in range(3):
        a0 = time.perf_counter(); tb.run(4000); tb.synchronize(); print("ms/iter %.5f" % ((time.perf_counter() - a0) / 4000 * 1e3))
    for rep in range(3):
        ms, b = tb.time_kernel(1, 14, reps=500); print("k_step_fd us %.3f" % (ms * 1e3))
