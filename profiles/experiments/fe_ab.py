"""one-launch iterations (step_fused.hip) against the two-launch sequence: bitwise comparison of the iterates and ms per
iteration of dpgo_team_run on the bench configuration.  DPGO_FUSED_EVAL is read when a team is created."""
import os
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import bench
from dpgo_ros_amd import capi

m, mp, n, T, Y = bench.load_problem(capi)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def run(fe, iters):
    os.environ["DPGO_FUSED_EVAL"] = "1" if fe else "0"
    prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
    team = capi.Team.from_measurements(mp, prm, device=0)
    team.set_initial(T, Y)
    team.prepare(iters)
    team.run(iters)
    team.synchronize()
    X = [np.array(a.get_X()) for a in team.agents.values()] if isinstance(team.agents, dict) else [np.array(a.get_X()) for a in team.agents]
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        team.run(iters)
        team.synchronize()
        best = min(best, (time.perf_counter() - t0) / iters * 1e3)
    return X, best, team


for iters in (37, 129, K):
    Xa, ta, _ = run(False, iters)
    Xb, tb, _ = run(True, iters)
    d = max(float(np.max(np.abs(a - b))) for a, b in zip(Xa, Xb))
    print("iters %5d  two-launch %.5f ms  one-launch %.5f ms  max|dX| %.3e" % (iters, ta, tb, d), flush=True)
