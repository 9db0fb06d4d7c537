"""soak of the one-launch iteration: long runs, bitwise against the two-launch sequence, several team sizes"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(bench.ROOT, "data", "sphere2500.g2o"))
T, Y = capi.odometry_init(m, n), capi.fixed_stiefel(5)
for robots in (5, 6, 7, 8):
    mp = capi.partition(m, n, robots)
    teams = []
    for fe in (0, 1):
        os.environ["DPGO_FUSED_EVAL"] = str(fe)
        t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=robots, **bench.RGD), device=0)
        t.set_initial(T, Y)
        teams.append(t)
    for chunk in (100000, 33333, 1, 2, 50001):
        for t in teams:
            t0 = time.perf_counter(); t.run(chunk); t.synchronize(); dt = time.perf_counter() - t0
        d = max(float(np.max(np.abs(teams[0].agents[k].get_X() - teams[1].agents[k].get_X()))) for k in teams[0].ids)
        print("robots %d  +%6d iterations  max|dX| %.1e  cost %.9f  one-launch iterations so far %d  (%.4f ms/it)"
              % (robots, chunk, d, teams[1].cost(), teams[1].counters()[7], dt / chunk * 1e3), flush=True)
        assert d == 0.0
    for t in teams:
        t.close()
print("soak ok")
