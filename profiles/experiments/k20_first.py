"""Why is the FIRST replay of the K = 20 graph (bench.py's ms_per_step_k_region at --steps 20 --warmup 5) slower than the
15 that follow (0.0184 vs 0.0166 ms per step)?  The same sequence as bench.py, with different things between prepare() and
the timed region: nothing (the bench), read-only GPU work (cost evaluations), an idle sleep."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from dpgo_ros_amd import capi
import bench

m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)


def trial(between, K=20, W=5):
    team = capi.Team.from_measurements(mp, prm, device=0)
    team.set_initial(T, Y)
    team.run(W)
    team.prepare(K)
    team.synchronize()
    between(team)
    t0 = time.perf_counter()
    team.run(K)
    team.synchronize()
    first = (time.perf_counter() - t0) / K * 1e3
    rest = []
    for _ in range(15):
        a0 = time.perf_counter(); team.run(K); team.synchronize(); rest.append((time.perf_counter() - a0) / K * 1e3)
    team.close()
    return first, float(np.mean(rest)), float(np.min(rest))


def busy(k):
    def f(team):
        for _ in range(k):
            team.cost()
    return f


for name, fn in (("nothing (bench.py)", lambda t: None), ("20 cost evaluations", busy(20)), ("200 cost evaluations", busy(200)),
                 ("sleep 50 ms", lambda t: time.sleep(0.05)), ("nothing (bench.py)", lambda t: None)):
    for rep in range(2):
        f, mean, mn = trial(fn)
        print("%-22s first %.4f  mean of 15 %.4f  min %.4f ms per step" % (name, f, mean, mn), flush=True)
