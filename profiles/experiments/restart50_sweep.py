"""restart interval 50 (launch/PGOAgent.launch:25) with smaller RGD steps on the bench workload: which step converges,
and in how many iterations (bench.py `rgd_nesterov_restart_50`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dpgo_ros_amd import capi

m, mp, n, T, Y = bench.load_problem(capi)
fstar = bench.F_STAR["sphere2500"]
for step in (0.18, 0.15, 0.12, 0.1, 0.05):
    prm = capi.default_params(r=5, num_robots=5, **dict(bench.RGD, restart_interval=50, rgd_stepsize=step))
    t = capi.Team.from_measurements(mp, prm, device=0)
    t.set_initial(T, Y)
    k, gap, hit, worst = 0, float("inf"), None, 0.0
    while k < 40000:
        t.run(200); t.synchronize(); k += 200
        gap = (t.cost() - fstar) / fstar
        worst = max(worst, gap) if k > 2000 else worst
        if not gap == gap or gap > 1e6:
            break
        if gap <= 1e-6:
            hit = k
            break
    print("step %.2f restart 50: hit %s gap %.3e worst-after-2000 %.3e" % (step, hit, gap, worst), flush=True)
    t.close()
