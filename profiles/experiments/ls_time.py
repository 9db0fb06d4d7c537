"""ms per iteration of the accelerated RGD with backtracking line search on the bench configuration, and of the eager
(agent by agent) RTR / RGD iterations that use the pose-wise kernels"""
import sys, time
sys.path.insert(0, ".")
import torch
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
kw = dict(bench.RGD, rgd_line_search=1, rgd_stepsize=1.0)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, **kw), device=0)
t.set_initial(T, Y); t.run(200); t.synchronize()
best = 1e9
for rep in range(4):
    a0 = time.perf_counter(); t.run(1000); t.synchronize(); best = min(best, (time.perf_counter() - a0))
print("line search: ms/iter %.5f" % best)
t.close()
