import sys; sys.path.insert(0, ".")
import torch
import bench, time
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
t = capi.Team.from_measurements(mp, prm, device=0); t.set_initial(T, Y); t.run(100); t.prepare(4000); t.synchronize()
for rep in range(3):
    a0=time.perf_counter(); t.run(4000); t.synchronize(); print("ms/iter %.5f" % ((time.perf_counter()-a0)/4000*1e3))
for rep in range(3):
    ms, b = t.time_kernel(1, 14, reps=500); print("k_step_fe us %.3f" % (ms*1e3))
