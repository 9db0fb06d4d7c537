import sys, time, json, os
sys.path.insert(0, ".")
import numpy as np
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
team = capi.Team.from_measurements(mp, prm, device=0)
team.set_initial(T, Y)
team.run(200); team.prepare(2000); team.synchronize()
for rep in range(3):
    t0=time.perf_counter(); team.run(2000); team.synchronize(); dt=time.perf_counter()-t0
    print("fused=%s ms/iter %.5f cost %.9f" % (os.environ.get("DPGO_FUSED_ITER","1"), dt/2000*1e3, team.cost()))
