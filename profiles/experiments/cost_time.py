import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, **bench.RGD), device=0)
t.set_initial(T, Y); t.run(20); t.synchronize()
c = t.cost()
t0 = time.perf_counter()
for _ in range(200): c = t.cost()
print("cost %.9f  %.1f us per call" % (c, (time.perf_counter() - t0) / 200 * 1e6))
