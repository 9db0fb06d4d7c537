"""one UPDATE_WEIGHT round on the graph of bench.py's GNC leg (torus3D + 10 % seeded outliers, 8 agents, the launch file's
GNC-TLS parameters): stage times (DPGO_TIMING=1) and the per-round wall time, as bench.py measures it"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from dpgo_ros_amd import capi
import bench
m, n = capi.read_g2o(os.path.join(ROOT, 'data/torus3D.g2o'))
mo = bench.add_outliers(capi, m, n)
N = 8
mp = capi.partition(mo, n, N)
T = capi.odometry_init(mo, n); Y = capi.fixed_stiefel(5)
kw = dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, robust_cost_type=5,
          gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=3, robust_opt_inner_iters=50 * N)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, **kw), device=0)
t.set_initial(T, Y); t.synchronize()
print(t.agents[0].preconditioner_info())
ts = []
for u in range(4):
    t.run(int(os.environ.get("ITERS", "400"))); t.synchronize()
    t0 = time.perf_counter(); t.update_weights(); t.synchronize(); ts.append(time.perf_counter() - t0)
print("update_weights ms:", ["%.2f" % (x * 1e3) for x in ts])
