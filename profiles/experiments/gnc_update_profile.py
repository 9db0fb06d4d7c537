"""UPDATE_WEIGHT rounds of the bench's GNC leg alone (torus3D + 10 % outliers, 8 agents, RTR): wall time per round and,
under rocprofv3 --kernel-trace --stats, the kernels behind it"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from dpgo_ros_amd import capi
N = 8
kw = dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, robust_cost_type=5,
          gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=30, robust_opt_inner_iters=50 * N)
m, n = capi.read_g2o(os.path.join(bench.ROOT, "data", "torus3D.g2o"))
mo = bench.add_outliers(capi, m, n)
mp = capi.partition(mo, n, N)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, **kw), device=0)
t.set_initial(capi.odometry_init(mo, n), capi.fixed_stiefel(5))
t.run(2 * N); t.synchronize()
print(t.agents[0].preconditioner_info())
ts = []
for k in range(6):
    t0 = time.perf_counter(); t.update_weights(); t.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t.run(N); t.synchronize()
print("update_weights ms:", ["%.2f" % x for x in ts])
