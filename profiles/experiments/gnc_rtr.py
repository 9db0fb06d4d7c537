# where the time of the GNC / RTR configuration goes: torus3D + outliers, 8 agents, RTR 3/50/0.5 without acceleration
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import bench
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/torus3D.g2o'))
mo = bench.add_outliers(capi, m, n); N = 8
mp = capi.partition(mo, n, N)
kw = dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, robust_cost_type=5,
          gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=3, robust_opt_inner_iters=400)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, **kw))
t.set_initial(capi.odometry_init(mo, n), capi.fixed_stiefel(5)); t.synchronize()
for u in range(4):
    c0 = t.counters(); t0 = time.perf_counter(); t.run(400); t.synchronize(); dt = time.perf_counter() - t0
    c = t.counters() - c0
    print("block %d: %.3f ms/iter  precond/iter %.1f  spmm/iter %.1f" % (u, dt / 400 * 1e3, c[0] / 400, c[2] / 400), flush=True)
    if u < 3: t.update_weights()
