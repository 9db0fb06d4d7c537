"""the GPU half of bench.py's GNC leg (torus3D + 10 % outliers, 8 agents, GNC-TLS, RTR 3/50/0.5) for a precond_mode
(argv[1]: 0 automatic, 1 dense, 3 two-level): total of the robust schedule, per UPDATE_WEIGHT round, final cost"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from dpgo_ros_amd import capi
N = 8
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kw = dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, robust_cost_type=5, gnc_barc=3.0,
          gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=3, robust_opt_inner_iters=50 * N, precond_mode=mode)
m, n = capi.read_g2o(os.path.join(bench.ROOT, "data", "torus3D.g2o"))
mo = bench.add_outliers(capi, m, n)
mp = capi.partition(mo, n, N)
for rep in range(2):
    t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, **kw), device=0)
    t.set_initial(capi.odometry_init(mo, n), capi.fixed_stiefel(5))
    t.synchronize()
    upd, t0 = 0.0, time.perf_counter()
    for u in range(3):
        t.run(50 * N); t.synchronize()
        u0 = time.perf_counter(); t.update_weights(); t.synchronize(); upd += time.perf_counter() - u0
    t.run(50 * N); t.synchronize()
    total = time.perf_counter() - t0
    print("precond_mode %d (agents run form %d): total %.1f ms, UPDATE_WEIGHT %.2f ms per round, cost %.10f" % (
        mode, t.agents[0].preconditioner(), total * 1e3, upd / 3 * 1e3, t.cost()))
    c = t.counters()
    print("   preconditioner applies %.1f, sparse evaluations %.1f per iteration over %d iterations" % (c[0] / c[4], c[2] / c[4], int(c[4])))
    t.close()
