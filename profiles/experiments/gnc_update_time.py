# cost of one UPDATE_WEIGHT round (residual kernel, GNC-TLS weights, Q / G / dense preconditioner rebuild of every agent)
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
for N in (8, 5):
    mp = capi.partition(m, n, N)
    t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, method=0, robust_cost_type=5, gnc_barc=5.0, gradnorm_tol=1e-2))
    t.set_initial(T, Y); t.run(2 * N); t.synchronize()
    ts = []
    for k in range(3):
        t0 = time.perf_counter(); t.update_weights(); t.synchronize(); ts.append(time.perf_counter() - t0)
        t.run(N)
    print("agents %d: update_weights %.1f ms (min of 3: %s)" % (N, min(ts) * 1e3, ["%.1f" % (x * 1e3) for x in ts]))
    t.close()
