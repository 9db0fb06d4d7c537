# one UPDATE_WEIGHT round of the 8-agent torus3D team (625-pose agents), by preconditioner form (PRECOND_MODE 0 auto = two-level under RTR, 1 dense)
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/torus3D.g2o'))
T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
N = 8
mp = capi.partition(m, n, N)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, method=0, robust_cost_type=5, gnc_barc=5.0, gradnorm_tol=1e-2,
                                                        precond_mode=int(os.environ.get("PRECOND_MODE", "0"))))
t.set_initial(T, Y); t.run(2 * N); t.synchronize()
print(t.agents[0].preconditioner_info())
ts = []
for k in range(4):
    t0 = time.perf_counter(); t.update_weights(); t.synchronize(); ts.append(time.perf_counter() - t0)
    t.run(N); t.synchronize()
print("update_weights ms:", ["%.2f" % (x * 1e3) for x in ts])
