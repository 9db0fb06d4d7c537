# BASELINE configs[2] on one GPU: sphere2500 over 8 agents (7 x 312 + 316 poses), RTR 3/50 + Nesterov, and the same
# team under accelerated RGD; ms per RBCD iteration
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
mp = capi.partition(m, n, 8); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
for name, kw, warm, K in (("rtr_nesterov", dict(method=0, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=1e-2, restart_interval=50), 50, 200),
                          ("rgd_nesterov", dict(method=1, rgd_stepsize=0.2, acceleration=1, restart_interval=20), 100, 2000)):
    t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=8, **kw))
    t.set_initial(T, Y); t.run(warm); t.synchronize()
    t0 = time.perf_counter(); t.run(K); t.synchronize()
    print(name, "8 agents: ms/iter %.4f" % ((time.perf_counter() - t0) / K * 1e3), "precond us", t.time_kernel(1, 0, reps=200)[0] * 1e3)
    t.close()
