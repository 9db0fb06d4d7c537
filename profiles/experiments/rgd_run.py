# ms per accelerated-RGD iteration of the bench workload (dpgo_team_run only); argv[1] = iterations
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
mp = capi.partition(m, n, 5); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, method=1, rgd_stepsize=0.2, acceleration=1, restart_interval=20))
t.set_initial(T, Y)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
t.run(100); t.synchronize()
t0 = time.perf_counter(); t.run(K); t.synchronize(); print("ms/iter", (time.perf_counter() - t0) / K * 1e3)
