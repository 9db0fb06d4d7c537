import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
N = int(os.environ.get("NAGENTS", "5"))
mp = capi.partition(m, n, N) if N > 1 else m
for rep in range(3):
    t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, method=1))
    t0 = time.perf_counter(); t.set_initial(T, Y); t.synchronize(); dt = time.perf_counter() - t0
    print("agents %d poses %d: set-up %.2f ms" % (N, t.agents[0].n, dt * 1e3))
    t.close()
