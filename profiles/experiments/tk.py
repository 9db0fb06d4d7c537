# dpgo_team_time_kernel variants for the fused step kernel (9 back to back, 10 inside the running iteration)
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
mp = capi.partition(m, n, 5); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, method=1, rgd_stepsize=0.2, acceleration=1, restart_interval=20))
t.set_initial(T, Y); t.run(2100); t.synchronize()
for which in (10, 9, 0):
    print(which, t.time_kernel(1, which, reps=60 if which == 10 else 500))
