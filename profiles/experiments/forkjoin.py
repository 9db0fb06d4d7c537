"""planning aid: per-iteration time of { [preconditioner apply || evaluation of another agent] -> tiny kernel }
in a 64-iteration hipGraph on two streams, against the same kernels in one chain"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from dpgo_ros_amd import capi
m, n = capi.read_g2o('data/sphere2500.g2o')
mp = capi.partition(m, n, 5); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, method=1, rgd_stepsize=0.2, acceleration=1, restart_interval=20))
t.set_initial(T, Y); t.run(10); t.synchronize()
for rep in range(2):
    fj, _ = t.time_kernel(0, 12, reps=40)
    ch, _ = t.time_kernel(0, 13, reps=40)
    ev, _ = t.time_kernel(1, 1, reps=200)
    pc, _ = t.time_kernel(0, 0, reps=200)
    print("fork/join %.2f us per iteration | one chain %.2f us | alone: apply %.2f, evaluation %.2f" % (fj * 1e3, ch * 1e3, pc * 1e3, ev * 1e3))
