# does accelerated RGD reach a 1e-6 relative cost gap from the chordal initial guess?  (step, restart) sweep
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
FSTAR = 843.5029071410438
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
mp = capi.partition(m, n, 5); Y = capi.fixed_stiefel(5)
Tch = capi.chordal_init(m, n); Tod = capi.odometry_init(m, n)
for init, T in (("chordal", Tch), ("odometry", Tod)):
    for step, restart in ((0.2, 20), (0.5, 20), (1.0, 20), (0.2, 50), (1.0, 50), (1.0, 100)):
        t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, method=1, rgd_stepsize=step, acceleration=1, restart_interval=restart))
        t.set_initial(T, Y)
        hit = None
        for k in range(200):
            t.run(100)
            gap = (t.cost() - FSTAR) / FSTAR
            if not (gap < 1e6): break
            if gap <= 1e-6: hit = (k + 1) * 100; break
        print(init, "step", step, "restart", restart, "hit", hit, "gap %.3e" % gap, "iters", (k + 1) * 100, flush=True)
        t.close()
