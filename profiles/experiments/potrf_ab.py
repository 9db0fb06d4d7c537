"""the two-wave factorisation of the diagonal block (dense_inverse.hip, round 6) against the library built before it
(DPGO_HIP_LIB=profiles/experiments/build/potrf_old/libdpgo_hip.so): the same run in both, final iterates to a file, and timings
of what the factorisation is in (team set-up, an UPDATE_WEIGHT round, the chordal initialisation)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from dpgo_ros_amd import capi
import bench
out = sys.argv[1]
m, n = capi.read_g2o(os.path.join(ROOT, 'data/torus3D.g2o'))
mo = bench.add_outliers(capi, m, n)
N = 8
mp = capi.partition(mo, n, N)
T = capi.odometry_init(mo, n); Y = capi.fixed_stiefel(5)
kw = dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, robust_cost_type=5, gnc_barc=3.0)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, **kw), device=0)
t.set_initial(T, Y); t.synchronize()
ts = []
for u in range(4):
    t.run(40); t.synchronize()
    t0 = time.perf_counter(); t.update_weights(); t.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
X = np.concatenate([t.agents[a].get_X() for a in t.ids])
t.close()
# dense agents: sphere2500 / 5 (set-up = 5 inverses of 2000^2) + a few iterations
m2, n2 = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
mp2 = capi.partition(m2, n2, 5)
a0 = time.perf_counter()
t2 = capi.Team.from_measurements(mp2, capi.default_params(r=5, num_robots=5, **bench.RGD), device=0)
t2.set_initial(capi.odometry_init(m2, n2), Y); t2.run(5); t2.synchronize()
setup_ms = (time.perf_counter() - a0) * 1e3
t2.run(200); t2.synchronize()
X2 = np.concatenate([t2.agents[a].get_X() for a in t2.ids])
t2.close()
cs = []
for k in range(5):
    c0 = time.perf_counter(); Tc = capi.chordal_init(m2, n2); cs.append((time.perf_counter() - c0) * 1e3)
np.savez(out, X=X, X2=X2, Tc=Tc)
print("update_weights ms %s | set-up + 5 iterations %.1f ms | chordal init ms %s" % (["%.2f" % x for x in ts], setup_ms, ["%.2f" % x for x in cs]))
