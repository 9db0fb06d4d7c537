"""measured floor of the HIP-vs-oracle differences in the two parity tests the round-3 verdict called loose
(tests/test_gpu_parity.py::test_tunnels_eight_agents, ::test_config3_torus_eight_agents_gnc_with_outliers)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load, load_tunnels, add_outliers
from tests.test_gpu_parity import _pair_from

for mode, accel in ((capi.WEIGHT_WRAPPER, 0), (capi.WEIGHT_LIBRARY, 1)):
    m = load_tunnels(mode)
    N = 8
    if mode == capi.WEIGHT_LIBRARY:
        m = m.copy(); m["weight"] = 1.0
    nk = [0] * N
    for e in m:
        nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1); nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
    Ts = []
    for k in range(N):
        odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy(); odo["r1"] = 0; odo["r2"] = 0
        Ts.append(O.odometry_init(odo, nk[k]))
    T = np.concatenate(Ts)
    kw = dict(method=capi.METHOD_RTR, gradnorm_tol=1e-2, acceleration=accel, restart_interval=11)
    th, to = _pair_from(m, sum(nk), N, T, **kw)
    th.run(16)
    for _ in range(16): to.iterate()
    X = to.global_X()
    print("tunnels mode %d accel %d: max|dX| %.3e (max|X| %.3e), rel cost diff %.3e" % (mode, accel, np.abs(th.global_X() - X).max(), np.abs(X).max(), abs(th.cost() - to.cost()) / abs(to.cost())))
    th.close()

N = 8
m, _, n = load("torus3D", 1)
mo = add_outliers(m, n, frac=0.02, seed=0)
mp = O.partition(mo, n, N)
T = O.odometry_init(mo, n)
kw = dict(method=capi.METHOD_RTR, gradnorm_tol=0.5, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-5,
          robust_opt_num_weight_updates=3, robust_opt_inner_iters=8)
th, to = _pair_from(mp, n, N, T, **kw)
for rnd in range(2):
    th.run(8)
    for _ in range(8): to.iterate()
    dx = np.abs(th.global_X() - to.global_X()).max()
    th.update_weights(); to.update_weights()
    wh = np.concatenate([th.agents[a].measurements()["weight"] for a in range(N)])
    wo = np.concatenate([to.agents[a].measurements()["weight"] for a in range(N)])
    print("config3 torus round %d: max|dX| %.3e (max|X| %.3e), max|dw| %.3e" % (rnd, dx, np.abs(to.global_X()).max(), np.abs(wh - wo).max()))
th.close()
