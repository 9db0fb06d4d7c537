"""BASELINE configs[3] (merged torus3D + cubicle + parking-garage, 8 agents, GNC-TLS): per agent, how far the GPU
preconditioner apply and the iterates are from the oracle's (sparse Cholesky), by preconditioner form.
PRECOND_MODE=0 automatic (two-level for these 1551-pose agents), 1 dense inverse."""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from dpgo_ros_amd import capi
from oracle import oracle as O
from util import merged_graph, params_pair, random_point, relerr
N = 8
mo, n = merged_graph()
mp = O.partition(mo, n, N)
T = O.odometry_init(mo, n)
kw = dict(method=capi.METHOD_RTR, gradnorm_tol=0.5, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0,
          gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=3, robust_opt_inner_iters=8,
          precond_mode=int(os.environ.get("PRECOND_MODE", "0")))
ph, po = params_pair(r=5, num_robots=N, **kw)
po.precond_mode = 0
th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
to = O.Team(mp, n, po)
Y = O.fixed_stiefel(5)
th.set_initial(T, Y); to.set_initial(T, Y)
rng = np.random.default_rng(0)
for a in range(N):
    ah, ao = th.agents[a], to.agents[a]
    ah.build_problem(False); ao.build_problem(False)
    X = random_point(rng, 5, ah.n); V = rng.standard_normal(X.size)
    print("agent %d: %s, precondition relerr vs sparse Cholesky %.2e" % (a, ah.preconditioner_info(), relerr(ah.precondition(X, V), ao.precondition(X, V))))
for rnd in range(2):
    th.run(8)
    for _ in range(8): to.iterate()
    off = 0
    Xh, Xo = th.global_X(), to.global_X()
    for a in range(N):
        na = th.agents[a].n * 20
        print("round %d agent %d: max|X - X_oracle| = %.2e (max |X| %.1f)" % (rnd, a, np.abs(Xh[off:off + na] - Xo[off:off + na]).max(), np.abs(Xo[off:off + na]).max()))
        off += na
    print("cost rel diff %.2e" % (abs(th.cost() - to.cost()) / abs(to.cost())))
    th.update_weights(); to.update_weights()
