import os, sys, numpy as np
sys.path.insert(0, ".")
from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import DATA, params_pair, random_point, relerr
m, n = O.read_g2o(os.path.join(DATA, "cubicle.g2o"))
T = O.chordal_init(m, n); Y = O.fixed_stiefel(5)
kw = dict(r=5, num_robots=1, method=capi.METHOD_RTR, rtr_iterations=2, rtr_tcg_iterations=8, gradnorm_tol=1e-3)
ph, po = params_pair(**kw)
th = capi.Team.from_measurements(m.view(capi.MEAS_DTYPE), ph); to = O.Team(m, n, po)
th.set_initial(T, Y); to.set_initial(T, Y)
ah, ao = th.agents[0], to.agents[0]
ah.build_problem(False); ao.build_problem(False)
rng = np.random.default_rng(0)
X = random_point(rng, 5, n); V = rng.standard_normal(X.size)
ph_, po_ = ah.precondition(X, V), ao.precondition(X, V)
print("precond relerr", relerr(ph_, po_), np.abs(po_).max())
fh, egh, rgh = ah.eval(X); fo, ego, rgo = ao.eval(X)
print("eval", abs(fh-fo)/abs(fo), relerr(egh, ego))
for k in range(2):
    th.run(1); to.iterate()
    rh, ro = ah.opt_result(), ao.opt_result()
    print(k, th.cost(), to.cost(), rh.tcg_iters_total, ro.tcg_iters_total, rh.accepted, ro.accepted, rh.f_init, ro.f_init, rh.gradnorm_init, ro.gradnorm_init)
