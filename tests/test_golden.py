"""The C oracle against the committed golden vectors of the independent numpy implementation
(oracle/np_crosscheck.py -> tests/golden/*.npz).  fp64; tolerances reflect dense-LU vs sparse
Cholesky and SVD vs Jacobi differences."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import DATA, ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


def _setup(ds, N, r, **kw):
    g = np.load(os.path.join(GOLD, "%s_N%d_r%d.npz" % (ds, N, r)))
    m, n = O.read_g2o(os.path.join(DATA, ds + ".g2o"))
    mp = O.partition(m, n, N)
    t = O.Team(mp, n, O.default_params(r=r, num_robots=N, **kw))
    T = O.odometry_init(m, n)
    assert np.abs(T - g["T0"]).max() < 1e-12
    t.set_initial(T, O.fixed_stiefel(r))
    return g, t


def _dense_Q(rowptr, col, val, n):
    Q = np.zeros((4 * n, 4 * n))
    for j in range(n):
        for p in range(rowptr[j], rowptr[j + 1]):
            Q[4 * col[p]:4 * col[p] + 4, 4 * j:4 * j + 4] = val[16 * p:16 * p + 16].reshape(4, 4, order="F")
    return Q


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D"])
def test_problem_surface_vs_numpy(ds):
    g, t = _setup(ds, 2, 5)
    for a in range(2):
        ag = t.agents[a]
        ag.build_problem(False)
        Q = _dense_Q(*ag.get_Q(), ag.n)
        assert np.abs(Q - g["a%d_Qdense" % a]).max() < 1e-10 * np.abs(Q).max()
        assert np.abs(Q - Q.T).max() == 0.0
        assert np.abs(ag.get_G() - g["a%d_G" % a]).max() < 1e-10 * max(1, np.abs(g["a%d_G" % a]).max())
        X, eta, V = g["a%d_X" % a], g["a%d_eta" % a], g["a%d_V" % a]
        f, eg, rg = ag.eval(X)
        assert abs(f - g["a%d_f" % a]) < 1e-11 * abs(f)
        assert np.abs(eg - g["a%d_egrad" % a]).max() < 1e-10 * np.abs(eg).max()
        assert np.abs(rg - g["a%d_rgrad" % a]).max() < 1e-10 * np.abs(rg).max()
        h = ag.hessvec(X, eta)
        assert np.abs(h - g["a%d_hess" % a]).max() < 1e-10 * np.abs(h).max()
        pc = ag.precondition(X, V)
        assert np.abs(pc - g["a%d_precond" % a]).max() < 1e-9 * np.abs(pc).max()
        assert np.abs(O.retract(X, 0.3 * eta, 5, ag.n) - g["a%d_retract" % a]).max() < 1e-12
        assert np.abs(O.project_manifold(X + 0.2 * V, 5, ag.n) - g["a%d_project" % a]).max() < 1e-12


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D"])
def test_single_solves_vs_numpy(ds):
    for method, key in ((O.METHOD_RGD, "rgd"), (O.METHOD_RTR, "rtr")):
        g, t = _setup(ds, 2, 5, method=method, rgd_stepsize=0.2, gradnorm_tol=1e-2)
        for a in range(2):
            g2, t2 = _setup(ds, 2, 5, method=method, rgd_stepsize=0.2, gradnorm_tol=1e-2)
            ag = t2.agents[a]
            assert ag.iterate(True)
            assert np.abs(ag.get_X() - g["a%d_%s" % (a, key)]).max() < 1e-7
            if key == "rtr":
                res = ag.opt_result()
                assert res.tcg_iters_total == int(g["a%d_rtr_tcg" % a]) and res.accepted == int(g["a%d_rtr_acc" % a])


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D"])
@pytest.mark.parametrize("name,kw", [
    ("rtr", dict(method=0)),
    ("rtr_acc", dict(method=0, acceleration=1, restart_interval=7)),
    ("rgd_acc", dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=7)),
])
def test_ten_rbcd_iterations_vs_numpy(ds, name, kw):
    g, t = _setup(ds, 2, 5, **kw)
    costs = []
    for _ in range(10):
        t.iterate()
        costs.append(t.cost())
    ref = g["team_%s_cost" % name]
    assert np.abs(np.array(costs) - ref).max() < 1e-7 * np.abs(ref).max()
    assert np.abs(t.global_X() - g["team_%s_X" % name]).max() < 1e-6
