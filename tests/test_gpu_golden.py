"""The HIP path against the committed golden vectors of the INDEPENDENT numpy implementation
(oracle/np_crosscheck.py -> tests/golden/*.npz), directly -- the C oracle is not in the loop here.
Same assertions and tolerances as tests/test_golden.py holds the C oracle to (dense LU vs dense Cholesky
inverse, SVD vs Newton-Schulz polar factors); every call goes through the C-ABI."""
import os

import numpy as np
import pytest

from dpgo_ros_amd import capi
from tests.util import DATA, ROOT

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, "tests", "golden")


def _setup(ds, N, r, **kw):
    g = np.load(os.path.join(GOLD, "%s_N%d_r%d.npz" % (ds, N, r)))
    m, n = capi.read_g2o(os.path.join(DATA, ds + ".g2o"))
    T = capi.odometry_init(m, n)
    assert np.abs(T - g["T0"]).max() < 1e-12
    mp = capi.partition(m, n, N)
    t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=N, **kw))
    t.set_initial(T, capi.fixed_stiefel(r))
    return g, t


def _dense_Q(rowptr, col, val, n):
    Q = np.zeros((4 * n, 4 * n))
    for j in range(n):
        for p in range(rowptr[j], rowptr[j + 1]):
            Q[4 * col[p]:4 * col[p] + 4, 4 * j:4 * j + 4] = val[16 * p:16 * p + 16].reshape(4, 4, order="F")
    return Q


def _manifold(t, fn, n, *arrays):
    out = np.zeros_like(arrays[0])
    args = [capi._d(np.ascontiguousarray(a)) for a in arrays]
    capi._chk(getattr(capi.lib(), fn)(t.h, *args, n, capi._d(out)), fn)
    return out


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D"])
def test_hip_problem_surface_vs_numpy(ds):
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
        assert np.abs(_manifold(t, "dpgo_retract", ag.n, X, 0.3 * eta) - g["a%d_retract" % a]).max() < 1e-12
        assert np.abs(_manifold(t, "dpgo_project_manifold", ag.n, X + 0.2 * V) - g["a%d_project" % a]).max() < 1e-12
    t.close()


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D"])
def test_hip_single_solves_vs_numpy(ds):
    for method, key in ((capi.METHOD_RGD, "rgd"), (capi.METHOD_RTR, "rtr")):
        for a in range(2):
            g, t = _setup(ds, 2, 5, method=method, rgd_stepsize=0.2, gradnorm_tol=1e-2)
            ag = t.agents[a]
            assert ag.iterate(True)
            assert np.abs(ag.get_X() - g["a%d_%s" % (a, key)]).max() < 1e-7
            if key == "rtr":
                res = ag.opt_result()
                assert res.tcg_iters_total == int(g["a%d_rtr_tcg" % a]) and res.accepted == int(g["a%d_rtr_acc" % a])
            t.close()


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D"])
@pytest.mark.parametrize("name,kw", [
    ("rtr", dict(method=0)),
    ("rtr_acc", dict(method=0, acceleration=1, restart_interval=7)),
    ("rgd_acc", dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=7)),
])
def test_hip_ten_rbcd_iterations_vs_numpy(ds, name, kw):
    g, t = _setup(ds, 2, 5, **kw)
    costs = []
    for _ in range(10):
        t.run(1)
        costs.append(t.cost())
    ref = g["team_%s_cost" % name]
    assert np.abs(np.array(costs) - ref).max() < 1e-7 * np.abs(ref).max()
    assert np.abs(t.global_X() - g["team_%s_X" % name]).max() < 1e-6
    t.close()


@pytest.mark.parametrize("name,kw", [
    ("rtr_acc", dict(method=0, acceleration=1, restart_interval=7)),
    ("rgd_acc", dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=7)),
])
def test_hip_ten_iterations_in_one_run_vs_numpy(name, kw):
    """the same ten iterations as ONE device-resident run (pipelined graph for RGD): final iterate vs numpy"""
    g, t = _setup("smallGrid3D", 2, 5, **kw)
    t.run(10)
    assert abs(t.cost() - g["team_%s_cost" % name][-1]) < 1e-7 * abs(g["team_%s_cost" % name][-1])
    assert np.abs(t.global_X() - g["team_%s_X" % name]).max() < 1e-6
    t.close()
