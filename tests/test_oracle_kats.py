"""First-principles known-answer tests of the oracle (KATs 1-8).  These are what pins
the restatement in the absence of a buildable reference."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.util import DATA, load, random_point

R = 5


def _agent_problem(ds="smallGrid3D", N=2, a=0, **kw):
    m, mp, n = load(ds, N)
    t = O.Team(mp, n, O.default_params(r=R, num_robots=N, **kw))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(R))
    ag = t.agents[a]
    ag.build_problem(False)
    return t, ag


def test_kat1_gradient_vs_finite_differences():
    t, ag = _agent_problem()
    rng = np.random.default_rng(0)
    X = random_point(rng, R, ag.n)
    f, eg, rg = ag.eval(X)
    for _ in range(5):
        D = rng.standard_normal(X.size)
        h = 1e-6
        fd = (ag.eval(X + h * D)[0] - ag.eval(X - h * D)[0]) / (2 * h)
        assert abs(fd - eg @ D) < 1e-6 * max(1.0, abs(eg @ D))
    # the Riemannian gradient is tangent: sym(Y^T g) = 0
    Y = X.reshape(ag.n, 4, R)[:, :3, :]
    Gt = rg.reshape(ag.n, 4, R)[:, :3, :]
    S = np.einsum("nia,nja->nij", Y, Gt)
    assert np.abs(S + S.transpose(0, 2, 1)).max() < 1e-9 * np.abs(rg).max()


def test_kat2_hessian_second_order_model_and_symmetry():
    t, ag = _agent_problem()
    rng = np.random.default_rng(1)
    X = random_point(rng, R, ag.n)
    f0, _, rg = ag.eval(X)
    U = O.tangent_project(X, rng.standard_normal(X.size), R, ag.n)
    V = O.tangent_project(X, rng.standard_normal(X.size), R, ag.n)
    HU, HV = ag.hessvec(X, U), ag.hessvec(X, V)
    assert abs(U @ HV - V @ HU) < 1e-9 * abs(U @ HV)  # <U, H V> = <V, H U>
    U /= np.linalg.norm(U)
    HU = ag.hessvec(X, U)
    errs = []
    for tstep in (1e-2, 5e-3):
        # the polar retraction is second order (the QF retraction used by the solver is only first
        # order, so its pull-back differs from the Riemannian model at O(t^2))
        fr = ag.eval(O.project_manifold(X + tstep * U, R, ag.n))[0]
        model = f0 + tstep * (rg @ U) + 0.5 * tstep ** 2 * (U @ HU)
        errs.append(abs(fr - model))
    assert errs[1] < errs[0] / 6.0  # third-order remainder: halving t divides the error by ~8


def test_kat3_retraction_and_projection_stay_on_manifold():
    rng = np.random.default_rng(2)
    n = 300
    for r in (3, 4, 5, 7):
        # second projection: polishes the (possibly ill-conditioned) random draw to round-off
        X = O.project_manifold(O.project_manifold(rng.standard_normal(r * 4 * n), r, n), r, n)
        eta = O.tangent_project(X, rng.standard_normal(X.size), r, n)
        for Z in (X, O.retract(X, 0.5 * eta, r, n)):
            Y = Z.reshape(n, 4, r)[:, :3, :]
            assert np.abs(np.einsum("nia,nja->nij", Y, Y) - np.eye(3)).max() < 1e-12
        # qf retraction: first-order agreement with X + eta
        d = O.retract(X, 1e-5 * eta, r, n) - X - 1e-5 * eta
        assert np.abs(d).max() < 1e-8


@pytest.mark.parametrize("method", [O.METHOD_RTR, O.METHOD_RGD])
def test_kat4_block_coordinate_descent_is_monotone(method):
    m, mp, n = load("smallGrid3D", 3)
    t = O.Team(mp, n, O.default_params(r=R, num_robots=3, method=method, rgd_stepsize=0.2))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(R))
    prev = t.cost()
    for _ in range(30):
        t.iterate()
        c = t.cost()
        assert c <= prev + 1e-9 * abs(prev)
        prev = c


def test_kat5_gauge_invariance():
    m, _, n = load("smallGrid3D", 1)
    rng = np.random.default_rng(3)
    X = random_point(rng, R, n)
    A, _ = np.linalg.qr(rng.standard_normal((R, R)))
    AX = (A @ X.reshape(4 * n, R).T).T.reshape(-1)
    f0, f1 = O.measurement_cost(m, X, R), O.measurement_cost(m, AX, R)
    assert abs(f0 - f1) < 1e-11 * f0


def test_kat6_noise_free_graph_is_recovered():
    rng = np.random.default_rng(4)
    n, N = 40, 2
    Rs = [np.linalg.qr(rng.standard_normal((3, 3)))[0] for _ in range(n)]
    Rs = [Q * np.sign(np.linalg.det(Q)) for Q in Rs]
    ts = [3 * rng.standard_normal(3) for _ in range(n)]
    pairs = [(i, i + 1) for i in range(n - 1)] + [(int(a), int(b)) for a, b in rng.integers(0, n, (40, 2)) if a != b]
    m = np.zeros(len(pairs), dtype=O.MEAS_DTYPE)
    for k, (i, j) in enumerate(pairs):
        m[k]["p1"], m[k]["p2"] = i, j
        m[k]["R"] = (Rs[i].T @ Rs[j]).reshape(-1)
        m[k]["t"] = Rs[i].T @ (ts[j] - ts[i])
        m[k]["kappa"], m[k]["tau"], m[k]["weight"] = 50.0, 20.0, 1.0
    T = O.chordal_init(m, n)
    assert O.measurement_cost(m, O.lift(T, n, O.fixed_stiefel(3), 3), 3) < 1e-18  # chordal is exact without noise
    mp = O.partition(m, n, N)
    t = O.Team(mp, n, O.default_params(r=R, num_robots=N, gradnorm_tol=1e-10, rtr_iterations=10))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(R))
    for _ in range(12):
        t.iterate()
    assert t.cost() < 1e-16
    # exact recovery up to gauge: relative rotations of the solution equal the ground truth
    X = t.global_X().reshape(n, 4, R)
    for (i, j) in pairs[:20]:
        Yi, Yj = X[i, :3, :].T, X[j, :3, :].T
        assert np.abs(Yi.T @ Yj - Rs[i].T @ Rs[j]).max() < 1e-7


def test_kat7_sesync_optimum_sphere2500():
    """SE-Sync (Rosen et al. 2019, Table 2) reports f* = 1.6870e3 for sphere2500 with the same cost
    convention 2f = sum kappa|.|^2 + tau|.|^2; chordal initialisation cost 1.971e3."""
    m, n = O.read_g2o(os.path.join(DATA, "sphere2500.g2o"))
    T = O.chordal_init(m, n)
    X0 = O.lift(T, n, O.fixed_stiefel(R), R)
    assert abs(2 * O.measurement_cost(m, X0, R) - 1971.175) < 0.01
    t = O.Team(m, n, O.default_params(r=R, num_robots=1, rtr_iterations=20, rtr_tcg_iterations=200, gradnorm_tol=1e-6))
    t.set_initial(T, O.fixed_stiefel(R))
    t.iterate()
    assert abs(2 * t.cost() - 1687.0) < 0.05          # published, 5 significant digits
    assert abs(t.cost() - 843.5029071410438) < 1e-6   # the constant bench.py uses as f*
    assert t.agents[0].opt_result().gradnorm_opt < 1e-3  # rho test hits the fp64 floor of f1 - f2


@pytest.mark.parametrize("name,published,tol,outer", [
    ("torus3D", 2.4227e4, 0.5, 8),            # SE-Sync Table 2: 2.4227e4
    ("cubicle", 7.1713e2, 0.005, 10),         # 7.1713e2
    ("parking-garage", 1.2625e0, 1e-4, 25),   # 1.2625e0 (kappa spans 2e-9 .. 2: slow tail)
])
def test_kat7_sesync_optima_of_the_other_bundled_datasets(name, published, tol, outer):
    """SURVEY 8c KAT 7, the remaining three of the four SE-Sync optima (Rosen et al. 2019): the g2o reader's
    information-matrix -> (kappa, tau) conversion on anisotropic (cubicle) and wildly scaled (garage) inputs, the
    cost, the chordal initialisation and the RTR solve reproduce the published optimal objective values
    (2f = sum kappa|.|^2 + tau|.|^2) to the digits published.  The datasets are the reference's own data files."""
    m, n = O.read_g2o(os.path.join(DATA, name + ".g2o"))
    T = O.chordal_init(m, n)
    t = O.Team(m, n, O.default_params(r=R, num_robots=1, rtr_iterations=outer, rtr_tcg_iterations=400, gradnorm_tol=1e-6))
    t.set_initial(T, O.fixed_stiefel(R))
    t.iterate()
    assert abs(2 * t.cost() - published) < tol


def test_kat8_colour_class_order_is_irrelevant():
    """Agents without a shared edge commute: sweeps [0,2,4,1,3] and [4,0,2,3,1] give identical iterates."""
    m, mp, n = load("sphere2500", 5)
    outs = []
    for order in ([0, 2, 4, 1, 3], [4, 0, 2, 3, 1]):
        t = O.Team(mp, n, O.default_params(r=R, num_robots=5, gradnorm_tol=0.5))
        t.set_schedule(order)
        t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(R))
        for _ in range(5):
            t.iterate()
        outs.append(t.global_X())
    assert np.array_equal(outs[0], outs[1])


def _readme_demo(accel, **kw):
    """launch/dpgo_demo.launch through the wrapper: sphere2500 / 5 robots, kappa = 1e4, tau = 1e2 on every edge
    (src/utils.cpp:141-142), RTR 3 / 50 / 0.5, rel-change 0.2, odometry initial guess (README.md:32), and the
    leader alone deciding to terminate right after its own block update (src/PGOAgentROS.cpp:206-214)."""
    m, n = O.read_g2o(os.path.join(DATA, "sphere2500.g2o"), O.WEIGHT_WRAPPER)
    mp = O.partition(m.copy(), n, 5, O.WEIGHT_WRAPPER)
    p = dict(r=R, num_robots=5, gradnorm_tol=0.5, rel_change_tol=0.2, acceleration=accel, max_num_iters=1000)
    p.update(kw)
    t = O.Team(mp, n, O.default_params(**p))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(R))
    done, term, _ = t.run_schedule(1000)
    assert term
    return done


def test_readme_iteration_band():
    """README.md:44: the demo terminates after 'around 240' iterations, 'around 150' with acceleration -- the one
    solver output the reference tree states.  With the recalled trust-region radius (100, max 5x) the restatement
    terminates after 196 / 106; the count is insensitive to every other recalled constant (tCG cap, preconditioner
    shift, status rule) and sensitive to the radius alone, and an initial radius of 30 reproduces the README to the
    iteration (241 / 151).  DESIGN.md 0 has the table."""
    plain, accel = _readme_demo(0), _readme_demo(1)
    assert abs(plain - 196) <= 2 and abs(accel - 106) <= 2          # regression pins of the restatement
    assert 0.8 * 240 <= plain <= 1.2 * 240                           # README band, +-20 %
    assert 0.7 * 150 <= accel <= 1.2 * 150                           # accelerated: -29 % with radius 100
    assert 1.4 <= plain / accel <= 2.0                               # README ratio 1.6
    # the status rule does not move the count: the leader is the last robot to become ready
    assert _readme_demo(0, status_every_iterate=1) == plain
    # sensitivity to the one constant that matters
    p30 = _readme_demo(0, rtr_initial_radius=30.0, rtr_max_radius=150.0)
    a30 = _readme_demo(1, rtr_initial_radius=30.0, rtr_max_radius=150.0)
    assert abs(p30 - 240) <= 5 and abs(a30 - 150) <= 5


def test_gnc_tls_weight_function():
    p = O.default_params(r=R, num_robots=1, robust_cost_type=O.COST_GNC_TLS, gnc_barc=3.0, gnc_init_mu=0.5)
    ag = O.Agent(0, p)
    b2, mu = 9.0, 0.5
    assert ag.robust_weight(np.sqrt((mu + 1) / mu * b2) + 1e-9) == 0.0
    assert ag.robust_weight(np.sqrt(mu / (mu + 1) * b2) - 1e-9) == 1.0
    r = 3.0
    assert abs(ag.robust_weight(r) - (np.sqrt(b2 * mu * (mu + 1) / r ** 2) - mu)) < 1e-15


def test_other_robust_weight_functions():
    """the other five names src/PGOAgentROSNode.cpp:178-188 accepts, closed forms ([UPSTREAM-RECALL] RobustCost::weight;
    thresholds TLS 10, Huber 3): L2 1, L1 1 / r, Huber min(1, c / r), TLS step at c, GM 1 / (1 + r^2)^2"""
    def w(kind, r, **kw):
        return O.Agent(0, O.default_params(r=R, num_robots=1, robust_cost_type=kind, **kw)).robust_weight(r)
    for r in (0.3, 2.9999, 3.0, 7.5, 10.0, 25.0):
        assert w(O.COST_L2, r) == 1.0
        assert w(O.COST_L1, r) == 1.0 / r
        assert w(O.COST_HUBER, r) == (1.0 if r < 3.0 else 3.0 / r)
        assert w(O.COST_TLS, r) == (1.0 if r < 10.0 else 0.0)
        assert w(O.COST_GM, r) == 1.0 / ((1.0 + r * r) * (1.0 + r * r))
    assert w(O.COST_HUBER, 4.0, huber_threshold=2.0) == 0.5
    assert w(O.COST_TLS, 4.0, tls_threshold=2.0) == 0.0

