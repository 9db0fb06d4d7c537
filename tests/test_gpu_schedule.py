"""a9 on the GPU: status flags, PGOAgent::shouldTerminate() and the synchronous schedule with the leader's decisions
(src/PGOAgentROS.cpp:206-214: TERMINATE / UPDATE_WEIGHT / pass the token), HIP path vs the oracle run live."""
import os

import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import DATA, add_outliers, load, make_pair, merged_graph, params_pair

pytestmark = pytest.mark.gpu


def _status_equal(th, to, N):
    for a in range(N):
        sh, so = th.agents[a].status(), to.agents[a].status()
        assert sh.state == so.state
        assert sh.iteration_number == so.iteration_number
        assert bool(sh.ready_to_terminate) == bool(so.ready_to_terminate), (a, sh.relative_change, so.relative_change)
        assert abs(sh.relative_change - so.relative_change) < 1e-8, a


@pytest.mark.parametrize("method,accel,tol", [(capi.METHOD_RTR, 0, 0.05), (capi.METHOD_RTR, 1, 0.02),
                                              (capi.METHOD_RGD, 0, 0.05), (capi.METHOD_RGD, 1, 0.05)])
def test_status_flags_and_should_terminate_follow_the_oracle(method, accel, tol):
    """relativeChange / readyToTerminate describe each agent's last iterate(true); the leader's shouldTerminate()
    flips at the same iteration on both sides (checked every iteration, run in uneven chunks so that the status
    of an agent is read 0..N-1 iterations after its block update)."""
    N = 3
    kw = dict(method=method, acceleration=accel, rgd_stepsize=0.2, restart_interval=7, gradnorm_tol=1e-2,
              rel_change_tol=tol, max_num_iters=400)
    th, to, n = make_pair("smallGrid3D", N, **kw)
    _status_equal(th, to, N)  # before any iterate: zero-initialised status, not ready
    assert not th.should_terminate() and not to.should_terminate()
    flipped = None
    k = 0
    for chunk in [1, 1, 1, 2, 3, 1, 4, 5, 7] + [3] * 70:
        th.run(chunk)
        for _ in range(chunk):
            to.iterate()
        k += chunk
        _status_equal(th, to, N)
        st_h, st_o = th.should_terminate(), to.should_terminate()
        assert st_h == st_o, k
        if st_o and flipped is None:
            flipped = k
        if flipped is not None and k > flipped + 6:
            break
    assert flipped is not None and 20 < flipped < 220
    th.close()


@pytest.mark.parametrize("method,accel,tol", [(capi.METHOD_RTR, 0, 0.05), (capi.METHOD_RTR, 1, 0.02),
                                              (capi.METHOD_RGD, 1, 0.05)])
def test_run_schedule_terminates_where_the_oracle_does(method, accel, tol):
    N = 3
    kw = dict(method=method, acceleration=accel, rgd_stepsize=0.2, restart_interval=7, gradnorm_tol=1e-2,
              rel_change_tol=tol, max_num_iters=400)
    th, to, n = make_pair("smallGrid3D", N, **kw)
    rh, ro = th.run_schedule(500), to.run_schedule(500)
    assert rh == ro and ro[1] and ro[2] == 0
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    _status_equal(th, to, N)
    th.close()


def test_run_schedule_stops_at_max_num_iters():
    N = 3
    kw = dict(method=capi.METHOD_RTR, gradnorm_tol=1e-2, rel_change_tol=1e-9, max_num_iters=10)
    th, to, n = make_pair("smallGrid3D", N, **kw)
    rh, ro = th.run_schedule(100), to.run_schedule(100)
    assert rh == ro == (13, True, 0)  # the leader's first block update with iteration_number() > 10 is iteration 13
    th.close()


@pytest.mark.parametrize("accel", [0, 1])
def test_readme_demo_terminates_at_the_oracles_iteration(accel):
    """the demo of README.md:32,44 end to end on the device (sphere2500 / 5 robots, wrapper weighting kappa 1e4 /
    tau 1e2, RTR 3-50-0.5, rel-change 0.2, leader-only termination check) against the oracle run live here:
    196 / 106 iterations (README: around 240 / 150; tests/test_oracle_kats.py::test_readme_iteration_band)."""
    m, n = O.read_g2o(os.path.join(DATA, "sphere2500.g2o"), O.WEIGHT_WRAPPER)
    mp = O.partition(m.copy(), n, 5, O.WEIGHT_WRAPPER)
    kw = dict(r=5, num_robots=5, gradnorm_tol=0.5, rel_change_tol=0.2, acceleration=accel, max_num_iters=1000)
    ph, po = params_pair(**kw)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    rh, ro = th.run_schedule(1000), to.run_schedule(1000)
    assert ro[1] and abs(ro[0] - (106 if accel else 196)) <= 2
    assert rh == ro
    assert abs(th.cost() - to.cost()) <= 1e-7 * to.cost()
    th.close()


@pytest.mark.parametrize("ratio", [0.0, 0.97])
def test_gnc_schedule_with_leader_decisions(ratio):
    """The robust schedule as the wrapper runs it (launch/dpgo_gnc_demo.launch:35-42 scaled down): the leader calls an
    UPDATE_WEIGHT round every robust_opt_inner_iters iterations, never terminates before the last round, and
    readyToTerminate also needs robustOptMinConvergenceRatio of the loop closures at weight 0 or 1."""
    N = 3
    m, _, n = load("smallGrid3D", 1)
    mo = add_outliers(m, n, frac=0.1, seed=0)
    mp = O.partition(mo, n, N)
    T = O.odometry_init(mo, n)
    kw = dict(r=5, num_robots=N, method=capi.METHOD_RTR, gradnorm_tol=1e-2, robust_cost_type=capi.COST_GNC_TLS,
              gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-2, robust_opt_num_weight_updates=3,
              robust_opt_inner_iters=2 * N, robust_opt_min_convergence_ratio=ratio, rel_change_tol=0.05,
              max_num_iters=(3 + 1) * 2 * N - 2 if ratio == 0.0 else 200)  # Node.cpp:228-232 for the first case
    ph, po = params_pair(**kw)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    Y = O.fixed_stiefel(5)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    rh, ro = th.run_schedule(300), to.run_schedule(300)
    assert rh == ro, (rh, ro)
    assert ro[2] == 3                     # three UPDATE_WEIGHT rounds
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-6
    for a in range(N):
        wh, wo = th.agents[a].measurements(), to.agents[a].measurements()
        assert np.abs(wh["weight"] - wo["weight"]).max() < 1e-7
    _status_equal(th, to, N)
    th.close()


def test_config3_merged_graph_eight_agents_gnc():
    """BASELINE configs[3] in merged form (torus3D + cubicle + parking-garage for the absent grid3D / rim):
    ONE pose graph of 12411 poses split over 8 agents by the reference's contiguous partition rule, so agents hold
    different components (agents 0-2 torus, 3-6 cubicle, 7 cubicle tail + garage: kappa 2e-9 .. 200, anisotropic tau),
    10 % seeded outlier loop closures, GNC_TLS barc 3, mu 1e-5 x 2, two UPDATE_WEIGHT rounds."""
    N = 8
    mo, n = merged_graph()
    mp = O.partition(mo, n, N)
    T = O.odometry_init(mo, n)
    kw = dict(method=capi.METHOD_RTR, gradnorm_tol=0.5, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0,
              gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=3, robust_opt_inner_iters=8)
    ph, po = params_pair(r=5, num_robots=N, **kw)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    Y = O.fixed_stiefel(5)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    assert [th.agents[k].n for k in range(N)] == [n // N] * (N - 1) + [n - (N - 1) * (n // N)]
    # the operator the kernels apply inverts Q + 0.1 I to round-off on every component, the ill-conditioned one included
    for k in range(N):
        assert th.agents[k].preconditioner() == capi.PRECOND_TWO_LEVEL
        assert th.agents[k].preconditioner_residual() < 1e-10, k
    for rnd in range(2):
        th.run(8)
        for _ in range(8):
            to.iterate()
        # iterates, absolute (positions reach 220): measured floor per component (profiles/experiments/cfg3_parity.py) --
        # torus agents 8e-13, cubicle agents 1.3e-12, the agents that hold the garage (kappa 2e-9 .. 2, cond ~ 1e9) 6e-11,
        # with either exact form of the preconditioner (these 1551-pose agents run the two-level one)
        Xh, Xo = th.global_X(), to.global_X()
        assert np.abs(Xh - Xo).max() < 1e-9, rnd
        assert abs(th.cost() - to.cost()) <= 1e-11 * abs(to.cost())
        assert th.update_weights() == to.update_weights()
        wh = np.concatenate([th.agents[a].measurements()["weight"] for a in range(N)])
        wo = np.concatenate([to.agents[a].measurements()["weight"] for a in range(N)])
        assert np.abs(wh - wo).max() < 1e-6
        assert (wo < 1).sum() > 0.05 * len(wo)  # mu = 1e-5: anything with residual^2 > 9e-5 is down-weighted
    th.close()


@pytest.mark.parametrize("accel,restart", [(1, 50), (1, 2), (0, 50)])
def test_iterate_true_with_a_missing_neighbour_pose_keeps_x(accel, restart):
    """the delayed-message case (src/PGOAgentROS.cpp:136-149 lets a robot run iterate(true) only when its neighbours'
    poses are there; the library itself skips the solve when one is missing): X stays where it is -- it does NOT move
    to Y -- while Y, V and the periodic restart are updated as in any accelerated iteration; iterate returns false and
    the status says not ready"""
    N = 3
    m, mp, n = load("smallGrid3D", N)
    kw = dict(r=5, num_robots=N, method=capi.METHOD_RTR, acceleration=accel, restart_interval=restart, gradnorm_tol=1e-2)
    ph, po = params_pair(**kw)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    X0 = O.lift(O.odometry_init(m, n), n, O.fixed_stiefel(5), 5)
    off = 0
    for k in range(N):
        nk = to.agents[k].n
        for team in (th, to):
            team.agents[k].set_X(X0[20 * off:20 * (off + nk)])
        off += nk
    # agent 1 hears from agent 0 only; agent 2's poses never arrive
    for team in (th, to):
        for aux in ((False, True) if accel else (False,)):
            ids, P = team.agents[0].get_public_poses(1, aux)
            team.agents[1].update_neighbor_poses(0, ids, P, aux)
    for it in range(3):
        rh, ro = th.agents[1].iterate(True), to.agents[1].iterate(True)
        assert not rh and not ro
        ah, ao = th.agents[1], to.agents[1]
        assert np.abs(ah.get_X() - ao.get_X()).max() < 1e-12
        assert np.abs(ah.get_Y() - ao.get_Y()).max() < 1e-12
        assert np.abs(ah.get_V() - ao.get_V()).max() < 1e-12
        sh, so = ah.status(), ao.status()
        assert not sh.ready_to_terminate and not so.ready_to_terminate
        assert abs(sh.relative_change - so.relative_change) < 1e-12 and sh.iteration_number == so.iteration_number
    th.close()


def test_iteration_log_of_a_real_gnc_run(tmp_path):
    """dpgo_team_run_schedule writes the reference's per-robot iteration log (createIterationLog /
    logIteration / logString, src/PGOAgentROS.cpp:853-909) -- same header text, same column order, one row per block
    update of the robot, the UPDATE_WEIGHT / TERMINATE strings in every robot's file -- plus global_cost.  The logged run
    must be the un-logged run (same iterates, same decisions), and every logged figure must be the solver's own."""
    N = 3
    m, _, n = load("smallGrid3D", 1)
    mo = add_outliers(m, n, frac=0.1, seed=0)
    mp = O.partition(mo, n, N)
    T, Y = O.odometry_init(mo, n), O.fixed_stiefel(5)
    kw = dict(r=5, num_robots=N, method=capi.METHOD_RTR, gradnorm_tol=1e-2, robust_cost_type=capi.COST_GNC_TLS,
              gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-2, robust_opt_num_weight_updates=3,
              robust_opt_inner_iters=2 * N, robust_opt_min_convergence_ratio=0.97, rel_change_tol=0.05, max_num_iters=200)
    ph, po = params_pair(**kw)
    plain = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    plain.set_initial(T, Y)
    r_plain = plain.run_schedule(300)
    plain.synchronize()  # (gives the device's one-launch-solve lock back: the second team must take the same solve path)
    logged = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    logged.set_initial(T, Y)
    logged.set_iteration_log(tmp_path)
    r_log = logged.run_schedule(300)
    assert r_log == r_plain and r_log[1] and r_log[2] == 3
    assert np.array_equal(logged.global_X(), plain.global_X())
    final_cost = logged.cost()
    logged.set_iteration_log(None)
    ref_header = ("robot_id, cluster_id, num_active_robots, iteration, num_poses, bytes_received, "
                  "iter_time_sec, total_time_sec, rel_change")  # src/PGOAgentROS.cpp:863-864, verbatim
    rows_total, last = 0, None
    for a in range(N):
        lines = open(os.path.join(tmp_path, "dpgo_log_robot%d.csv" % a)).read().split("\n")
        assert lines[0] == ref_header + ", global_cost "
        body = [l for l in lines[1:] if l]
        assert body.count("UPDATE_WEIGHT") == 3 and body[-1] == "TERMINATE" and body.count("TERMINATE") == 1
        rows = [l.split(",") for l in body if l[0].isdigit()]
        assert all(len(r) == 10 for r in rows)
        its = [int(r[3]) for r in rows]
        # robot a optimizes in global iterations a + 1, a + 1 + N, ...: its iteration_number() at those moments
        assert its == [a + 1 + N * k for k in range(len(rows))]
        assert all(int(r[0]) == a and int(r[1]) == 0 and int(r[2]) == N and int(r[4]) == logged.agents[a].n for r in rows)
        br = [float(r[5]) for r in rows]
        assert br[0] > 0 and all(y > x for x, y in zip(br, br[1:]))  # cumulative payload received
        tt = [float(r[7]) for r in rows]
        assert all(y >= x for x, y in zip(tt, tt[1:])) and all(float(r[6]) > 0 for r in rows)
        st = logged.agents[a].status()
        assert abs(float(rows[-1][8]) - st.relative_change) <= 1e-15 * max(1.0, st.relative_change)
        rows_total += len(rows)
        if last is None or its[-1] > last[0]:
            last = (its[-1], float(rows[-1][9]))
    assert rows_total == r_log[0]            # one row per global iteration, in the file of the robot that optimized
    assert abs(last[1] - final_cost) <= 1e-12 * abs(final_cost)   # the last row's global_cost is the final cost
    plain.close()
    logged.close()


def _libstdcxx_uniform_draws(seed, n, length):
    """std::discrete_distribution<int> with n equal weights fed by std::mt19937(seed), as libstdc++ evaluates it:
    generate_canonical<double, 53> = (x0 + x1 * 2^32) / 2^64 from two engine outputs, then the first cumulative probability
    that is not below it.  numpy's legacy RandomState seeds its MT19937 with the same init_genrand as std::mt19937."""
    rs = np.random.RandomState(seed)
    raw = rs.randint(0, 2 ** 32, size=2 * length, dtype=np.uint64)  # raw 32-bit outputs in engine order
    cp = np.cumsum(np.full(n, 1.0 / n))[:-1]  # (libstdc++ keeps n - 1 partial sums of the normalised weights)
    out = []
    for k in range(length):
        u = (float(raw[2 * k]) + float(raw[2 * k + 1]) * 4294967296.0) / 18446744073709551616.0
        out.append(int(np.searchsorted(cp, u, side="left")))
    return out


def test_uniform_update_rule_follows_the_oracle_on_the_same_draws():
    """UpdateRule::Uniform (include/dpgo_ros/PGOAgentROS.h:35-41, the struct default; src/PGOAgentROS.cpp:446-463): token
    holders drawn with replacement by the wrapper's own recipe from a seeded engine -- the draws equal an independent replay
    of that recipe, a robot may follow itself, and the run on that order follows the oracle"""
    N = 5
    kw = dict(method=capi.METHOD_RTR, acceleration=1, restart_interval=9, gradnorm_tol=1e-2, rtr_iterations=3, rtr_tcg_iterations=50)
    th, to, n = make_pair("sphere2500", N, **kw)
    order = th.set_uniform_schedule(2024, 64)
    assert list(order) == _libstdcxx_uniform_draws(2024, N, 64)
    assert set(order) == set(range(N)) and any(order[k] == order[k + 1] for k in range(63))
    to.set_schedule(order)
    th.run(64)
    for _ in range(64):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-8
    assert abs(th.cost() - to.cost()) <= 1e-10 * abs(to.cost())
    th.close()
