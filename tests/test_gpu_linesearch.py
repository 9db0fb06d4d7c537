"""RGD with a backtracking (Armijo) line search (SURVEY App. B "a backtracking variant exists"; north_star "RTR/RGD line
search"): the HIP path -- all trial points, all trial costs from one pass over Q, decision on the device
(csrc/linesearch.hip) -- against the oracle's sequential back-off loop (oracle/orc_core.c): iterates, the number of
back-offs of every block update (bit-exact), costs."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import make_pair

pytestmark = pytest.mark.gpu


def _follow(th, to, iters, tol):
    hist = []
    for k in range(iters):
        th.run(1)
        sel = to.iterate()
        rh, ro = th.agents[sel].opt_result(), to.agents[sel].opt_result()
        assert (rh.ls_backoffs, rh.accepted) == (ro.ls_backoffs, ro.accepted), (k, sel, rh.ls_backoffs, ro.ls_backoffs)
        assert abs(rh.f_init - ro.f_init) <= 1e-9 * abs(ro.f_init) and abs(rh.f_opt - ro.f_opt) <= 1e-9 * abs(ro.f_opt)
        assert abs(rh.gradnorm_opt - ro.gradnorm_opt) <= 1e-7 * max(1.0, ro.gradnorm_opt)
        hist.append(ro.ls_backoffs)
        if k % 7 == 6:
            assert np.abs(th.global_X() - to.global_X()).max() < tol, k
    return hist


@pytest.mark.parametrize("accel,precond,step", [(0, 1, 1.0), (1, 1, 1.6), (0, 0, 1e-3), (1, 1, 0.2)])
def test_small_grid_two_agents(accel, precond, step):
    kw = dict(method=capi.METHOD_RGD, rgd_stepsize=step, rgd_use_preconditioner=precond, acceleration=accel,
              restart_interval=7, rgd_line_search=1)
    th, to, n = make_pair("smallGrid3D", 2, **kw)
    f0 = to.cost()
    hist = _follow(th, to, 40, 1e-9)
    assert abs(th.cost() - to.cost()) <= 1e-10 * abs(to.cost()) and to.cost() < f0
    if step >= 1.0:
        assert max(hist) >= 1  # the search did back off somewhere
    for k in range(2):
        assert abs(th.agents[k].status().relative_change - to.agents[k].status().relative_change) < 1e-9
    th.close()


def test_every_trial_rejected_keeps_x():
    """a step no back-off can rescue within the allowed trials: x stays put, accepted = 0"""
    kw = dict(method=capi.METHOD_RGD, rgd_stepsize=4096.0, rgd_use_preconditioner=1, rgd_line_search=1, rgd_ls_max_backoffs=2)
    th, to, n = make_pair("smallGrid3D", 2, **kw)
    X0 = to.global_X().copy()
    hist = _follow(th, to, 4, 1e-12)
    assert hist == [3, 3, 3, 3]
    assert np.array_equal(to.global_X(), X0) and np.abs(th.global_X() - X0).max() == 0.0
    th.close()


@pytest.mark.parametrize("step", [0.2, 1.0])
def test_sphere2500_five_agents(step):
    """the bench workload with the safeguard, at the launch default restart interval (launch/PGOAgent.launch:25)"""
    kw = dict(method=capi.METHOD_RGD, rgd_stepsize=step, acceleration=1, restart_interval=50, rgd_line_search=1)
    th, to, n = make_pair("sphere2500", 5, **kw)
    hist = _follow(th, to, 120, 1e-8)
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    if step == 1.0:
        assert max(hist) >= 2
    # a longer stretch through captured graphs (run(k) replays the un-fused sequence), compared at the end
    th.run(200)
    for _ in range(200):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    th.close()


def test_agent_api_and_colour_classes():
    """the per-agent API (what a ROS wrapper drives) and the colour-parallel sweep take the same search"""
    from tests.util import load
    kw = dict(method=capi.METHOD_RGD, rgd_stepsize=1.0, rgd_line_search=1)
    m, mp, n = load("smallGrid3D", 2)
    ph, po = capi.default_params(r=5, num_robots=2, **kw), O.default_params(r=5, num_robots=2, **kw)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    th.run_colored(3)  # two agents, two classes: the sequential order
    for _ in range(6):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-10
    for b in range(2):  # (the sweep read its neighbours in place: the slabs the per-agent calls read are filled by messages)
        for c in th.agents[b].neighbors():
            ids, poses = th.agents[b].get_public_poses(c)
            th.agents[c].update_neighbor_poses(b, ids, poses)
    for it in range(6):  # per-agent calls with host exchange
        sel = it % 2
        for b in range(2):
            th.agents[b].iterate(b == sel)
        to.iterate()
        for b in range(2):
            for c in th.agents[b].neighbors():
                ids, poses = th.agents[b].get_public_poses(c)
                th.agents[c].update_neighbor_poses(b, ids, poses)
        rh, ro = th.agents[sel].opt_result(), to.agents[sel].opt_result()
        assert (rh.ls_backoffs, rh.accepted) == (ro.ls_backoffs, ro.accepted)
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-10
    th.close()
