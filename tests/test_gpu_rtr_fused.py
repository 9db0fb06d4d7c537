"""The one-launch RTR solve (dpgo_ros_amd/csrc/rtr_fused.hip: persistent kernel, the agent's dense preconditioner resident in
LDS) against (1) the launch-per-step kernel sequence it replaces, selected with DPGO_FUSED_RTR=0, and (2) the oracle -- over
the lifted ranks the templates are instantiated for, agents with an odd pose count (a workgroup that owns ONE pose), rows
longer than the cached ELL slots, trust-region rejections and boundary steps (small initial radius)."""
import ctypes as C
import os

import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load, params_pair

pytestmark = pytest.mark.gpu


def _team(mp, prm, fused):
    old = os.environ.get("DPGO_FUSED_RTR")
    os.environ["DPGO_FUSED_RTR"] = "1" if fused else "0"
    try:
        return capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), prm)
    finally:
        if old is None:
            del os.environ["DPGO_FUSED_RTR"]
        else:
            os.environ["DPGO_FUSED_RTR"] = old


def _handoffs(team, agent):
    """epoch of the agent's grid hand-off counters: > 0 iff one-launch solves ran (and passed that many hand-offs)"""
    out = (C.c_ulonglong * 320)()
    rc = capi.lib().dpgo_agent_read_rtr_handoff(team.h, agent, out, 320)
    return int(out[17 * 16]) if rc == 0 else 0


@pytest.mark.parametrize("r", [3, 4, 5, 6, 8])
def test_one_launch_solve_equals_the_launch_per_step_sequence_and_the_oracle(r):
    N, iters = 3, 12
    kw = dict(method=capi.METHOD_RTR, acceleration=1, restart_interval=5, gradnorm_tol=1e-3, rtr_iterations=3, rtr_tcg_iterations=30)
    m, mp, n = load("smallGrid3D", N)
    ph, po = params_pair(r=r, num_robots=N, **kw)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(r)
    tf, ts = _team(mp, ph, True), _team(mp, ph, False)
    to = O.Team(mp, n, po)
    for t in (tf, ts, to):
        t.set_initial(T, Y)
    tf.run(iters)
    ts.run(iters)
    for _ in range(iters):
        to.iterate()
    assert _handoffs(tf, 0) > 0 and _handoffs(ts, 0) == 0
    assert np.abs(tf.global_X() - ts.global_X()).max() < 1e-9
    assert np.abs(tf.global_X() - to.global_X()).max() < 1e-7
    assert abs(tf.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    for a in range(N):
        rf, rs = tf.agents[a].opt_result(), ts.agents[a].opt_result()
        assert (rf.rtr_outer_iters, rf.tcg_iters_total, rf.accepted) == (rs.rtr_outer_iters, rs.tcg_iters_total, rs.accepted)
        assert abs(rf.f_opt - rs.f_opt) <= 1e-10 * abs(rs.f_opt)
    tf.close()
    ts.close()


@pytest.mark.parametrize("dataset,N,radius", [("parking-garage", 5, 100.0), ("sphere2500", 5, 0.5), ("sphere2500", 7, 100.0)])
def test_one_launch_solve_odd_agents_long_rows_and_rejections(dataset, N, radius):
    """parking-garage / 5: 333- and 332-pose agents (odd: the last workgroup owns one pose); sphere2500 / 7: 358- and
    357-pose agents; radius 0.5: the first solves hit the trust-region boundary and take rejected steps"""
    iters = 2 * N
    kw = dict(method=capi.METHOD_RTR, acceleration=0, gradnorm_tol=1e-2, rtr_iterations=3, rtr_tcg_iterations=8, rtr_initial_radius=radius)
    m, mp, n = load(dataset, N)
    ph, po = params_pair(r=5, num_robots=N, **kw)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    tf = _team(mp, ph, True)
    to = O.Team(mp, n, po)
    tf.set_initial(T, Y)
    to.set_initial(T, Y)
    tf.run(iters)
    for _ in range(iters):
        to.iterate()
    assert _handoffs(tf, 0) > 0
    scale = max(1.0, np.abs(to.global_X()).max())
    assert np.abs(tf.global_X() - to.global_X()).max() < 1e-7 * scale
    assert abs(tf.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    for a in range(N):
        rf, ro = tf.agents[a].opt_result(), to.agents[a].opt_result()
        assert (rf.rtr_outer_iters, rf.tcg_iters_total, rf.accepted) == (ro.rtr_outer_iters, ro.tcg_iters_total, ro.accepted)
    tf.close()


def test_graphs_with_the_schedule_baked_in_equal_the_device_selected_ones():
    """DPGO_BAKE_SEL: the pipelined iteration's graphs address every launch's agent from a kernel argument (one graph per
    schedule phase) or through the device-side schedule state -- the same launches, bitwise the same iterates, also for
    run lengths that leave the schedule phase and the graph window misaligned"""
    m, mp, n = load("sphere2500", 5)
    ph, _ = params_pair(r=5, num_robots=5, method=capi.METHOD_RGD, rgd_stepsize=0.2, acceleration=1, restart_interval=20)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    X = []
    for bake in ("1", "0"):
        old = os.environ.get("DPGO_BAKE_SEL")
        os.environ["DPGO_BAKE_SEL"] = bake
        try:
            t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
        finally:
            if old is None:
                del os.environ["DPGO_BAKE_SEL"]
            else:
                os.environ["DPGO_BAKE_SEL"] = old
        t.set_initial(T, Y)
        for k in (7, 64, 131, 3, 200):
            t.run(k)
        X.append(t.global_X().copy())
        st = [t.agents[a].status() for a in range(5)]
        X.append(np.array([s.relative_change for s in st]))
        t.close()
    assert np.array_equal(X[0], X[2]) and np.array_equal(X[1], X[3])


@pytest.mark.parametrize("dataset,N,r,accel,mode", [
    ("torus3D", 8, 5, 1, capi.PRECOND_AUTO),        # 625-pose agents: too large for the dense slabs, automatic mode -> two-level
    ("sphere2500", 5, 5, 0, capi.PRECOND_TWO_LEVEL),
    ("sphere2500", 4, 3, 1, capi.PRECOND_AUTO),     # 625-pose agents, rank 3
    ("smallGrid3D", 2, 8, 1, capi.PRECOND_TWO_LEVEL),
])
def test_one_launch_solve_with_the_two_level_preconditioner(dataset, N, r, accel, mode):
    """agents whose 8-column slabs of the dense inverse do not fit LDS (more than 512 poses) keep the solve in ONE launch
    with the two-level form of the preconditioner resident in LDS instead (rtr_fused.hip, k_rtr_solve<R, true>: 128-thread
    workgroups, two per CU, one more grid hand-off per apply): the oracle's tCG / acceptance counts and iterates, and the
    launch-per-step sequence's"""
    iters = 2 * N
    kw = dict(method=capi.METHOD_RTR, acceleration=accel, restart_interval=5, gradnorm_tol=1e-2, rtr_iterations=3,
              rtr_tcg_iterations=12, precond_mode=mode)
    m, mp, n = load(dataset, N)
    ph, po = params_pair(r=r, num_robots=N, **kw)
    po.precond_mode = 0
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(r)
    tf, ts = _team(mp, ph, True), _team(mp, ph, False)
    to = O.Team(mp, n, po)
    for t in (tf, ts, to):
        t.set_initial(T, Y)
    assert all(a.preconditioner() == capi.PRECOND_TWO_LEVEL for a in tf.agents.values())
    # one team at a time on the device: the one-launch solve is a persistent kernel whose workgroups wait for each other
    # (two per CU here), and a second stream's launches in between can keep part of its grid from becoming resident
    tf.run(iters)
    tf.synchronize()
    ts.run(iters)
    for _ in range(iters):
        to.iterate()
    assert _handoffs(tf, 0) > 0 and _handoffs(ts, 0) == 0
    scale = max(1.0, np.abs(to.global_X()).max())
    assert np.abs(tf.global_X() - ts.global_X()).max() < 1e-8 * scale
    assert np.abs(tf.global_X() - to.global_X()).max() < 1e-7 * scale
    assert abs(tf.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    for a in range(N):
        rf, ro = tf.agents[a].opt_result(), to.agents[a].opt_result()
        assert (rf.rtr_outer_iters, rf.tcg_iters_total, rf.accepted) == (ro.rtr_outer_iters, ro.tcg_iters_total, ro.accepted)
    tf.close()
    ts.close()


@pytest.mark.parametrize("dataset,N,r,accel", [("torus3D", 8, 5, 1), ("sphere2500", 4, 3, 0), ("sphere2500", 4, 5, 1)])
def test_one_launch_dense_solve_with_three_poses_per_workgroup(dataset, N, r, accel):
    """agents of 513 .. 640 poses whose DENSE preconditioner is asked for (precond_mode 1) keep the solve in one launch:
    three poses = 12 columns of M per workgroup, 7 of them in LDS and 5 in the lanes' own registers (rtr_fused.hip,
    k_rtr_solve<R, false, 3>).  The oracle's counts and iterates, and the launch-per-step sequence's"""
    iters = 2 * N
    kw = dict(method=capi.METHOD_RTR, acceleration=accel, restart_interval=5, gradnorm_tol=1e-2, rtr_iterations=3,
              rtr_tcg_iterations=12, precond_mode=capi.PRECOND_DENSE)
    m, mp, n = load(dataset, N)
    ph, po = params_pair(r=r, num_robots=N, **kw)
    po.precond_mode = 0
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(r)
    tf, ts = _team(mp, ph, True), _team(mp, ph, False)
    to = O.Team(mp, n, po)
    for t in (tf, ts, to):
        t.set_initial(T, Y)
    assert all(a.preconditioner() == capi.PRECOND_DENSE for a in tf.agents.values()) and 512 < n // N <= 640
    tf.run(iters)
    tf.synchronize()
    ts.run(iters)
    for _ in range(iters):
        to.iterate()
    assert _handoffs(tf, 0) > 0 and _handoffs(ts, 0) == 0
    scale = max(1.0, np.abs(to.global_X()).max())
    assert np.abs(tf.global_X() - ts.global_X()).max() < 1e-8 * scale
    assert np.abs(tf.global_X() - to.global_X()).max() < 1e-7 * scale
    for a in range(N):
        rf, ro = tf.agents[a].opt_result(), to.agents[a].opt_result()
        assert (rf.rtr_outer_iters, rf.tcg_iters_total, rf.accepted) == (ro.rtr_outer_iters, ro.tcg_iters_total, ro.accepted)
    tf.close()
    ts.close()


def test_h_delta_ring_wraps_within_one_solve():
    """H delta of tCG iteration k sits in slot k mod 32 of a ring that is read with ordinary (cached) loads: launches of
    more than 32 tCG iterations reuse slots (one agent-scope acquire per wrap), and slots of agents whose vectors are not
    whole cache lines long are padded -- a line shared by two slots served stale bytes of the second (found by this file's
    first test on 41-pose agents when the ring was introduced).  Many outer iterations per solve at a tolerance the first
    block updates are far from, so that the counts do not hinge on round-off at a stop test."""
    for dataset, N, r, outer in (("smallGrid3D", 3, 5, 8), ("smallGrid3D", 3, 3, 8), ("sphere2500", 7, 5, 12)):
        kw = dict(method=capi.METHOD_RTR, acceleration=0, gradnorm_tol=1e-3, rtr_iterations=outer, rtr_tcg_iterations=30)
        m, mp, n = load(dataset, N)
        ph, po = params_pair(r=r, num_robots=N, **kw)
        T, Y = O.odometry_init(m, n), O.fixed_stiefel(r)
        tf, ts = _team(mp, ph, True), _team(mp, ph, False)
        to = O.Team(mp, n, po)
        for t in (tf, ts, to):
            t.set_initial(T, Y)
        most = 0
        for k in range(N):
            tf.run(1)
            ts.run(1)
            sel = to.iterate()
            rf, rs, ro = tf.agents[sel].opt_result(), ts.agents[sel].opt_result(), to.agents[sel].opt_result()
            assert (rf.rtr_outer_iters, rf.tcg_iters_total, rf.accepted) == (rs.rtr_outer_iters, rs.tcg_iters_total, rs.accepted)
            assert (rf.rtr_outer_iters, rf.tcg_iters_total, rf.accepted) == (ro.rtr_outer_iters, ro.tcg_iters_total, ro.accepted)
            most = max(most, rf.tcg_iters_total)
        assert most > 34, (dataset, most)  # the ring did wrap
        assert _handoffs(tf, 0) > 0
        assert np.abs(tf.global_X() - ts.global_X()).max() < 1e-9
        assert np.abs(tf.global_X() - to.global_X()).max() < 1e-7
        tf.close()
        ts.close()
