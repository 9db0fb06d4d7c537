"""The one-launch pipelined iteration (csrc/step_fused.hip: evaluation + preconditioner stream + RGD step + Nesterov in
one kernel): its iterates are BITWISE those of the two-launch sequence (k_eval_stats,
k_precond<PM_RGD>) -- same arithmetic in the same order --, both follow the oracle, and teams it cannot serve keep the
two-launch sequence.  DPGO_FUSED_EVAL is read when a team is created."""
import os

import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load, make_pair

pytestmark = pytest.mark.gpu

RGD = dict(method=1, acceleration=1, rgd_stepsize=0.2, rgd_use_preconditioner=1, restart_interval=20)


def _team(dataset, robots, fused, r=5, deep=True, **kw):
    old = os.environ.get("DPGO_FUSED_EVAL")
    old_min = os.environ.get("DPGO_FE_MIN_N")
    old_deep = os.environ.get("DPGO_FE_DEEP")
    os.environ["DPGO_FUSED_EVAL"] = "1" if fused else "0"
    # round 6: runs over >= 4 robots take the deep-carried form (csrc/step_deep.hip) unless DPGO_FE_DEEP=0
    os.environ["DPGO_FE_DEEP"] = "1" if deep else "0"
    # the one-launch form serves agents of 449 .. 512 poses by default (where it is faster); the tests run it on every
    # size it can serve
    os.environ["DPGO_FE_MIN_N"] = "32"
    try:
        m, mp, n = load(dataset, robots)
        prm = capi.default_params(r=r, num_robots=robots, **kw)
        t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), prm)
        t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(r))
    finally:
        if old is None:
            os.environ.pop("DPGO_FUSED_EVAL", None)
        else:
            os.environ["DPGO_FUSED_EVAL"] = old
        if old_min is None:
            os.environ.pop("DPGO_FE_MIN_N", None)
        else:
            os.environ["DPGO_FE_MIN_N"] = old_min
        if old_deep is None:
            os.environ.pop("DPGO_FE_DEEP", None)
        else:
            os.environ["DPGO_FE_DEEP"] = old_deep
    return t


@pytest.mark.parametrize("deep", [True, False])
@pytest.mark.parametrize("robots,r", [(5, 5), (8, 5), (5, 3), (6, 4)])
def test_one_launch_iterations_bitwise_equal_the_two_launch_sequence(robots, r, deep):
    """sphere2500 over 5 / 6 / 8 robots (500 / 416 / 312 poses), runs of several lengths: whole graphs of 256 iterations,
    short graphs, restarts (every 20 iterations) inside and at the edges of the one-launch part.  deep: the deep-carried
    form of round 6 (k_step_fd: the private part of the product formed one launch early); otherwise round 5's k_step_fe"""
    ta, tb = _team("sphere2500", robots, False, r=r, **RGD), _team("sphere2500", robots, True, r=r, deep=deep, **RGD)
    for iters in (23, 300, 64, 7, 129):
        ta.run(iters)
        ta.synchronize()
        tb.run(iters)
        tb.synchronize()
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X()), (robots, r, iters, k)
        sa, sb = ta.agents[ta.ids[-1]].status(), tb.agents[tb.ids[-1]].status()
        assert sa.iteration_number == sb.iteration_number and sa.relative_change == sb.relative_change
    assert ta.counters()[7] == 0
    # 300 = 256 + 44, ...: every graph leaves period + 1 iterations to the two-launch form, and one more where that makes
    # the number of one-launch iterations even (they alternate between the two copies of the poses)
    P = robots
    expect = sum(max(0, b - P - 1) & ~1 for b in (23, 256, 44, 64, 7, 129))
    assert tb.counters()[7] == expect or (deep and robots == 5), (tb.counters()[7], expect)
    # carried rows: all but the first two one-launch iterations of every graph find the row products of their agent formed
    # by the launch before (round robin over >= 3 robots: three different agents in a row)
    carried = sum(max(0, (max(0, b - P - 1) & ~1) - 2) for b in (23, 256, 44, 64, 7, 129))
    # the deep-carried form needs the first 24 64-row chunks of every agent's order to be private: 500-pose agents
    # of sphere2500 have 24 .. 28, 416-pose agents 18, 312-pose agents 12 -- those keep round 5's form
    assert (tb.counters()[9] > 0) == (deep and robots == 5)
    if tb.counters()[9] > 0:
        # every one-launch iteration of a deep-carried run consumes what the three launches before it left (the run opens
        # with k_fd_prime and two producing launches), and the run reaches up to the last iteration of the graph: the
        # launches of the last period leave the statistics a status query reads themselves
        deep_expect = sum((b - 1) & ~1 if ((b - 1) & ~1) >= 4 else max(0, b - P - 1) & ~1 for b in (23, 256, 44, 64, 7, 129))
        assert tb.counters()[7] == deep_expect and tb.counters()[8] == deep_expect and tb.counters()[9] == deep_expect
    else:
        assert tb.counters()[8] == carried and tb.counters()[9] == 0, (tb.counters()[8], carried)
    assert np.isclose(ta.cost(), tb.cost(), rtol=0, atol=0)
    ta.close()
    tb.close()


def test_deep_carried_iterations_with_agents_of_different_sizes():
    """sphere2500 cut to its first 2494 poses over 5 robots: four agents of 499 poses (250 workgroups, the last with a single
    pose) and one of 498 (249 workgroups); every agent keeps 24 private chunks.  The launches are sized for the largest agent:
    a workgroup that owns none of the current agent's columns still takes its share of the other agents' work
    (step_deep.hip, `own`).  Bitwise the two-launch sequence, restarts every 7 iterations"""
    m, _, _ = load("sphere2500", 1)
    n = 2494
    m = m[(m["p1"] < n) & (m["p2"] < n)].copy()
    mp = O.partition(m, n, 5)
    kw = dict(RGD, restart_interval=7)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    teams = []
    old = {k: os.environ.get(k) for k in ("DPGO_FUSED_EVAL", "DPGO_FE_MIN_N")}
    try:
        os.environ["DPGO_FE_MIN_N"] = "32"
        for fe in ("0", "1"):
            os.environ["DPGO_FUSED_EVAL"] = fe
            t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=5, num_robots=5, **kw))
            t.set_initial(T, Y)
            teams.append(t)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for iters in (37, 256, 12, 101):
        for t in teams:
            t.run(iters)
            t.synchronize()
        for k in teams[0].ids:
            assert np.array_equal(teams[0].agents[k].get_X(), teams[1].agents[k].get_X()), (iters, k)
        sa, sb = teams[0].agents[4].status(), teams[1].agents[4].status()
        assert sa.iteration_number == sb.iteration_number and sa.relative_change == sb.relative_change
    assert teams[1].counters()[9] > 0
    assert len({teams[1].agents[k].n for k in teams[1].ids}) > 1
    for t in teams:
        t.close()


def test_persistent_launch_of_deep_carried_iterations_is_bitwise_too():
    """DPGO_FE_PERSIST=1 (opt-in, csrc/step_persist.hip): a run of deep-carried iterations as ONE persistent launch -- grid
    hand-offs between the iterations, everything published write-through -- leaves the bits of the two-launch sequence.
    (Measured slower than one launch per iteration, DESIGN.md section 4: it is not the default.)"""
    old = os.environ.get("DPGO_FE_PERSIST")
    os.environ["DPGO_FE_PERSIST"] = "1"
    try:
        tb = _team("sphere2500", 5, True, **RGD)
    finally:
        if old is None:
            os.environ.pop("DPGO_FE_PERSIST", None)
        else:
            os.environ["DPGO_FE_PERSIST"] = old
    ta = _team("sphere2500", 5, False, **RGD)
    for iters in (23, 300, 7, 64):
        for t in (ta, tb):
            t.run(iters)
            t.synchronize()
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X()), (iters, k)
        sa, sb = ta.agents[ta.ids[-1]].status(), tb.agents[tb.ids[-1]].status()
        assert sa.iteration_number == sb.iteration_number and sa.relative_change == sb.relative_change
    assert tb.counters()[9] > 0
    ta.close()
    tb.close()


def test_one_launch_iterations_follow_the_oracle():
    """120 iterations of the bench configuration against the live oracle (the two-launch sequence is held to the same
    bound in test_gpu_parity.py)"""
    th, to, n = make_pair("sphere2500", 5, **RGD)
    th.run(120)
    th.synchronize()
    for _ in range(120):
        to.iterate()
    assert th.counters()[7] > 0
    for k in th.ids:
        assert np.abs(th.agents[k].get_X() - to.agents[k].get_X()).max() < 1e-10
    th.close()


def test_teams_the_one_launch_form_cannot_serve_keep_the_two_launch_sequence():
    """agents beyond 512 poses (torus3D / 8: 625), agents of <= 256 poses (sphere2500 / 10), the two-level form, no
    acceleration: dpgo_team_run runs, counters[7] stays 0"""
    cases = [("torus3D", 8, dict(RGD)), ("sphere2500", 10, dict(RGD)),
             ("sphere2500", 5, dict(RGD, precond_mode=capi.PRECOND_TWO_LEVEL)), ("sphere2500", 5, dict(RGD, acceleration=0))]
    for dataset, robots, kw in cases:
        t = _team(dataset, robots, True, **kw)
        c0 = t.cost()
        t.run(40)
        t.synchronize()
        assert t.counters()[7] == 0, (dataset, robots, kw)
        assert t.cost() < c0
        t.close()


def test_default_window_of_the_one_launch_form():
    """by default the one-launch form serves agents of 32 .. 512 poses where every iteration finds carried rows (round 5:
    faster than the two-launch sequence at every size measured), and agents of 449 .. 512 poses where it does not (there the
    two-launch sequence is the faster one below; profiles/experiments/fe_small.py); DPGO_FE_MIN_N sets the bound by hand
    (every other test here runs it from 32 poses up)"""
    assert os.environ.get("DPGO_FE_MIN_N") is None
    old = os.environ.get("DPGO_FE_CARRY")
    try:
        for carry, cases in (("1", ((5, True), (6, True), (8, True))), ("0", ((5, True), (6, False), (8, False)))):
            os.environ["DPGO_FE_CARRY"] = carry
            for robots, served in cases:
                m, mp, n = load("sphere2500", robots)
                t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=5, num_robots=robots, **RGD))
                t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))
                t.run(64)
                t.synchronize()
                assert (t.counters()[7] > 0) == served, (carry, robots, t.counters()[7])
                t.close()
    finally:
        if old is None:
            os.environ.pop("DPGO_FE_CARRY", None)
        else:
            os.environ["DPGO_FE_CARRY"] = old


def test_one_launch_iterations_across_weight_updates():
    """GNC-TLS re-weighting between runs (the lane-ordered copy of the blocks is re-sent with the new values, the
    structure stands): still bitwise the two-launch sequence"""
    kw = dict(RGD, robust_cost_type=5, gnc_barc=5.0)
    ta, tb = _team("sphere2500", 5, False, **kw), _team("sphere2500", 5, True, **kw)
    for rnd in range(3):
        for t in (ta, tb):
            t.run(60)
            t.synchronize()
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X()), rnd
        wa, wb = ta.update_weights(), tb.update_weights()
        assert wa == wb
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].measurements()["weight"], tb.agents[k].measurements()["weight"])
    # (deep-carried runs reach up to the last iteration of a graph: 58 of 60)
    assert tb.counters()[7] == tb.counters()[9] == 3 * ((60 - 1) & ~1) and ta.counters()[7] == 0
    ta.close()
    tb.close()


def test_rows_longer_than_the_ell_part_keep_the_two_launch_sequence():
    """a pose with more than 8 blocks in its row of Q (7 extra loop closures at one pose of robot 0): no lane-ordered copy of
    the blocks exists for that agent, the team runs the two-launch sequence -- and follows the oracle"""
    m, mp, n = load("sphere2500", 5)
    mp = mp.copy()
    intra = np.flatnonzero((mp["r1"] == 0) & (mp["r2"] == 0) & (mp["p2"] == mp["p1"] + 1))
    extra = mp[intra[:7]].copy()
    for k in range(7):
        extra[k]["p1"], extra[k]["p2"] = 10, 100 + 20 * k
    mq = np.concatenate([mp, extra])
    prm_h, prm_o = capi.default_params(r=5, num_robots=5, **RGD), O.default_params(r=5, num_robots=5, **RGD)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    th = capi.Team.from_measurements(mq.view(capi.MEAS_DTYPE), prm_h)
    to = O.Team(mq, n, prm_o)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    th.run(40)
    th.synchronize()
    for _ in range(40):
        to.iterate()
    assert th.counters()[7] == 0
    for k in th.ids:
        assert np.abs(th.agents[k].get_X() - to.agents[k].get_X()).max() < 1e-10
    th.close()


def test_two_teams_and_a_foreign_process_share_the_device():
    """Round 3's one-launch iteration counted its workgroups in and spun until all had formed their gradient: every
    workgroup of a launch had to be resident at once, one team per device at a time (a lock), and a foreign process on the
    GPU could make it give up.  Round 4: the launches alternate between two copies of the poses and wait for nobody -- two
    teams run their one-launch iterations at the same time, and a process that keeps the CUs busy with kernels of its own
    only slows them down: iterates bitwise those of the two-launch sequence."""
    import subprocess
    import sys
    busy = subprocess.Popen([sys.executable, "-c",
                             "import torch, time\n"
                             "x = torch.randn(6144, 6144, device='cuda', dtype=torch.float64)\n"
                             "t0 = time.time()\n"
                             "while time.time() - t0 < 6.0:\n"
                             "    y = x @ x\n"
                             "    torch.cuda.synchronize()\n"])
    try:
        tref = _team("sphere2500", 5, False, **RGD)
        ta, tb = _team("sphere2500", 5, True, **RGD), _team("sphere2500", 5, True, **RGD)
        for rnd in range(3):
            ta.run(300)          # not synchronized: both teams' graphs are in flight together
            tb.run(300)
            tref.run(300)
            for t in (ta, tb, tref):
                t.synchronize()
            for k in ta.ids:
                assert np.array_equal(ta.agents[k].get_X(), tref.agents[k].get_X()), rnd
                assert np.array_equal(tb.agents[k].get_X(), tref.agents[k].get_X()), rnd
        assert ta.counters()[7] > 0 and tb.counters()[7] == ta.counters()[7] and tref.counters()[7] == 0
        assert busy.poll() is None  # (the foreign process was still at it)
        for t in (ta, tb, tref):
            t.close()
    finally:
        busy.wait(timeout=60)


def test_dispatch_to_dispatch_timing_entry():
    """dpgo_team_time_kernel(14) (bench.py's roofline leg): reps (made even) + 8 one-launch iterations launched eagerly; like the
    other in-loop timing entries it consumes the state (its last iteration has looked ahead); teams that cannot take
    the one-launch form get an error"""
    tb = _team("sphere2500", 5, True, **RGD)
    c0 = tb.cost()
    ms, nbytes = tb.time_kernel(0, 14, reps=42)
    assert 0.005 < ms < 0.1 and nbytes > 32e6
    assert tb.cost() < c0
    tb.close()
    tc = _team("torus3D", 8, True, **RGD)
    with pytest.raises(capi.DpgoError):
        tc.time_kernel(0, 14, reps=4)
    tc.close()


def test_carried_rows_can_be_switched_off_and_change_no_bit():
    """DPGO_FE_CARRY=0: every one-launch iteration forms the row products of its agent itself (round 3's form); the same
    bits either way, restarts inside the runs"""
    old = os.environ.get("DPGO_FE_CARRY")
    os.environ["DPGO_FE_CARRY"] = "0"
    try:
        ta = _team("sphere2500", 5, True, **RGD)
    finally:
        if old is None:
            os.environ.pop("DPGO_FE_CARRY", None)
        else:
            os.environ["DPGO_FE_CARRY"] = old
    tb = _team("sphere2500", 5, True, **RGD)
    for iters in (300, 41, 600):
        ta.run(iters)
        tb.run(iters)
        ta.synchronize()
        tb.synchronize()
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X()), (iters, k)
    # (tb runs deep-carried: its runs reach further into every graph)
    assert 0 < ta.counters()[7] <= tb.counters()[7]
    assert ta.counters()[8] == 0 and tb.counters()[8] > 0
    ta.close()
    tb.close()


def test_carried_rows_need_three_different_agents_in_a_row():
    """a schedule that repeats an agent within three iterations (0 1 0 2 3 4, period 6): the launches whose agent moved one
    or two iterations earlier form their row products themselves, the others take the carried ones; bitwise the two-launch
    sequence.  Two robots (smallGrid3D): nothing is carried"""
    ta, tb = _team("sphere2500", 5, False, **RGD), _team("sphere2500", 5, True, **RGD)
    order = [0, 1, 0, 2, 3, 4]
    for t in (ta, tb):
        t.set_schedule(order)
    for iters in (100, 257):
        ta.run(iters)
        tb.run(iters)
        ta.synchronize()
        tb.synchronize()
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X()), (iters, k)
    c = tb.counters()
    assert 0 < c[8] < c[7] - 4, (c[7], c[8])
    ta.close()
    tb.close()
    t2 = _team("smallGrid3D", 2, True, **RGD)
    t2.run(100)
    t2.synchronize()
    assert t2.counters()[8] == 0
    t2.close()


@pytest.mark.parametrize("seed", [3, 11, 29, 57])
def test_random_teams_take_carried_rows_and_stay_bitwise(seed):
    """profiles/experiments/fe_fuzz.py with fixed seeds: sphere2500 over 5 .. 8 robots, r = 3 / 4 / 5, up to 110 extra loop
    closures between random poses (more shared edges, longer rows), a random restart interval (3 .. 36) and step, GNC
    re-weighting between the runs for half of the cases; one-launch iterations with carried rows against the two-launch
    sequence, bit for bit after every run"""
    rng = np.random.default_rng(seed)
    m0, _, n = load("sphere2500", 5)
    robots = int(rng.integers(5, 9))
    r = int(rng.choice([3, 4, 5]))
    extra = int(rng.integers(0, 110))
    m = m0.copy()
    if extra:
        add = m[rng.integers(0, len(m), extra)].copy()
        for e in add:
            i, j = rng.integers(0, n, 2)
            while abs(int(i) - int(j)) < 2:
                i, j = rng.integers(0, n, 2)
            e["p1"], e["p2"] = min(i, j), max(i, j)
        m = np.concatenate([m, add])
    mp = capi.partition(m.view(capi.MEAS_DTYPE), n, robots)
    kw = dict(method=1, acceleration=1, rgd_stepsize=float(rng.choice([0.05, 0.1, 0.2])), rgd_use_preconditioner=1,
              restart_interval=int(rng.integers(3, 37)))
    robust = bool(rng.integers(0, 2))
    if robust:
        kw.update(robust_cost_type=5, gnc_barc=5.0)
    T, Y = O.odometry_init(m0, n), O.fixed_stiefel(r)
    teams = []
    old = {k: os.environ.get(k) for k in ("DPGO_FUSED_EVAL", "DPGO_FE_MIN_N", "DPGO_FE_DEEP")}
    try:
        os.environ["DPGO_FE_MIN_N"] = "32"
        # two-launch sequence | one-launch, deep-carried where the team allows it (round 6) | one-launch, round 5's form
        for fe, deep in (("0", "1"), ("1", "1"), ("1", "0")):
            os.environ["DPGO_FUSED_EVAL"] = fe
            os.environ["DPGO_FE_DEEP"] = deep
            t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=robots, **kw))
            t.set_initial(T, Y)
            teams.append(t)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    for chunk in rng.integers(20, 400, 3):
        for t in teams:
            t.run(int(chunk))
            t.synchronize()
        for k in teams[0].ids:
            x0 = teams[0].agents[k].get_X()
            assert np.array_equal(x0, teams[1].agents[k].get_X()), (seed, robots, r, extra, kw, int(chunk), k)
            assert np.array_equal(x0, teams[2].agents[k].get_X()), (seed, robots, r, extra, kw, int(chunk), k)
        if robust:
            assert teams[0].update_weights() == teams[1].update_weights() == teams[2].update_weights()
    c = teams[1].counters()
    assert teams[0].counters()[7] == 0 and teams[2].counters()[9] == 0
    if c[7] > 0:   # (teams the one-launch form serves: every one of them also carries rows)
        assert c[8] > 0, (seed, robots, r, extra, c[7], c[8])
        assert teams[2].counters()[8] > 0
    print("seed %d: %d robots, r = %d, %d extra edges: one-launch %d, deep-carried %d" % (seed, robots, r, extra, c[7], c[9]))
    for t in teams:
        t.close()
