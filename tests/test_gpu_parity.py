"""GPU parity tests proper: every call goes through the C-ABI (libdpgo_hip.so) and is compared with
the CPU oracle on the same seeded inputs.  Tolerances are fp64 round-off level per operation
(summation order differs) and are written next to each assertion."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load, make_pair, params_pair, random_point, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scratch_team():
    t = capi.Team(capi.default_params(r=5, num_robots=1), [0])
    yield t
    t.close()


@pytest.mark.parametrize("r", [3, 4, 5, 6, 8])
def test_manifold_ops(r):
    rng = np.random.default_rng(r)
    n = 257
    t = capi.Team(capi.default_params(r=r, num_robots=1), [0])
    # well-conditioned inputs, as in the algorithm (combinations of nearby Stiefel points):
    # the polar factor's sensitivity is cond(A)^2 * eps
    X = O.project_manifold(rng.standard_normal(r * 4 * n), r, n) + 0.2 * rng.standard_normal(r * 4 * n)
    out = np.zeros_like(X)
    capi._chk(capi.lib().dpgo_project_manifold(t.h, capi._d(X), n, capi._d(out)), "project")
    ref = O.project_manifold(X, r, n)
    assert relerr(out, ref) < 1e-12
    Y = out.reshape(n, 4, r)[:, :3, :]
    gram = np.einsum("nia,nja->nij", Y, Y)
    assert np.abs(gram - np.eye(3)).max() < 1e-12  # KAT 3: Y^T Y = I
    V = rng.standard_normal(r * 4 * n)
    tp = np.zeros_like(X)
    capi._chk(capi.lib().dpgo_tangent_project(t.h, capi._d(ref), capi._d(V), n, capi._d(tp)), "tangent")
    assert relerr(tp, O.tangent_project(ref, V, r, n)) < 1e-13
    eta = 0.3 * O.tangent_project(ref, V, r, n)
    rt = np.zeros_like(X)
    capi._chk(capi.lib().dpgo_retract(t.h, capi._d(ref), capi._d(eta), n, capi._d(rt)), "retract")
    assert relerr(rt, O.retract(ref, eta, r, n)) < 1e-13
    t.close()


@pytest.mark.parametrize("dataset,N,r", [("tinyGrid3D", 2, 5), ("smallGrid3D", 2, 5), ("smallGrid3D", 3, 3),
                                         ("sphere2500", 5, 5)])
def test_problem_surface(dataset, N, r):
    """Q, G, f, EucGrad, RieGrad, Hess-vec and preconditioner of every agent vs the oracle."""
    th, to, n = make_pair(dataset, N, r=r)
    rng = np.random.default_rng(7)
    for k in range(N):
        ah, ao = th.agents[k], to.agents[k]
        assert ah.n == ao.n
        assert ah.neighbors() == ao.neighbors()
        for nb in ao.neighbors():
            assert np.array_equal(ah.public_pose_ids(nb), ao.public_pose_ids(nb))
            assert np.array_equal(ah.neighbor_pose_ids(nb), ao.neighbor_pose_ids(nb))
        ah.build_problem(False)
        ao.build_problem(False)
        rp_h, col_h, val_h = ah.get_Q()
        rp_o, col_o, val_o = ao.get_Q()
        assert np.array_equal(rp_h, rp_o) and np.array_equal(col_h, col_o)  # index work: bit exact
        assert relerr(val_h, val_o) < 1e-15
        assert relerr(ah.get_G(), ao.get_G()) < 1e-14
        X = random_point(rng, r, ah.n)
        fh, egh, rgh = ah.eval(X)
        fo, ego, rgo = ao.eval(X)
        assert abs(fh - fo) <= 1e-12 * abs(fo)
        assert relerr(egh, ego) < 1e-13 and relerr(rgh, rgo) < 1e-13
        eta = O.tangent_project(X, rng.standard_normal(X.size), r, ah.n)
        assert relerr(ah.hessvec(X, eta), ao.hessvec(X, eta)) < 1e-13
        V = rng.standard_normal(X.size)
        # dense inverse vs sparse Cholesky: cond(Q + 0.1 I) ~ 1e5..1e6 bounds the agreement
        assert relerr(ah.precondition(X, V), ao.precondition(X, V)) < 1e-9
    th.close()


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("method", [capi.METHOD_RGD, capi.METHOD_RTR])
def test_agent_api_iterates(method, accel):
    """Drive both implementations through the per-agent API exactly as PGOAgentROS does:
    iterate(false) on everyone else, publish, iterate(true) on the token holder, publish."""
    N, r = 2, 5
    kw = dict(method=method, acceleration=accel, rgd_stepsize=0.2, restart_interval=7, gradnorm_tol=1e-2)
    th, to, n = make_pair("smallGrid3D", N, r=r, **kw)

    def publish(team, b, with_aux):
        a = team.agents[b]
        for c in a.neighbors():
            ids, P = a.get_public_poses(c, False)
            team.agents[c].update_neighbor_poses(b, ids, P, False)
            if with_aux:
                ids, P = a.get_public_poses(c, True)
                team.agents[c].update_neighbor_poses(b, ids, P, True)

    for k in range(16):
        sel = k % N
        for team in (th, to):
            for b in range(N):
                if b != sel:
                    team.agents[b].iterate(False)
                    if accel:
                        publish(team, b, True)
            assert team.agents[sel].iterate(True)
            publish(team, sel, bool(accel))
        for b in range(N):
            # iterates: 1e-9 absolute after k RBCD iterations (preconditioner solves differ at 1e-10)
            assert np.abs(th.agents[b].get_X() - to.agents[b].get_X()).max() < 1e-8, (k, b)
            sh, so = th.agents[b].status(), to.agents[b].status()
            assert abs(sh.relative_change - so.relative_change) < 1e-8
        rh, ro = th.agents[sel].opt_result(), to.agents[sel].opt_result()
        assert abs(rh.f_init - ro.f_init) <= 1e-9 * abs(ro.f_init)
        assert abs(rh.f_opt - ro.f_opt) <= 1e-9 * abs(ro.f_opt)
        assert abs(rh.gradnorm_init - ro.gradnorm_init) <= 1e-7 * max(1.0, ro.gradnorm_init)
        if method == capi.METHOD_RTR:
            assert rh.tcg_iters_total == ro.tcg_iters_total and rh.accepted == ro.accepted
    th.close()


@pytest.mark.parametrize("accel", [0, 1])
@pytest.mark.parametrize("method", [capi.METHOD_RGD, capi.METHOD_RTR])
def test_report_on_the_last_launch_is_the_report_kernel_bit_for_bit(method, accel, monkeypatch):
    """The report of an RGD iterate(true) (public poses of both sequences, status, fInit / fOpt / gradient norms: what
    src/PGOAgentROS.cpp:160-172 and :666-668 read behind the call) rides on the call's last launch, the closing statistics
    evaluation (k_eval_report), instead of a launch of its own (k_report; DPGO_REPORT_TAIL=0).  One single-agent team per
    robot as the wrapper runs them, sphere2500 / 5, restart interval 7 (restart iterations keep k_report): every published
    pose, every status, every result and the final iterates are bitwise equal, and the counter says which form ran.  An RTR
    iterate(true) keeps the report kernel (measured: profiles/r06_agent_api.md) behind a solve that carries the iteration's
    tail; the switch must not change a bit of it."""
    N, r = 5, 5
    m, mp, n = load("sphere2500", N)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(r)
    prm = capi.default_params(r=r, num_robots=N, method=method, acceleration=accel, rgd_stepsize=0.2, restart_interval=7,
                              gradnorm_tol=1e-2)
    sets = []
    for tail in ("1", "0"):
        monkeypatch.setenv("DPGO_REPORT_TAIL", tail)
        teams = [capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), prm, device=0, local_ids=[a]) for a in range(N)]
        for a in range(N):
            teams[a].set_initial(T, Y, offsets=np.array([a * (n // N)], dtype=np.int32))
        sets.append(teams)
    log = [[], []]

    def publish(k, ags, b):
        for c in ags[b].neighbors():
            for aux in ((False, True) if accel else (False,)):
                ids, P = ags[b].get_public_poses(c, aux)
                log[k].append(np.array(P, copy=True))
                ags[c].update_neighbor_poses(b, ids, P, aux)

    for k, teams in enumerate(sets):
        ags = [teams[a].agents[a] for a in range(N)]
        for b in range(N):
            publish(k, ags, b)
        for it in range(23):
            sel = it % N
            for b in range(N):
                if b != sel:
                    ags[b].iterate(False)
                    log[k].append(ags[b].status().relative_change)
            for b in range(N):
                if b != sel and ags[b].publish_requested(True):
                    publish(k, ags, b)
            assert ags[sel].iterate(True)
            st, res = ags[sel].status(), ags[sel].opt_result()
            log[k] += [st.relative_change, res.f_init, res.f_opt, res.gradnorm_init, res.gradnorm_opt, res.tcg_iters_total, res.accepted]
            if ags[sel].publish_requested(True):
                publish(k, ags, sel)
    assert len(log[0]) == len(log[1])
    for u, v in zip(log[0], log[1]):
        assert np.array_equal(np.asarray(u), np.asarray(v))
    for a in range(N):
        assert np.array_equal(sets[0][a].agents[a].get_X(), sets[1][a].agents[a].get_X())
    folded = sum(t.counters()[10] for t in sets[0])
    # (23 block updates; accelerated: 3 of them restarts)
    assert folded == (0 if method == capi.METHOD_RTR else 23 - (3 if accel else 0)), folded
    assert sum(t.counters()[10] for t in sets[1]) == 0
    for teams in sets:
        for t in teams:
            t.close()


@pytest.mark.parametrize("dataset,N,method,accel,iters", [
    ("smallGrid3D", 2, capi.METHOD_RGD, 0, 30),
    ("smallGrid3D", 2, capi.METHOD_RGD, 1, 30),
    ("smallGrid3D", 2, capi.METHOD_RTR, 0, 12),
    ("smallGrid3D", 2, capi.METHOD_RTR, 1, 12),
    ("sphere2500", 5, capi.METHOD_RGD, 1, 60),
    ("sphere2500", 5, capi.METHOD_RTR, 0, 10),
])
def test_team_run_matches_oracle(dataset, N, method, accel, iters):
    kw = dict(method=method, acceleration=accel, rgd_stepsize=0.1, restart_interval=25, gradnorm_tol=1e-2)
    th, to, n = make_pair(dataset, N, **kw)
    th.run(iters)
    for _ in range(iters):
        to.iterate()
    Xh, Xo = th.global_X(), to.global_X()
    assert np.abs(Xh - Xo).max() < 1e-7
    fh, fo = th.cost(), to.cost()
    assert abs(fh - fo) <= 1e-9 * abs(fo)
    th.close()


@pytest.mark.parametrize("N,restart,schedule,splits", [
    (3, 25, None, (40,)),                 # one long window, look-ahead on every iteration but the last
    (3, 4, None, (1, 2, 3, 7, 11)),       # windows of 3 fused iterations between restarts, ragged run() calls
    (3, 6, [0, 0, 1, 2, 2, 1], (5, 19)),  # an agent selected twice in a row: its look-ahead step keeps V
    (2, 5, [1], (9,)),                    # the same agent every time
])
def test_pipelined_rgd_windows(N, restart, schedule, splits):
    """dpgo_team_run for accelerated RGD: [restart iteration] + fused iterations in one graph per window, two
    launches per fused iteration with the Nesterov step of k+1 taken inside the step kernel of k.  Iterates,
    per-agent status (relative change from the look-ahead partials) and optimisation results after every run()
    call against the oracle's one-iteration-at-a-time schedule."""
    kw = dict(method=capi.METHOD_RGD, acceleration=1, rgd_stepsize=0.1, restart_interval=restart)
    th, to, n = make_pair("smallGrid3D", N, **kw)
    if schedule is not None:
        th.set_schedule(schedule)
        to.set_schedule(schedule)
    done = 0
    for cnt in splits:
        th.run(cnt)
        sel = None
        for _ in range(cnt):
            sel = to.iterate()
        done += cnt
        assert np.abs(th.global_X() - to.global_X()).max() < 1e-8, done
        for a in range(N):
            sh, so = th.agents[a].status(), to.agents[a].status()
            assert sh.iteration_number == so.iteration_number == done
            assert abs(sh.relative_change - so.relative_change) < 1e-8, (done, a)
            for which in ("get_Y", "get_V"):
                assert np.abs(getattr(th.agents[a], which)() - getattr(to.agents[a], which)()).max() < 1e-8, (done, a, which)
        rh, ro = th.agents[sel].opt_result(), to.agents[sel].opt_result()
        assert abs(rh.f_init - ro.f_init) <= 1e-9 * abs(ro.f_init) and abs(rh.f_opt - ro.f_opt) <= 1e-9 * abs(ro.f_opt)
        assert abs(rh.gradnorm_opt - ro.gradnorm_opt) <= 1e-7 * max(1.0, ro.gradnorm_opt)
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    th.close()


def test_accelerated_rgd_beyond_the_lookahead_limit():
    """teams of more than 8 agents keep the 3-launch iteration and the per-window graphs ([restart iteration] +
    fused iterations up to the next restart): torus3D over 10 agents, restart every 7 iterations"""
    kw = dict(method=capi.METHOD_RGD, acceleration=1, rgd_stepsize=0.05, restart_interval=7)
    th, to, n = make_pair("torus3D", 10, **kw)
    done = 0
    for cnt in (3, 22):
        th.run(cnt)
        for _ in range(cnt):
            to.iterate()
        done += cnt
        assert np.abs(th.global_X() - to.global_X()).max() < 1e-7, done
        for a in range(10):
            assert abs(th.agents[a].status().relative_change - to.agents[a].status().relative_change) < 1e-8
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    th.close()


def _two_rank_colored_worker(rank, world, port, outdir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dpgo_ros_amd.distributed import DistributedRBCD, HipBackend, owner_of
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, r = 4, 5
    kw = dict(method=capi.METHOD_RTR, acceleration=0, gradnorm_tol=1e-2)
    m, mp, n = load("smallGrid3D", N)
    mine = [a for a in range(N) if owner_of(a, world) == rank]
    be = HipBackend(mp.view(capi.MEAS_DTYPE), capi.default_params(r=r, num_robots=N, **kw), mine, 0, torch, host_staging=True)
    per = n // N
    be.team.set_initial(O.odometry_init(m, n), O.fixed_stiefel(r), offsets=np.array([a * per for a in mine], dtype=np.int32))
    drv = DistributedRBCD(dist, be, mp, N, 0, rank, world)
    drv.exchange_all()
    for _ in range(3):
        drv.sweep_colored()
    cost = drv.global_cost(torch, "cpu")
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), cost=cost, groups=np.array([a for g in drv.groups for a in g]),
             **{"X%d" % a: be.team.agents[a].get_X() for a in mine})
    dist.barrier()
    be.close()
    dist.destroy_process_group()


def test_two_processes_colour_parallel_sweeps():
    """multi-process colour-parallel RBCD: members of a class on different ranks update concurrently."""
    import socket
    import tempfile
    import torch.multiprocessing as mp_
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_two_rank_colored_worker, args=(2, port, d), nprocs=2, join=True)
        outs = [np.load(d + "/rank%d.npz" % r) for r in range(2)]
    N = 4
    m, mp, n = load("smallGrid3D", N)
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, method=O.METHOD_RTR, gradnorm_tol=1e-2))
    ref.set_schedule(outs[0]["groups"])
    ref.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))
    for _ in range(3 * N):
        ref.iterate()
    for a in range(N):
        assert np.abs(outs[a % 2]["X%d" % a] - ref.agents[a].get_X()).max() < 1e-8
    assert abs(float(outs[0]["cost"]) - ref.cost()) <= 1e-9 * ref.cost()


def _two_rank_worker(rank, world, port, outdir):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dpgo_ros_amd.distributed import DistributedRBCD, HipBackend, owner_of
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, r = 3, 5
    kw = dict(method=capi.METHOD_RGD, acceleration=1, rgd_stepsize=0.2, restart_interval=6)
    m, mp, n = load("smallGrid3D", N)
    mine = [a for a in range(N) if owner_of(a, world) == rank]
    be = HipBackend(mp.view(capi.MEAS_DTYPE), capi.default_params(r=r, num_robots=N, **kw), mine, 0, torch, host_staging=True)
    per = n // N
    be.team.set_initial(O.odometry_init(m, n), O.fixed_stiefel(r), offsets=np.array([a * per for a in mine], dtype=np.int32))
    drv = DistributedRBCD(dist, be, mp, N, 1, rank, world)
    drv.exchange_all()
    for _ in range(14):
        drv.step()
    cost = drv.global_cost(torch, "cpu")
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), cost=cost, **{"X%d" % a: be.team.agents[a].get_X() for a in mine})
    dist.barrier()
    be.close()
    dist.destroy_process_group()


def test_two_processes_one_gpu_exchange_through_device_slabs():
    """The N > 1 driver with the HIP backend: two processes share cuda:0, slabs are packed/unpacked by
    the device kernels of the C-ABI and bounced through gloo (RCCL refuses two ranks on one GPU)."""
    import socket
    import tempfile
    import torch.multiprocessing as mp_
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_two_rank_worker, args=(2, port, d), nprocs=2, join=True)
        outs = [np.load(d + "/rank%d.npz" % r) for r in range(2)]
    N = 3
    m, mp, n = load("smallGrid3D", N)
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, method=O.METHOD_RGD, acceleration=1, rgd_stepsize=0.2,
                                         restart_interval=6))
    ref.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))
    for _ in range(14):
        ref.iterate()
    for a in range(N):
        assert np.abs(outs[a % 2]["X%d" % a] - ref.agents[a].get_X()).max() < 1e-8
    assert abs(float(outs[0]["cost"]) - ref.cost()) <= 1e-9 * ref.cost()


def _pair_from(meas_partitioned, n_total, N, T, **kw):
    ph, po = params_pair(r=5, num_robots=N, **kw)
    th = capi.Team.from_measurements(meas_partitioned.view(capi.MEAS_DTYPE), ph)
    to = O.Team(meas_partitioned, n_total, po)
    Y = O.fixed_stiefel(5)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    return th, to


def test_gnc_tls_reweighting_rounds():
    """a8: residual kernel -> GNC-TLS weights (owner = lower-ID endpoint) -> weight hand-over to the
    higher-ID endpoint (optionally rounded to float32 as on the wire, msg/RelativeMeasurementWeights.msg:8)
    -> Q / G / dense preconditioner rebuilt; compared with the oracle after every UPDATE_WEIGHT round."""
    from tests.util import add_outliers
    N = 3
    m, _, n = load("smallGrid3D", 1)
    mo = add_outliers(m, n, frac=0.1, seed=0)
    mp = O.partition(mo, n, N)
    T = O.odometry_init(mo, n)
    for f32 in (0, 1):
        kw = dict(method=capi.METHOD_RTR, gradnorm_tol=1e-2, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0,
                  gnc_mu_step=2.0, gnc_init_mu=1e-2, robust_opt_num_weight_updates=3, robust_opt_inner_iters=6,
                  weights_as_float32=f32)
        th, to = _pair_from(mp, n, N, T, **kw)
        for rnd in range(3):
            th.run(6)
            for _ in range(6):
                to.iterate()
            assert np.abs(th.global_X() - to.global_X()).max() < 1e-7, rnd
            ch_h, ch_o = th.update_weights(), to.update_weights()
            assert ch_h == ch_o
            for a in range(N):
                wh, wo = th.agents[a].measurements(), to.agents[a].measurements()
                assert np.array_equal(wh["p1"], wo["p1"]) and np.array_equal(wh["fixed_weight"], wo["fixed_weight"])
                assert np.abs(wh["weight"] - wo["weight"]).max() < 1e-7
                assert 0 < (wo["weight"] < 1).sum()  # the outliers are being down-weighted
        th.run(4)
        for _ in range(4):
            to.iterate()
        assert np.abs(th.global_X() - to.global_X()).max() < 1e-6
        assert abs(th.cost() - to.cost()) <= 1e-7 * abs(to.cost())
        th.close()


@pytest.mark.parametrize("dataset,N,method", [("torus3D", 8, capi.METHOD_RTR), ("sphere2500", 5, capi.METHOD_RGD), ("long_rows", 1, capi.METHOD_RTR)])
def test_weight_update_refreshes_the_layouts_of_q_on_the_device_bit_for_bit(dataset, N, method, monkeypatch):
    """An UPDATE_WEIGHT round (same pattern, new values) re-creates Q's ELL copy, its CSR tail and the lane-ordered copy of the
    one-launch iteration ON THE DEVICE from the block-CSR values it uploads (k_q_layouts) instead of laying them out on the
    host and uploading them (DPGO_HOST_LAYOUTS=1): pure copies, so three GNC rounds with iterations between them end in
    bitwise the same iterates and weights (two-level agents, dense agents with the one-launch iteration, rows with a CSR
    tail)."""
    if dataset == "long_rows":  # (the graph of test_long_rows_take_the_csr_tail: rows of up to 13 blocks)
        rng = np.random.default_rng(5)
        n = 40
        pairs = [(i, i + 1) for i in range(n - 1)] + [(3, j) for j in range(6, 30, 2)] + [(j, 17) for j in range(20, 38, 3)]
        m = np.zeros(len(pairs), dtype=O.MEAS_DTYPE)
        for k, (i, j) in enumerate(pairs):
            Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            Q *= np.sign(np.linalg.det(Q))
            m[k]["p1"], m[k]["p2"] = i, j
            m[k]["R"], m[k]["t"] = Q.reshape(-1), rng.standard_normal(3)
            m[k]["kappa"], m[k]["tau"], m[k]["weight"] = 30.0 + k, 7.0, 1.0
        mp = m
    else:
        m, mp, n = load(dataset, N)
    kw = dict(method=method, acceleration=1 if method == capi.METHOD_RGD else 0, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0,
              gradnorm_tol=0.5, rgd_stepsize=0.2)
    outs = []
    for host in ("1", None):
        if host:
            monkeypatch.setenv("DPGO_HOST_LAYOUTS", host)
        else:
            monkeypatch.delenv("DPGO_HOST_LAYOUTS", raising=False)
        t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=5, num_robots=N, **kw))
        t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))
        for rnd in range(3):
            t.run(3 * N)
            t.update_weights()
        t.run(2 * N)
        outs.append(([t.agents[a].get_X() for a in t.ids], [t.agents[a].measurements()["weight"].copy() for a in t.ids]))
        t.close()
    for xa, xb in zip(outs[0][0], outs[1][0]):
        assert np.array_equal(xa, xb)
    for wa, wb in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(wa, wb)


@pytest.mark.parametrize("kind,kw", [(capi.COST_L1, {}), (capi.COST_HUBER, dict(huber_threshold=1.0)),
                                     (capi.COST_TLS, dict(tls_threshold=2.5)), (capi.COST_GM, {})])
def test_other_robust_cost_types_reweighting_rounds(kind, kw):
    """the robust cost types besides L2 / GNC_TLS that the node accepts (src/PGOAgentROSNode.cpp:178-188): the weight function
    in closed form through the C-ABI, then two UPDATE_WEIGHT rounds against the oracle (weights, iterates, cost)"""
    from tests.util import add_outliers
    N = 3
    m, _, n = load("smallGrid3D", 1)
    mo = add_outliers(m, n, frac=0.1, seed=1)
    mp = O.partition(mo, n, N)
    T = O.odometry_init(mo, n)
    prm = dict(method=capi.METHOD_RTR, gradnorm_tol=1e-2, robust_cost_type=kind, robust_opt_num_weight_updates=2,
               robust_opt_inner_iters=6, **kw)
    th, to = _pair_from(mp, n, N, T, **prm)
    for r in (0.25, 0.999, 1.0, 2.5, 3.0, 9.0, 40.0):
        expect = {capi.COST_L1: 1.0 / r, capi.COST_HUBER: 1.0 if r < 1.0 else 1.0 / r, capi.COST_TLS: 1.0 if r < 2.5 else 0.0,
                  capi.COST_GM: 1.0 / ((1.0 + r * r) ** 2)}[kind]
        assert th.agents[0].robust_weight(r) == expect and to.agents[0].robust_weight(r) == expect
    for rnd in range(2):
        th.run(6)
        for _ in range(6):
            to.iterate()
        assert np.abs(th.global_X() - to.global_X()).max() < 1e-7, rnd
        assert th.update_weights() == to.update_weights()
        moved = 0
        for a in range(N):
            wh, wo = th.agents[a].measurements(), to.agents[a].measurements()
            assert np.abs(wh["weight"] - wo["weight"]).max() <= 1e-7 * max(1.0, np.abs(wo["weight"]).max())
            moved += int((wo["weight"] != 1).sum())
        assert moved > 0
    th.run(4)
    for _ in range(4):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-6
    assert abs(th.cost() - to.cost()) <= 1e-7 * abs(to.cost())
    th.close()


@pytest.mark.parametrize("mode,accel", [(capi.WEIGHT_WRAPPER, 0), (capi.WEIGHT_LIBRARY, 1)])
def test_tunnels_eight_agents(mode, accel):
    """BASELINE configs[4] inputs (8 robots, all-to-all neighbours, 700-1000 shared edges per agent) under
    the synchronous schedule; per-robot odometry chains as the (deliberately crude) common initial guess."""
    from tests.util import load_tunnels
    m = load_tunnels(mode)
    N = 8
    if mode == capi.WEIGHT_LIBRARY:
        m = m.copy(); m["weight"] = 1.0
    nk = [0] * N
    for e in m:
        nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1)
        nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
    Ts = []
    for k in range(N):
        odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
        odo["r1"] = 0; odo["r2"] = 0
        Ts.append(O.odometry_init(odo, nk[k]))
    T = np.concatenate(Ts)
    kw = dict(method=capi.METHOD_RTR, gradnorm_tol=1e-2, acceleration=accel, restart_interval=11)
    th, to = _pair_from(m, sum(nk), N, T, **kw)
    assert [th.agents[k].n for k in range(N)] == nk
    assert all(len(th.agents[k].neighbors()) == 7 for k in range(N))
    f0 = to.cost()
    th.run(16)
    for _ in range(16):
        to.iterate()
    # measured on the MI355X (profiles/experiments/tolerance_floor.py): 4.5e-13 / 5.1e-13 on iterates of magnitude 260,
    # 2e-15 / 6e-15 relative on the cost
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-10
    assert abs(th.cost() - to.cost()) <= 1e-12 * abs(to.cost()) and to.cost() < f0
    th.close()
    if accel:
        # the same inputs through the pipelined accelerated-RGD windows: 8 agents of 105..191 poses, so a
        # workgroup's look-ahead share (up to 22 poses) straddles agents
        kw = dict(method=capi.METHOD_RGD, rgd_stepsize=0.05, acceleration=1, restart_interval=9)
        th, to = _pair_from(m, sum(nk), N, T, **kw)
        for cnt in (13, 20):
            th.run(cnt)
            for _ in range(cnt):
                to.iterate()
            assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
            for k in range(N):
                assert abs(th.agents[k].status().relative_change - to.agents[k].status().relative_change) < 1e-8
        th.close()


def test_long_rows_take_the_csr_tail():
    """a pose with more than 8 blocks in its row exercises the CSR tail behind the 8-slot ELL part."""
    rng = np.random.default_rng(5)
    n = 40
    pairs = [(i, i + 1) for i in range(n - 1)] + [(3, j) for j in range(6, 30, 2)] + [(j, 17) for j in range(20, 38, 3)]
    m = np.zeros(len(pairs), dtype=O.MEAS_DTYPE)
    for k, (i, j) in enumerate(pairs):
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        Q *= np.sign(np.linalg.det(Q))
        m[k]["p1"], m[k]["p2"] = i, j
        m[k]["R"], m[k]["t"] = Q.reshape(-1), rng.standard_normal(3)
        m[k]["kappa"], m[k]["tau"], m[k]["weight"] = 30.0 + k, 7.0, 1.0
    T = O.odometry_init(m, n)
    th, to = _pair_from(m, n, 1, T, method=capi.METHOD_RTR, gradnorm_tol=1e-6, rtr_iterations=5)
    ah, ao = th.agents[0], to.agents[0]
    ah.build_problem(False); ao.build_problem(False)
    rp, _, _ = ah.get_Q()
    assert np.diff(rp).max() > 8
    X = random_point(rng, 5, n)
    fh, egh, rgh = ah.eval(X)
    fo, ego, rgo = ao.eval(X)
    assert abs(fh - fo) <= 1e-12 * abs(fo) and relerr(egh, ego) < 1e-13 and relerr(rgh, rgo) < 1e-13
    eta = O.tangent_project(X, rng.standard_normal(X.size), 5, n)
    assert relerr(ah.hessvec(X, eta), ao.hessvec(X, eta)) < 1e-13
    th.run(3)
    for _ in range(3):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-8
    th.close()


@pytest.mark.parametrize("r", [3, 4, 6, 8])
@pytest.mark.parametrize("method", [capi.METHOD_RGD, capi.METHOD_RTR])
def test_other_relaxation_ranks(r, method):
    kw = dict(method=method, acceleration=1, rgd_stepsize=0.2, restart_interval=5, gradnorm_tol=1e-2)
    th, to, n = make_pair("smallGrid3D", 2, r=r, **kw)
    th.run(9)
    for _ in range(9):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    th.close()


def test_asapp_style_seeded_random_order_on_tunnels():
    """BASELINE configs[4] stand-in: the asynchronous ASAPP run is nondeterministic in the
    reference (Poisson clocks), so it is replayed as a SEEDED random activation order: each event is one
    preconditioned RGD step (stepsize 0.2, README.md:52) of one agent with whatever neighbour poses are
    current, no acceleration (launch/asapp_demo.launch)."""
    from tests.util import load_tunnels
    m = load_tunnels(capi.WEIGHT_WRAPPER)
    N = 8
    nk = [0] * N
    for e in m:
        nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1)
        nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
    Ts = []
    for k in range(N):
        odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
        odo["r1"] = 0; odo["r2"] = 0
        Ts.append(O.odometry_init(odo, nk[k]))
    T = np.concatenate(Ts)
    order = np.random.default_rng(2024).integers(0, N, 64).astype(np.int32)
    th, to = _pair_from(m, sum(nk), N, T, method=capi.METHOD_RGD, rgd_stepsize=0.2, acceleration=0)
    th.set_schedule(order)
    to.set_schedule(order)
    f0 = to.cost()
    th.run(64)
    for _ in range(64):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost()) and to.cost() < f0
    th.close()


def test_config2_sphere2500_eight_agents_rtr():
    """BASELINE configs[2]: sphere2500 over 8 agents (7 x 312 + 316 poses), RTR 3/50/0.5 (launch/dpgo_demo.launch:33-35)."""
    kw = dict(method=capi.METHOD_RTR, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5)
    th, to, n = make_pair("sphere2500", 8, **kw)
    assert [th.agents[k].n for k in range(8)] == [312] * 7 + [316]
    th.run(16)
    for _ in range(16):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    for k in range(8):
        rh, ro = th.agents[k].opt_result(), to.agents[k].opt_result()
        assert rh.tcg_iters_total == ro.tcg_iters_total and rh.accepted == ro.accepted
    th.close()


def test_config3_torus_eight_agents_gnc_with_outliers():
    """BASELINE configs[3] with the declared substitution (grid3D / rim are absent): torus3D split over
    8 agents + seeded synthetic outlier loop closures, GNC_TLS barc 3, mu 1e-5 x2 (launch/dpgo_gnc_demo.launch:35-42)."""
    from tests.util import add_outliers
    N = 8
    m, _, n = load("torus3D", 1)
    mo = add_outliers(m, n, frac=0.02, seed=0)
    mp = O.partition(mo, n, N)
    T = O.odometry_init(mo, n)
    kw = dict(method=capi.METHOD_RTR, gradnorm_tol=0.5, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0,
              gnc_mu_step=2.0, gnc_init_mu=1e-5, robust_opt_num_weight_updates=3, robust_opt_inner_iters=8)
    th, to = _pair_from(mp, n, N, T, **kw)
    assert [th.agents[k].n for k in range(N)] == [625] * 8
    for rnd in range(2):
        th.run(8)
        for _ in range(8):
            to.iterate()
        # measured (profiles/experiments/tolerance_floor.py): iterates 4.7e-14 / 2.6e-13, weights 1e-16 / 3.5e-16
        assert np.abs(th.global_X() - to.global_X()).max() < 1e-10, rnd
        assert th.update_weights() == to.update_weights()
        wh = np.concatenate([th.agents[a].measurements()["weight"] for a in range(N)])
        wo = np.concatenate([to.agents[a].measurements()["weight"] for a in range(N)])
        assert np.abs(wh - wo).max() < 1e-12
    th.close()


def test_full_size_properties_sphere2500():
    """Size-independent properties at the bench workload's full size, HIP path only:
    monotone descent of plain RBCD, iterates stay on the manifold, gauge invariance of the cost."""
    m, mp, n = load("sphere2500", 5)
    t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=5, num_robots=5, method=capi.METHOD_RTR, gradnorm_tol=0.5))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))
    prev = t.cost()
    for _ in range(25):
        t.run(1)
        c = t.cost()
        assert c <= prev * (1 + 1e-12)
        prev = c
    X = t.global_X()
    Y = X.reshape(n, 4, 5)[:, :3, :]
    assert np.abs(np.einsum("nia,nja->nij", Y, Y) - np.eye(3)).max() < 1e-12
    A, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((5, 5)))
    AX = (A @ X.reshape(4 * n, 5).T).T.reshape(-1)
    for a in range(5):
        t.agents[a].set_X(AX[5 * 4 * 500 * a:5 * 4 * 500 * (a + 1)])
    t.exchange_all()
    assert abs(t.cost() - prev) <= 1e-10 * prev
    t.close()


@pytest.mark.parametrize("dataset,N,method,fused_rtr", [("sphere2500", 5, capi.METHOD_RTR, 1), ("sphere2500", 8, capi.METHOD_RGD, 1),
                                                         ("smallGrid3D", 3, capi.METHOD_RTR, 1), ("sphere2500", 5, capi.METHOD_RTR, 0)])
def test_colour_parallel_sweeps_equal_the_permuted_sequential_schedule(dataset, N, method, fused_rtr, monkeypatch):
    """agents of one colour class update together (RGD, and RTR without the one-launch solve: in the same
    launches; RTR with it: one one-launch solve after the other -- the members share no edge); the result must equal the
    sequential schedule [class 0 ..., class 1 ...] -- bitwise on the HIP path, to tolerance vs the oracle."""
    monkeypatch.setenv("DPGO_FUSED_RTR", str(fused_rtr))   # (read when a team is created)
    kw = dict(method=method, acceleration=0, rgd_stepsize=0.2, gradnorm_tol=1e-2)
    th, to, n = make_pair(dataset, N, **kw)
    nc, col = th.coloring()
    order = [a for c in range(nc) for a in range(N) if col[a] == c]
    assert nc == 2 and sorted(order) == list(range(N))  # chain of agents: two colours
    sweeps = 4
    th.run_colored(sweeps)
    to.set_schedule(order)
    for _ in range(sweeps * N):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    assert th.iteration() == sweeps * N
    ts, _, _ = make_pair(dataset, N, **kw)
    ts.set_schedule(order)
    ts.run(sweeps * N)
    # RGD, launch-per-step RTR: same kernels, same order of arithmetic per agent.  One-launch RTR solves: the sequential
    # schedule folds the iteration's tail into the solve kernel, the class does not -- round-off apart
    if method == capi.METHOD_RGD or not fused_rtr:
        assert np.array_equal(ts.global_X(), th.global_X())
    else:
        assert np.abs(ts.global_X() - th.global_X()).max() < 1e-9
    ts.close()
    th.close()


@pytest.mark.parametrize("dataset", ["tinyGrid3D", "smallGrid3D", "sphere2500", "torus3D", "parking-garage"])
def test_chordal_initialisation(dataset):
    """8f-1: chordal relaxation on the GPU vs the oracle's sparse-Cholesky version.  Round 5: through the solver's own
    machinery (csrc/chordal.hip: the translation-free connection Laplacian in its dense or two-level form, pose 0 pinned as
    a shared edge, one refinement step)."""
    m, _, n = load(dataset, 1)
    Th = capi.chordal_init(m.view(capi.MEAS_DTYPE), n)
    To = O.chordal_init(m, n)
    assert np.abs(Th - To).max() < 1e-8 * max(1.0, np.abs(To).max())
    R = Th.reshape(n, 4, 3)[:, :3, :]
    assert np.abs(np.einsum("nia,nja->nij", R, R) - np.eye(3)).max() < 1e-12
    assert np.all(np.linalg.det(R) > 0.999)
    if dataset == "sphere2500":
        c = 2 * O.measurement_cost(m, O.lift(Th, n, O.fixed_stiefel(5), 5), 5)
        assert abs(c - 1971.175) < 0.01  # SE-Sync's chordal-initialisation cost for sphere2500


def test_chordal_initialisation_paths_agree_and_honour_weights():
    """the team path and the dense fallback (DPGO_CHORDAL_DENSE=1, a fresh process: the switch is read once) give the same
    poses, also with GNC-style edge weights (weight x kappa / tau is what both stages use) and a backward edge into pose 0"""
    import os
    import subprocess
    import sys
    import tempfile
    from tests.util import ROOT
    m, _, n = load("smallGrid3D", 1)
    rng = np.random.default_rng(3)
    mw = m.copy()
    mw["weight"] = np.where(mw["p1"] + 1 == mw["p2"], 1.0, rng.uniform(0.05, 1.0, len(mw)))
    To = O.chordal_init(mw, n)
    Th = capi.chordal_init(mw.view(capi.MEAS_DTYPE), n)
    assert np.abs(Th - To).max() < 1e-9 * max(1.0, np.abs(To).max())
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "m.npy"), mw)
        code = ("import numpy as np, sys; sys.path.insert(0, %r); from dpgo_ros_amd import capi; m = np.load(%r); "
                "np.save(%r, capi.chordal_init(m.view(capi.MEAS_DTYPE), %d))" % (ROOT, os.path.join(d, "m.npy"), os.path.join(d, "T.npy"), n))
        subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, DPGO_CHORDAL_DENSE="1"), timeout=300)
        Td = np.load(os.path.join(d, "T.npy"))
    assert np.abs(Td - Th).max() < 1e-9 * max(1.0, np.abs(To).max())


def _oracle_robust_local_init(mo, n, kw):
    loc = mo.copy()
    odo = loc["p1"] + 1 == loc["p2"]
    loc["fixed_weight"][odo] = 1
    ag = O.Agent(0, O.default_params(r=3, num_robots=1, acceleration=0, **kw))
    ag.add_measurements(loc)
    ag.set_X(O.odometry_init(loc, n))
    for u in range(kw["robust_opt_num_weight_updates"] + 1):
        for _ in range(kw["robust_opt_inner_iters"]):
            ag.iterate(True)
        if u < kw["robust_opt_num_weight_updates"]:
            ag.update_measurement_weights()
    mw = ag.measurements()  # [odometry..., private...]
    w = np.zeros(len(mo))
    w[odo], w[~odo] = mw["weight"][:odo.sum()], mw["weight"][odo.sum():]
    return ag.get_X(), w, ag


def test_robust_local_initialization():
    """f-1: InitializationMethod::GNC_TLS -- single-robot GNC-TLS solve on the device (r = d = 3, odometry fixed)
    against the same sequence driven through the oracle's agent."""
    from tests.util import add_outliers
    m, _, n = load("smallGrid3D", 1)
    mo = add_outliers(m, n, frac=0.1, seed=3)
    nout = len(mo) - len(m)
    kw = dict(method=capi.METHOD_RTR, gradnorm_tol=1e-3, robust_cost_type=capi.COST_GNC_TLS, gnc_barc=3.0,
              gnc_mu_step=2.0, gnc_init_mu=1e-3, robust_opt_num_weight_updates=2, robust_opt_inner_iters=3,
              rtr_max_radius=500.0)
    # (A) iterate parity over two re-weightings (9 RTR iterations, tCG counts equal): 1e-7.  Longer schedules hit
    # borderline tCG terminations (one extra inner iteration out of ~60 in one run) after which the two
    # trajectories differ at the 1e-3 level while solving the same weighting -- compared in (B).
    T, w = capi.robust_local_init(mo, n, capi.default_params(r=5, num_robots=4, acceleration=1, **kw))
    To, wo, _ = _oracle_robust_local_init(mo, n, kw)
    assert np.abs(w - wo).max() < 1e-7 and ((wo > 0) & (wo < 1)).any()
    assert np.abs(T - To).max() < 1e-7
    # (B) the full schedule: planted outliers end at weight 0, everything else at 1; trajectories agree to the
    # solve tolerance, costs under the final weights tightly.
    kw.update(robust_opt_num_weight_updates=12, robust_opt_inner_iters=8, gnc_barc=5.0)  # chi(6 dof): P(res > 5) ~ 2e-4
    T, w = capi.robust_local_init(mo, n, capi.default_params(r=3, num_robots=1, **kw))
    To, wo, ag = _oracle_robust_local_init(mo, n, kw)
    assert np.abs(w - wo).max() < 1e-6
    genuine = w[:len(m)]
    assert (w[-nout:] < 1e-3).all() and (genuine > 0.99).mean() > 0.97 and (genuine[m["p1"] + 1 == m["p2"]] == 1).all()
    assert np.abs(T - To).max() < 5e-3
    fh, fo = ag.eval(T)[0], ag.eval(To)[0]
    assert abs(fh - fo) <= 1e-6 * abs(fo)
    R = T.reshape(n, 4, 3)[:, :3, :]
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-12
    assert (np.linalg.det(R) > 0).all()


def _oracle_simultaneous(to, N, ticks):
    """all agents step from the neighbour poses of the beginning of the tick (oracle, per-agent API)"""
    for _ in range(ticks):
        snap = {}
        for b in range(N):
            for c in to.agents[b].neighbors():
                snap[(b, c)] = to.agents[b].get_public_poses(c, False)
        for (b, c), (ids, P) in snap.items():
            to.agents[c].update_neighbor_poses(b, ids, P, False)
        for b in range(N):
            assert to.agents[b].iterate(True)


@pytest.mark.parametrize("dataset,N", [("smallGrid3D", 3), ("tunnels", 8)])
def test_simultaneous_updates(dataset, N):
    """ASAPP with all clocks in lockstep (dpgo_team_run_simultaneous): every agent takes an RGD step per tick in
    the same launches; equals the oracle's agents stepping one by one from poses frozen at the tick's start."""
    kw = dict(method=capi.METHOD_RGD, rgd_stepsize=0.05, acceleration=0)
    if dataset == "tunnels":
        from tests.util import load_tunnels
        m = load_tunnels(capi.WEIGHT_WRAPPER)
        nk = [0] * N
        for e in m:
            nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1)
            nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
        Ts = []
        for k in range(N):
            odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
            odo["r1"] = 0; odo["r2"] = 0
            Ts.append(O.odometry_init(odo, nk[k]))
        th, to = _pair_from(m, sum(nk), N, np.concatenate(Ts), **kw)
    else:
        th, to, n = make_pair(dataset, N, **kw)
    f0 = to.cost()
    done = 0
    for ticks in (1, 5, 70):   # 70 > one graph of 64 ticks
        th.run_simultaneous(ticks)
        _oracle_simultaneous(to, N, ticks)
        done += ticks
        assert np.abs(th.global_X() - to.global_X()).max() < 1e-8, done
        for a in range(N):
            sh, so = th.agents[a].status(), to.agents[a].status()
            assert sh.iteration_number == so.iteration_number == done
            assert abs(sh.relative_change - so.relative_change) < 1e-8
            rh, ro = th.agents[a].opt_result(), to.agents[a].opt_result()
            assert abs(rh.f_opt - ro.f_opt) <= 1e-9 * max(1.0, abs(ro.f_opt))
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost()) and to.cost() < f0
    # the sequential schedule still works afterwards (team counter advanced by ticks * N)
    th.run(4)
    th.close()


@pytest.mark.parametrize("name,kw,init,coarse,expected", [
    ("rgd_nesterov", dict(method=capi.METHOD_RGD, rgd_stepsize=0.2, acceleration=1, restart_interval=20), "odom", 100, 12354),
    ("rtr_nesterov", dict(method=capi.METHOD_RTR, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=1e-2,
                          restart_interval=50), "odom", 1, 740),
    ("rtr_nesterov_chordal", dict(method=capi.METHOD_RTR, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50,
                                  gradnorm_tol=1e-2, restart_interval=50), "chordal", 1, 227),
])
def test_headline_iterations_to_gap(name, kw, init, coarse, expected):
    """BASELINE metric, second half: iterations until (f - f*) / f* <= 1e-6 on sphere2500 / 5 agents with bench.py's
    configurations.  The RTR cases run the oracle LIVE with the same protocol (8 s / 2 s of CPU) and the HIP path must
    cross at the oracle's iteration give or take round-off drift; `expected` (740 / 227) is asserted of the oracle as
    well.  Accelerated RGD (step 0.2, restart 20) needs 30 s of CPU for its 12354 iterations: the oracle's count
    measured offline with the same protocol (gap checked every `coarse` iterations until < 3e-6, then every one)."""
    FSTAR = 843.5029071410438
    m, mp, n = load("sphere2500", 5)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=5, num_robots=5, **kw))
    T = O.odometry_init(m, n) if init == "odom" else capi.chordal_init(m.view(capi.MEAS_DTYPE), n)
    th.set_initial(T, O.fixed_stiefel(5))

    def crossing(step, cost):
        k, gap = 0, float("inf")
        while k < 20000:
            ch = coarse if gap > 3e-6 else 1
            step(ch)
            k += ch
            gap = (cost() - FSTAR) / FSTAR
            if gap <= 1e-6:
                break
        return k, gap

    k, gap = crossing(th.run, th.cost)
    th.close()
    if kw["method"] == capi.METHOD_RTR:
        to = O.Team(mp, n, O.default_params(r=5, num_robots=5, **kw))
        to.set_initial(O.odometry_init(m, n) if init == "odom" else O.chordal_init(m, n), O.fixed_stiefel(5))
        ko, _ = crossing(lambda c: [to.iterate() for _ in range(c)], to.cost)
        assert abs(ko - expected) <= 2, (name, ko)
        assert abs(k - ko) <= 2, (name, k, ko, gap)
    assert abs(k - expected) <= 2, (name, k, gap)


def test_time_kernel_variants_run():
    """the measurement entry point bench.py uses for its roofline object: bare preconditioner apply (0), evaluation (1),
    fused step kernel back to back (9) and inside the running pipelined iteration (10)"""
    kw = dict(method=capi.METHOD_RGD, rgd_stepsize=0.1, acceleration=1, restart_interval=20)
    th, to, n = make_pair("smallGrid3D", 3, **kw)
    th.run(5)
    n0 = th.agents[1].n
    for which in (0, 1, 9, 10):
        ms, nbytes = th.time_kernel(1, which, reps=20)
        assert 0 < ms < 1.0
        if which == 0:
            assert nbytes == 8.0 * (4 * n0) ** 2 + 3 * 8.0 * 5 * 4 * n0
        if which in (9, 10):
            others = sum(8.0 * 5 * 4 * th.agents[a].n for a in (0, 2))
            assert nbytes == 8.0 * (4 * n0) ** 2 + 7 * 8.0 * 5 * 4 * n0 + 4 * others
    th.close()


def test_bench_configuration_against_the_live_oracle():
    """The EXACT workload bench.py times (sphere2500 / 5 agents, preconditioned RGD step 0.2 + Nesterov, restart 20,
    odometry guess) against the oracle run live, 320 iterations -- sixteen restart iterations among them -- compared
    every 20: X, the auxiliary sequence Y and the momentum sequence V of every agent, the cost, and at the end every
    agent's status and the local result of the last block update.  Tolerance: the two runs differ in summation order only (1e-13 per operation);
    over 320 accelerated iterations that grows to what the asserts state."""
    import bench
    th, to, n = make_pair("sphere2500", 5, **bench.RGD)
    worst = 0.0
    for chunk in range(16):
        th.run(20)
        for _ in range(20):
            to.iterate()
        for a in range(5):
            ah, ao = th.agents[a], to.agents[a]
            for gh, go in ((ah.get_X, ao.get_X), (ah.get_Y, ao.get_Y), (ah.get_V, ao.get_V)):
                worst = max(worst, np.abs(gh() - go()).max())
        assert worst < 1e-10, (chunk, worst)   # measured: 2.7e-13 after 320 iterations
        assert abs(th.cost() - to.cost()) <= 1e-10 * abs(to.cost()), chunk
    for a in range(5):
        sh, so = th.agents[a].status(), to.agents[a].status()
        assert sh.iteration_number == so.iteration_number == 320
        assert abs(sh.relative_change - so.relative_change) < 1e-9 and sh.ready_to_terminate == so.ready_to_terminate
    # (a run keeps the local result of its LAST block update only: iteration 319 belongs to agent 4)
    rh, ro = th.agents[4].opt_result(), to.agents[4].opt_result()
    assert abs(rh.f_opt - ro.f_opt) <= 1e-10 * abs(ro.f_opt) and abs(rh.gradnorm_opt - ro.gradnorm_opt) <= 1e-8 * max(1.0, ro.gradnorm_opt)
    print("bench configuration vs live oracle, 320 iterations: max |X, Y, V difference| = %.2e" % worst)
    th.close()


@pytest.mark.parametrize("dataset,r", [("tunnels", 5), ("sphere2500", 5), ("tunnels", 3), ("sphere2500", 8)])
def test_staged_shared_edge_evaluation_is_bitwise_the_plain_one(dataset, r, monkeypatch):
    """k_eval_staged (helper waves put the operands of a tile's shared edges into LDS, csrc/spmm.hip) forms G in
    g_row_range's order: lockstep ticks, RTR block updates and the cost are BITWISE those of the plain k_eval -- on tunnels
    (up to 20 shared edges per pose: staged by default) and, forced, on sphere2500 / 5; ranks 3, 5, 8"""
    from tests.util import load_tunnels

    def team(staged, **kw):
        monkeypatch.setenv("DPGO_STAGED_EVAL", "0" if staged else "1000000")
        if dataset == "tunnels":
            m = load_tunnels(1)
            nk = [int(max(m["p1"][m["r1"] == k].max(), m["p2"][m["r2"] == k].max())) + 1 for k in range(8)]
            Ts = []
            for k in range(8):
                odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
                odo["r1"] = 0
                odo["r2"] = 0
                Ts.append(O.odometry_init(odo, nk[k]))
            T, N, mp = np.concatenate(Ts), 8, m
        else:
            m, mp, n = load(dataset, 5)
            T, N = O.odometry_init(m, n), 5
        t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), capi.default_params(r=r, num_robots=N, **kw))
        t.set_initial(T, O.fixed_stiefel(r))
        return t

    for kw, run in ((dict(method=1, rgd_stepsize=0.2, acceleration=0), lambda t: t.run_simultaneous(40)),
                    (dict(method=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, acceleration=0), lambda t: t.run(16))):
        ta, tb = team(False, **kw), team(True, **kw)
        for t in (ta, tb):
            run(t)
            t.synchronize()
        for k in ta.ids:
            assert np.array_equal(ta.agents[k].get_X(), tb.agents[k].get_X()), (dataset, kw["method"], k)
        assert ta.cost() == tb.cost()
        ta.close()
        tb.close()
