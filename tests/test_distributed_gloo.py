"""N > 1 path on CPU: world_size-2 `gloo` run of dpgo_ros_amd.distributed (ownership, neighbour
topology, pull-before-use exchange order, schedule) with the CPU oracle plugged in as the compute
backend.  The transport must not change a single bit relative to the single-process oracle team."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

from tests.util import DATA, ROOT, load


class OracleBackend:
    """test-only compute backend for DistributedRBCD: oracle agents + CPU tensors."""

    def __init__(self, meas, params, local_ids, T, Y, offsets):
        import torch
        from oracle import oracle as O
        self.torch, self.O, self.r = torch, O, params.r
        self.agents = {}
        for a in local_ids:
            ag = O.Agent(a, params)
            ag.add_measurements(meas)
            ag.set_X(O.lift(T[12 * offsets[a]:12 * (offsets[a] + ag.n)], ag.n, Y, params.r))
            self.agents[a] = ag
        self._buf = {}

    def iterate(self, agent, do_opt):
        return self.agents[agent].iterate(do_opt)

    def pack(self, agent, nbr, seqs, count):
        parts = []
        for aux in seqs:
            ids, P = self.agents[agent].get_public_poses(nbr, bool(aux))
            assert len(ids) == count
            parts.append(P)
        return self.torch.from_numpy(np.concatenate(parts))

    def recv_buffer(self, agent, nbr, seqs, count):
        return self._buf.setdefault((agent, nbr, len(seqs)), self.torch.empty(count * len(seqs) * 4 * self.r, dtype=self.torch.float64))

    def unpack(self, agent, nbr, seqs, tensor):
        ag = self.agents[agent]
        chunks = tensor.numpy().reshape(len(seqs), -1)
        for q, aux in enumerate(seqs):
            ag.update_neighbor_poses(nbr, ag.neighbor_pose_ids(nbr), chunks[q], bool(aux))

    def pull_local(self, agent):
        ag = self.agents[agent]
        for b in ag.neighbors():
            if b in self.agents:
                for aux in (False, True):
                    ids, P = self.agents[b].get_public_poses(agent, aux)
                    ag.update_neighbor_poses(b, ids, P, aux)

    def set_groups(self, groups):
        self.groups = groups

    def run_group(self, g, members):
        # sequential equivalent: the members' block updates in id order; everyone else only counts
        for m in members:
            for a, ag in self.agents.items():
                ag.iterate(a == m)

    def tick_local(self):
        # agents step one by one from their stored (frozen) copies of the neighbour poses = simultaneous
        for a in sorted(self.agents):
            self.pull_local(a)
        for a in sorted(self.agents):
            self.agents[a].iterate(True)

    def local_update_weights(self):
        changed = 0
        for a in sorted(self.agents):
            self.agents[a].update_measurement_weights()
        for a in sorted(self.agents):
            for e in self.agents[a].measurements():
                if e["r1"] == e["r2"]:
                    continue
                other = int(e["r2"] if e["r1"] == a else e["r1"])
                if other > a and other in self.agents:
                    self.agents[other].set_measurement_weight(int(e["r1"]), int(e["p1"]), int(e["r2"]), int(e["p2"]),
                                                              float(e["weight"]), bool(e["fixed_weight"]))
                    self.agents[other].clear_data_matrices()
                    changed += 1
        return changed

    def _pair_edges(self, agent, nbr):
        ms = self.agents[agent].measurements()
        return ms[(ms["r1"] != ms["r2"]) & ((ms["r1"] == nbr) | (ms["r2"] == nbr))]

    def owned_weights(self, agent, nbr):
        sel = self._pair_edges(agent, nbr)
        out = np.empty(2 * len(sel))
        out[0::2], out[1::2] = sel["weight"], sel["fixed_weight"]
        return out, sel

    def apply_weights(self, agent, nbr, payload):
        sel = self._pair_edges(agent, nbr)
        for e, w, f in zip(sel, payload[0::2], payload[1::2]):
            assert self.agents[agent].set_measurement_weight(int(e["r1"]), int(e["p1"]), int(e["r2"]), int(e["p2"]), float(w), bool(f))
        self.agents[agent].clear_data_matrices()
        return len(sel)

    def to_transport(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr))

    def from_transport(self, t):
        return t.numpy()

    def transport_empty(self, n):
        return self.torch.empty(n, dtype=self.torch.float64)

    def partial_cost(self):
        f = 0.0
        for a, ag in self.agents.items():
            for m in ag.measurements():
                if m["r1"] != m["r2"] and min(m["r1"], m["r2"]) != a:
                    continue
                ok, res = ag.compute_residual(m)
                assert ok
                f += 0.5 * m["weight"] * res * res
        return f


def _worker(rank, world, port, cfg, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dpgo_ros_amd.distributed import DistributedRBCD, owner_of
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds, N, kw, iters = cfg
    m, mp, n = load(ds, N)
    params = O.default_params(r=5, num_robots=N, **{k: v for k, v in kw.items() if k != "colored"})
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(5)
    per = n // N
    offsets = {a: a * per for a in range(N)}
    mine = [a for a in range(N) if owner_of(a, world) == rank]
    be = OracleBackend(mp, params, mine, T, Y, offsets)
    drv = DistributedRBCD(dist, be, mp, N, kw.get("acceleration", 0), rank, world)
    drv.exchange_all()
    costs = []
    if kw.get("colored"):
        for k in range(iters // N):
            drv.sweep_colored()
            costs.append(drv.global_cost(torch, "cpu"))
    else:
        for k in range(iters):
            drv.step()
            if k % 3 == 2:
                costs.append(drv.global_cost(torch, "cpu"))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), costs=np.array(costs),
             **{"X%d" % a: be.agents[a].get_X() for a in mine})
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    ("smallGrid3D", 3, dict(method=0, gradnorm_tol=1e-2), 9),
    ("smallGrid3D", 3, dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=4), 12),
    ("smallGrid3D", 2, dict(method=0, acceleration=1, restart_interval=5, gradnorm_tol=1e-2), 8),
    ("smallGrid3D", 4, dict(method=0, gradnorm_tol=1e-2, colored=1), 12),
    # BASELINE configs[2] as bench.py's multi-rank leg runs it (config2_ranks_leg): sphere2500 over 8 agents, RTR 3 / 50 / 0.5,
    # round robin, agent a on rank a % 2
    ("sphere2500", 8, dict(method=0, gradnorm_tol=0.5, rtr_iterations=3, rtr_tcg_iterations=50), 16),
]


@pytest.mark.parametrize("cfg", CASES)
def test_two_rank_gloo_run_is_bitwise_the_single_process_schedule(cfg):
    import torch.multiprocessing as mp_
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_worker, args=(2, port, cfg, d), nprocs=2, join=True)
        outs = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(2)]
        ds, N, kw, iters = cfg
        m, mp, n = load(ds, N)
        ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **{k: v for k, v in kw.items() if k != "colored"}))
        ref.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))
        costs = []
        if kw.get("colored"):
            from dpgo_ros_amd.distributed import greedy_coloring, topology
            groups = greedy_coloring(topology(mp, N)[0], N)
            ref.set_schedule([a for g in groups for a in g])
        for k in range(iters):
            ref.iterate()
            if (k % N == N - 1) if kw.get("colored") else (k % 3 == 2):
                costs.append(ref.cost())
        for a in range(N):
            Xa = outs[a % 2]["X%d" % a]
            assert np.array_equal(Xa, ref.agents[a].get_X()), "agent %d differs" % a
        for r in range(2):  # both ranks see the same all-reduced cost
            assert np.allclose(outs[r]["costs"], costs, rtol=1e-13, atol=0)


def test_topology_matches_agent_bookkeeping():
    from dpgo_ros_amd.distributed import topology
    from oracle import oracle as O
    m, mp, n = load("sphere2500", 5)
    nbrs, npub = topology(mp, 5)
    t = O.Team(mp, n, O.default_params(r=5, num_robots=5))
    for a in range(5):
        assert nbrs[a] == t.agents[a].neighbors()
        for b in nbrs[a]:
            assert npub[(a, b)] == len(t.agents[a].public_pose_ids(b)) == len(t.agents[b].neighbor_pose_ids(a))
    assert nbrs == {0: [1], 1: [0, 2], 2: [1, 3], 3: [2, 4], 4: [3]}  # chain


def _worker_modes(rank, world, port, mode, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dpgo_ros_amd.distributed import DistributedRBCD, owner_of
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, mp, n, T, kw = _mode_problem(mode)
    params = O.default_params(r=5, num_robots=N, **kw)
    per = n // N
    offsets = {a: a * per for a in range(N)}
    mine = [a for a in range(N) if owner_of(a, world) == rank]
    be = OracleBackend(mp, params, mine, T, O.fixed_stiefel(5), offsets)
    drv = DistributedRBCD(dist, be, mp, N, kw.get("acceleration", 0), rank, world)
    drv.exchange_all()
    changed = []
    if mode == "ticks":
        for _ in range(6):
            drv.tick_simultaneous()
    else:
        for rnd in range(2):
            for _ in range(2 * N):
                drv.step()
            changed.append(drv.update_weights())
        for _ in range(N):
            drv.step()
    cost = drv.global_cost(torch, "cpu")
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), cost=cost, changed=np.array(changed),
             **{"X%d" % a: be.agents[a].get_X() for a in mine},
             **{"W%d" % a: be.agents[a].measurements()["weight"] for a in mine})
    dist.barrier()
    dist.destroy_process_group()


def _mode_problem(mode):
    from oracle import oracle as O
    from tests.util import add_outliers
    N = 3
    m, _, n = load("smallGrid3D", 1)
    if mode == "ticks":
        kw = dict(method=1, rgd_stepsize=0.05, acceleration=0)
        mo = m
    else:
        kw = dict(method=0, gradnorm_tol=1e-2, acceleration=1, restart_interval=5, robust_cost_type=O.COST_GNC_TLS, gnc_barc=3.0,
                  gnc_mu_step=2.0, gnc_init_mu=1e-2, robust_opt_num_weight_updates=3, robust_opt_inner_iters=2 * N)
        mo = add_outliers(m, n, frac=0.1, seed=0)
    return N, O.partition(mo, n, N), n, O.odometry_init(mo, n), kw


@pytest.mark.parametrize("mode", ["ticks", "gnc"])
def test_two_rank_lockstep_ticks_and_weight_rounds(mode):
    """BASELINE configs[4] (lockstep ASAPP ticks) and configs[3] (UPDATE_WEIGHT rounds with the weights of shared edges
    crossing ranks, src/PGOAgentROS.cpp:721-754,1315-1353) over 2 `gloo` ranks: bitwise the single-process runs."""
    import torch.multiprocessing as mp_
    from oracle import oracle as O
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_worker_modes, args=(2, port, mode, d), nprocs=2, join=True)
        outs = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(2)]
    N, mp, n, T, kw = _mode_problem(mode)
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    ref.set_initial(T, O.fixed_stiefel(5))
    if mode == "ticks":
        for _ in range(6):
            ref.exchange_all()
            for a in ref.agents:
                a.iterate(True)
        ref.exchange_all()
    else:
        for rnd in range(2):
            for _ in range(2 * N):
                ref.iterate()
            ref.update_weights()
        for _ in range(N):
            ref.iterate()
    for a in range(N):
        assert np.array_equal(outs[a % 2]["X%d" % a], ref.agents[a].get_X()), "agent %d differs" % a
        assert np.array_equal(outs[a % 2]["W%d" % a], ref.agents[a].measurements()["weight"])
    assert abs(float(outs[0]["cost"]) - ref.cost()) <= 1e-12 * abs(ref.cost())
    if mode == "gnc":
        w = np.concatenate([ref.agents[a].measurements()["weight"] for a in range(N)])
        assert (w < 1).sum() > 0


def _worker_delay(rank, world, port, cfg, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from dpgo_ros_amd.distributed import DistributedRBCD, owner_of
    from oracle import oracle as O
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, kw, iters, delay = cfg
    m, mp, n = load("smallGrid3D", N)
    params = O.default_params(r=5, num_robots=N, **kw)
    per = n // N
    mine = [a for a in range(N) if owner_of(a, world) == rank]
    be = OracleBackend(mp, params, mine, O.odometry_init(m, n), O.fixed_stiefel(5), {a: a * per for a in range(N)})
    drv = DistributedRBCD(dist, be, mp, N, kw.get("acceleration", 0), rank, world, max_delayed_iterations=delay)
    drv.exchange_all()
    # peer access is a device feature: on a backend without it every rank must learn so together and carry on
    # (bench.py relies on this agreement to skip its free-running leg instead of hanging in a collective)
    assert drv.enable_peer_access() is False and "peer access" in drv.peer_error
    m0 = drv.messages
    for _ in range(iters):
        drv.step()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), messages=drv.messages - m0,
             **{"X%d" % a: be.agents[a].get_X() for a in mine})
    dist.barrier()
    dist.destroy_process_group()


def _staleness_reference(N, kw, iters, delay, world):
    """single-process emulation of the staleness gate (src/PGOAgentROS.cpp:136-149) with oracle agents: a remote
    neighbour's copy is refreshed only when it is more than `delay` iterations behind the neighbour's latest change;
    co-resident neighbours are always read fresh."""
    from dpgo_ros_amd.distributed import owner_of
    from oracle import oracle as O
    m, mp, n = load("smallGrid3D", N)
    t = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    t.set_initial(O.odometry_init(m, n), O.fixed_stiefel(5))   # everyone holds everyone's initial poses
    accel = bool(kw.get("acceleration", 0))
    version, sent = [0] * N, {}
    for a in range(N):
        for b in t.agents[a].neighbors():
            sent[(b, a)] = 0

    def deliver(b, a):
        for aux in ((False, True) if accel else (False,)):
            ids, P = t.agents[b].get_public_poses(a, aux)
            t.agents[a].update_neighbor_poses(b, ids, P, aux)

    for k in range(iters):
        sel = k % N
        for b in range(N):
            if b != sel:
                t.agents[b].iterate(False)
                if accel:
                    version[b] = k + 1
        for b in t.agents[sel].neighbors():
            if owner_of(b, world) == owner_of(sel, world):
                deliver(b, sel)
            else:
                behind = version[b] - sent[(b, sel)]
                if behind != 0 and behind > delay:
                    deliver(b, sel)
                    sent[(b, sel)] = version[b]
        t.agents[sel].iterate(True)
        version[sel] = k + 1
    return t


@pytest.mark.parametrize("cfg", [
    (3, dict(method=0, gradnorm_tol=1e-2), 12, 0),
    (3, dict(method=0, gradnorm_tol=1e-2), 12, 3),                                       # the struct default
    (4, dict(method=1, acceleration=1, rgd_stepsize=0.1, restart_interval=50), 16, 2),
])
def test_staleness_gate_with_delayed_iterations(cfg):
    """maxDelayedIterations (include/dpgo_ros/PGOAgentROS.h:83, src/PGOAgentROS.cpp:136-149): the 2-rank run equals the
    single-process emulation of the gate bit for bit, and a larger allowance sends fewer messages"""
    import torch.multiprocessing as mp_
    N, kw, iters, delay = cfg
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_worker_delay, args=(2, port, cfg, d), nprocs=2, join=True)
        outs = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(2)]
    ref = _staleness_reference(N, kw, iters, delay, 2)
    for a in range(N):
        assert np.array_equal(outs[a % 2]["X%d" % a], ref.agents[a].get_X()), "agent %d differs" % a
    msgs = int(outs[0]["messages"]) + int(outs[1]["messages"])
    if delay == 0:
        # plain RBCD: a neighbour that has not moved since it last sent sends nothing (2 ops per delivered message)
        assert 0 < msgs <= 2 * 2 * iters
    else:
        assert msgs < 2 * 2 * iters


def test_staleness_gate_is_kept_per_sequence():
    """an X-only exchange (colour sweep / lockstep tick / exchange_to(sel, (0,))) must not make the gate skip the
    auxiliary sequence of a later accelerated step, whatever max_delayed_iterations allows for X (host logic only: a
    recording stand-in for torch.distributed, rank 0 of 2)"""
    from dpgo_ros_amd.distributed import DistributedRBCD

    class FakeDist:
        isend, irecv = "isend", "irecv"

        @staticmethod
        def P2POp(op, t, peer):
            return (op, t, peer)

        @staticmethod
        def batch_isend_irecv(ops):
            return []

    class Recorder:
        def __init__(self):
            self.packed, self.unpacked = [], []

        def iterate(self, agent, do_opt):
            return True

        def pack(self, agent, nbr, seqs, count):
            self.packed.append((agent, nbr, tuple(seqs)))
            return None

        def recv_buffer(self, agent, nbr, seqs, count):
            return None

        def unpack(self, agent, nbr, seqs, tensor):
            self.unpacked.append((agent, nbr, tuple(seqs)))

        def pull_local(self, agent):
            pass

    m, mp, n = load("smallGrid3D", 2)
    for delay in (0, 2):
        be = Recorder()
        drv = DistributedRBCD(FakeDist, be, mp, 2, 1, 0, 2, max_delayed_iterations=delay)
        drv.exchange_to(0, (0,))                    # robot 1 (rank 1) -> robot 0 (rank 0): X only
        assert be.unpacked == [(0, 1, (0,))]
        assert drv.step() == 0                      # accelerated: robot 0 needs X AND Y of robot 1
        assert be.unpacked[-1] == (0, 1, (0, 1)), (delay, be.unpacked)
        n_before = len(be.unpacked)
        drv.exchange_to(0, (0, 1))                  # nothing moved since: both sequences are current
        assert len(be.unpacked) == n_before
        assert drv.step() == 1                      # robot 1 optimizes on rank 1: rank 0 packs robot 0's poses for it
        assert be.packed[-1] == (0, 1, (0, 1))
