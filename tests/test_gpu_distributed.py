"""The N > 1 driver with the HIP backend on the GPU box (one GPU): (1) bench.py's multi-rank code path under the `nccl`
(= RCCL) backend at world size 1, launched the way the driver launches it; (2) two processes sharing cuda:0 (gloo
transport, slabs bounced through the host: RCCL refuses two ranks on one GPU) for the lockstep ASAPP ticks of BASELINE
configs[4] and the UPDATE_WEIGHT rounds of configs[3] with shared-edge weights crossing ranks; (3) the same two processes
with each other's pose arrays imported over HIP IPC (dpgo_agent_export_state / dpgo_team_import_peer): the synchronous
schedule with neighbours read in place, bit-for-bit the message-passing iterates, and the free-running asynchronous mode."""
import json
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import ROOT, add_outliers, load

pytestmark = pytest.mark.gpu


def test_bench_multi_rank_path_runs_under_rccl_at_world_size_one():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DPGO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "60", "--warmup", "10"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and 0 < d["value"] < 1.0 and d["exchange"].startswith("RCCL")
    assert np.isfinite(d["relcost_after_run"]) and d["colour_parallel_plain_rtr"]["classes"] == 2
    assert d["asapp_ticks_tunnels"]["ms_per_tick"] > 0
    ex = d["exchange_timing"]
    assert ex["rccl_world_size"] == 1 and ex["backend"] == "nccl" and len(ex["ranks"]) == 1 and ex["ranks"][0]["device"]
    # the value is the library-side RCCL path; at world size 1 it must sit with the single-GPU headline (hipGraphs), and
    # the loopback leg shows one grouped self-send per iteration really crossing RCCL
    assert ex["value_is"].startswith("rccl_in_library") and d["value"] == ex["ms_per_step_rccl_in_library"] < 0.05
    assert ex["loopback"]["messages_per_step"] == 1.0 and ex["loopback"]["bytes_per_step"] > 0
    assert ex["global_cost_library_vs_torch_rel_diff"] < 1e-12 and ex["rccl_version_code"] > 0
    c2 = ex["config2_sphere2500_8_agents_rtr"]  # BASELINE configs[2] through the multi-rank driver
    assert 0 < c2["ms_per_iter"] < 5.0 and np.isfinite(c2["relcost_after_run"]) and c2["rccl_point_to_point_ops_per_iter_this_rank"] == 0


def test_bench_multi_rank_path_starts_without_a_launcher():
    """no torchrun, no RANK / MASTER_ADDR in the environment: `python bench.py` with the N > 1 driver forced at world size 1"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DPGO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "10"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["exchange_timing"]["rccl_world_size"] == 1 and 0 < d["value"] < 1.0


def test_bench_multi_rank_line_survives_a_stuck_extra_leg():
    """the legs behind the main measurement have never met a peer on another device: if they do not finish within
    DPGO_BENCH_EXTRAS_TIMEOUT, rank 0 still prints ONE JSON line with the library-side RCCL value and exits 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DPGO_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", DPGO_BENCH_EXTRAS_TIMEOUT="0.2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "10"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["extras"].startswith("timed out") and 0 < d["value"] < 1.0 and d["metric"].startswith("ms/RBCD-iteration")
    assert d["exchange_timing"]["rccl_world_size"] == 1 and np.isfinite(d["relcost_after_run"])


def _problem(mode):
    N = 3
    m, _, n = load("smallGrid3D", 1)
    if mode in ("ticks", "peer_free", "peer_agent_api"):
        kw = dict(method=1, rgd_stepsize=0.05, acceleration=0)
        mo = m
    elif mode in ("peer_sync", "peer_token"):
        kw = dict(method=0, gradnorm_tol=1e-2, acceleration=1, restart_interval=5)
        mo = m
    elif mode == "peer_token_rgd":
        kw = dict(method=1, rgd_stepsize=0.05, acceleration=1, restart_interval=7)
        mo = m
    elif mode == "peer_token_plain":
        kw = dict(method=0, gradnorm_tol=1e-2, acceleration=0)
        mo = m
    else:
        kw = dict(method=0, gradnorm_tol=1e-2, acceleration=1, restart_interval=5, robust_cost_type=O.COST_GNC_TLS, gnc_barc=3.0,
                  gnc_mu_step=2.0, gnc_init_mu=1e-2, robust_opt_num_weight_updates=3, robust_opt_inner_iters=2 * N)
        mo = add_outliers(m, n, frac=0.1, seed=0)
    return N, O.partition(mo, n, N), n, O.odometry_init(mo, n), kw


def _worker(rank, world, port, mode, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dpgo_ros_amd.distributed import DistributedRBCD, HipBackend, owner_of
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, mp, n, T, kw = _problem(mode)
    mine = [a for a in range(N) if owner_of(a, world) == rank]
    be = HipBackend(mp.view(capi.MEAS_DTYPE), capi.default_params(r=5, num_robots=N, **kw), mine, 0, torch, host_staging=True)
    per = n // N
    be.team.set_initial(T, O.fixed_stiefel(5), offsets=np.array([a * per for a in mine], dtype=np.int32))
    drv = DistributedRBCD(dist, be, mp, N, kw.get("acceleration", 0), rank, world)
    drv.exchange_all()
    extra = {}
    if mode == "ticks":
        for _ in range(6):
            drv.tick_simultaneous()
    elif mode == "peer_sync":
        drv.enable_peer_access()
        for _ in range(4 * N):
            drv.step_peer()
        be.sync()
        dist.barrier()
        extra["messages"] = drv.messages
    elif mode.startswith("peer_token"):
        assert drv.enable_peer_access(), drv.peer_error
        # the whole run is enqueued in three calls; no barrier, no stream synchronisation, no message in between
        for chunk in (5, 1, 4 * N - 6):
            drv.run_peer(chunk)
        be.sync()
        dist.barrier()
        extra["messages"] = drv.messages
    elif mode == "peer_agent_api":
        drv.enable_peer_access()
        # per-agent iterate on a team with imported peers: no update_neighbor_poses anywhere, every neighbour is read
        # in place; the ranks rendezvous around each iterate(true)
        for k in range(3 * N):
            sel = k % N
            be.sync()
            dist.barrier()
            if sel in mine:
                be.team.agents[sel].iterate(True)
            be.sync()
            dist.barrier()
            for a in mine:
                if a != sel:
                    be.team.agents[a].iterate(False)
    elif mode == "peer_free":
        c0 = drv.global_cost(torch, "cpu")
        drv.enable_peer_access()
        # no rendezvous from here to the end of the run; the ranks take different numbers of steps on purpose
        drv.free_run(300 + 150 * rank)
        be.sync()
        dist.barrier()
        drv.exchange_all()
        extra["cost0"] = c0
    else:
        for rnd in range(2):
            for _ in range(2 * N):
                drv.step()
            drv.update_weights()
        for _ in range(N):
            drv.step()
    cost = drv.global_cost(torch, "cpu")
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), cost=cost, **extra,
             **{"X%d" % a: be.team.agents[a].get_X() for a in mine},
             **{"W%d" % a: be.team.agents[a].measurements()["weight"] for a in mine})
    dist.barrier()
    be.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ticks", "gnc"])
def test_two_processes_one_gpu_ticks_and_weight_rounds(mode):
    import torch.multiprocessing as mp_
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_worker, args=(2, port, mode, d), nprocs=2, join=True)
        outs = [np.load(os.path.join(d, "rank%d.npz" % r)) for r in range(2)]
    N, mp, n, T, kw = _problem(mode)
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
        assert np.abs(outs[a % 2]["X%d" % a] - ref.agents[a].get_X()).max() < 1e-7, a
        assert np.abs(outs[a % 2]["W%d" % a] - ref.agents[a].measurements()["weight"]).max() < 1e-7
    assert abs(float(outs[0]["cost"]) - ref.cost()) <= 1e-8 * abs(ref.cost())


def _spawn(mode):
    import torch.multiprocessing as mp_
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp_.spawn(_worker, args=(2, port, mode, d), nprocs=2, join=True)
        return [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(2)]


def test_peer_access_synchronous_schedule_reads_neighbours_in_place():
    """accelerated RBCD++ over two processes with NO pose message after the first exchange: every neighbour pose is a
    load from the other process's arrays (HIP IPC).  Same iterates as the oracle's sequential schedule."""
    outs = _spawn("peer_sync")
    N, mp, n, T, kw = _problem("peer_sync")
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    ref.set_initial(T, O.fixed_stiefel(5))
    for _ in range(4 * N):
        ref.iterate()
    for a in range(N):
        assert np.abs(outs[a % 2]["X%d" % a] - ref.agents[a].get_X()).max() < 1e-7, a
    first = sum(1 for a in range(N) for b in range(N) if a != b and a % 2 != b % 2)  # exchange_all, one op per ordered pair
    assert int(outs[0]["messages"]) <= first and int(outs[1]["messages"]) <= first


def test_peer_access_free_running_asynchronous_mode_descends():
    """BASELINE configs[4]'s mode proper across processes: unsynchronised steps from whatever the neighbour arrays
    hold.  Not reproducible bit for bit by construction; what it promises is descent to the synchronous answer."""
    outs = _spawn("peer_free")
    N, mp, n, T, kw = _problem("peer_free")
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    ref.set_initial(T, O.fixed_stiefel(5))
    for _ in range(300):
        ref.exchange_all()
        for a in ref.agents:
            a.iterate(True)
    ref.exchange_all()
    c0, c = float(outs[0]["cost0"]), float(outs[0]["cost"])
    assert np.isfinite(c) and c < 0.05 * c0
    # (how far the run gets depends on how the two processes' unsynchronised steps interleave: 12 runs in a row ended
    # between 0.988 and 1.50 of the synchronous reference's cost, 11 of them within 2.3 % -- and all below 1.3 % of the
    # initial cost; profiles/experiments/peer_free_spread.py)
    assert c <= 2.0 * ref.cost() and c >= 0.5 * ref.cost()


def test_peer_access_per_agent_iterate_reads_neighbours_in_place():
    """dpgo_agent_iterate on teams that imported their remote neighbours: the sequential schedule with NO pose message
    and no update_neighbor_poses call after the first exchange -- the oracle's sequential RBCD iterates"""
    outs = _spawn("peer_agent_api")
    N, mp, n, T, kw = _problem("peer_agent_api")
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    ref.set_initial(T, O.fixed_stiefel(5))
    for _ in range(3 * N):
        ref.iterate()
    for a in range(N):
        assert np.abs(outs[a % 2]["X%d" % a] - ref.agents[a].get_X()).max() < 1e-7, a


@pytest.mark.parametrize("mode", ["peer_token", "peer_token_rgd", "peer_token_plain"])
def test_device_side_update_token_over_peer_access(mode):
    """The synchronous schedule across two processes with the host out of the loop (dpgo_team_run_peer): each process
    enqueues the schedule, wait / signal kernels around the launches that read the other process's poses in place -- or
    overwrite what it was reading -- order the two streams through mailboxes written over HIP IPC.  RTR + Nesterov with
    restarts, RGD + Nesterov, plain RTR: the oracle's sequential iterates, no message after the first exchange."""
    outs = _spawn(mode)
    N, mp, n, T, kw = _problem(mode)
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    ref.set_initial(T, O.fixed_stiefel(5))
    for _ in range(4 * N):
        ref.iterate()
    for a in range(N):
        assert np.abs(outs[a % 2]["X%d" % a] - ref.agents[a].get_X()).max() < 1e-7, a
    assert abs(float(outs[0]["cost"]) - ref.cost()) <= 1e-8 * abs(ref.cost())
    first = sum(1 for a in range(N) for b in range(N) if a != b and a % 2 != b % 2)
    assert int(outs[0]["messages"]) <= first and int(outs[1]["messages"]) <= first
