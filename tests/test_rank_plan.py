"""The N > 1 exchange of csrc/rank_exchange.cpp is correct BY CONSTRUCTION only if every rank derives the same messages: what
rank r sends to rank p in a batch must be what p posts a receive for -- same length, same slabs in the same order -- or the
grouped ncclSend / ncclRecv deadlock or scatter poses into the wrong slots.  No box available here holds two GPUs, so the
planning layer (the code dpgo_team_run_ranks runs, host arithmetic only) is replayed for EVERY rank of worlds of 2 .. 8 through
dpgo_rank_plan_simulate and the two ends of every message are compared, batch by batch; the slabs that cross are also checked
against an independent model of the staleness gate (src/PGOAgentROS.cpp:136-149) written here.  CPU test: no device needed."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from dpgo_ros_amd.distributed import topology
from tests.util import load, load_tunnels


def _graph(name):
    if name == "tunnels":
        return load_tunnels(1), 8
    ds, N = name.split("/")
    return load(ds, int(N))[1], int(N)


def _npub(meas, N):
    nbrs, npub = topology(meas, N)
    M = np.zeros((N, N), dtype=np.int32)
    for (b, a), c in npub.items():  # poses of b that appear in edges with a = what a needs of b
        M[b, a] = c
    return nbrs, M


def _gate_model(nbrs, M, owner, sels, accel, delay, r=5):
    """doubles that cross ranks per batch (the full exchange first): an independent restatement of the gate"""
    N = len(owner)
    seqs = 2 if accel else 1
    out = [sum(int(M[b, a]) for a in range(N) for b in nbrs[a] if owner[a] != owner[b]) * 2 * 4 * r]
    version, sent = [0] * N, {}
    for a in range(N):
        for b in nbrs[a]:
            sent[(b, a)] = 0
    for k, sel in enumerate(sels):
        if accel:
            version = [k + 1 if a != sel else version[a] for a in range(N)]
        tot = 0
        for b in nbrs[sel]:
            if owner[b] == owner[sel]:
                continue
            behind = version[b] - sent[(b, sel)]
            if behind == 0 or behind <= delay:
                continue
            sent[(b, sel)] = version[b]
            tot += int(M[b, sel]) * seqs * 4 * r
        out.append(tot)
        version[sel] = k + 1
    return out


@pytest.mark.parametrize("graph", ["sphere2500/5", "sphere2500/8", "tunnels", "smallGrid3D/3"])
@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("accel,delay", [(1, 0), (0, 0), (1, 3), (0, 2)])
def test_every_send_meets_its_receive(graph, world, accel, delay):
    meas, N = _graph(graph)
    nbrs, M = _npub(meas, N)
    owner = [a % world for a in range(N)]
    rng = np.random.default_rng(7)
    sels = [k % N for k in range(3 * N)] + [int(x) for x in rng.integers(0, N, 4 * N)]  # round robin, then any order
    plans = [capi.rank_plan_simulate(owner, M, rk, world, sels, acceleration=accel, max_delayed_iterations=delay) for rk in range(world)]
    for it in range(1 + len(sels)):
        for rk in range(world):
            assert plans[rk][it, rk, 0] == 0 and plans[rk][it, rk, 1] == 0        # nothing is sent to oneself
            for p in range(world):
                assert plans[rk][it, p, 0] == plans[p][it, rk, 1], (it, rk, p)     # length sent == length expected
                assert plans[rk][it, p, 2] == plans[p][it, rk, 3], (it, rk, p)     # same slabs, same order
    crossed = [int(sum(plans[rk][it, :, 0].sum() for rk in range(world))) for it in range(1 + len(sels))]
    assert crossed == _gate_model(nbrs, M, owner, sels, accel, delay)
    if world >= N and accel and delay == 0:
        assert all(c > 0 for c in crossed[1:])   # one robot per rank: every accelerated iteration moves its neighbours' slabs


def test_ranks_without_robots_take_no_part():
    """5 robots on 8 GPUs: ranks 5 - 7 own nothing, send nothing, receive nothing"""
    meas, N = _graph("sphere2500/5")
    nbrs, M = _npub(meas, N)
    owner = [a % 8 for a in range(N)]
    for rk in (5, 6, 7):
        plan = capi.rank_plan_simulate(owner, M, rk, 8, [k % N for k in range(20)])
        assert not plan.any()
