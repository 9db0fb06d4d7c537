"""The exchange between ranks carried by RCCL from INSIDE the library (csrc/rank_exchange.cpp, include/dpgo_hip.h
dpgo_comm_* / dpgo_team_run_ranks): ncclSend / ncclRecv really execute on the MI355X.

A one-GPU box cannot hold two ranks of one communicator (RCCL refuses a duplicate device), so the message path is
exercised end to end in LOOPBACK: world size 1, every neighbour pair -- co-resident ones included -- exchanges its
public-pose slabs through a grouped self-send (legal point-to-point traffic) and NOTHING is read in place.  The iterates
must be those of the host-driven message schedule (dpgo_team_step_begin / pack / unpack / dpgo_team_step_end, the path
`DistributedRBCD.step` drives and tests/test_distributed_gloo.py pins on the single-process schedule) bit for bit: they
are only if every slab crossed RCCL intact, in the right order, into the right slots.

Replaces src/PGOAgentROS.cpp:662-690 (publishPublicPoses), :1255-1284 (publicPosesCallback), :136-149 (staleness gate)."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load

pytestmark = pytest.mark.gpu


def _teams(dataset, N, r=5, **kw):
    m, mp, n = load(dataset, N)
    T, Y = O.odometry_init(m, n), O.fixed_stiefel(r)
    prm = capi.default_params(r=r, num_robots=N, **kw)

    def make():
        t = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), prm)
        t.set_initial(T, Y)
        return t
    return make, mp, n, T, Y


def _host_driven_schedule(team, sels):
    """the reference schedule of the split iteration: iterate(false) part of everyone, then the token holder's block
    update with its neighbours read in place (what DistributedRBCD.step drives on each rank; pinned on the
    single-process schedule and on the oracle by tests/test_distributed_gloo.py and tests/test_gpu_parity.py)"""
    for sel in sels:
        team.step_begin(sel)
        team.step_end(sel)
    team.synchronize()


CASES = [
    ("smallGrid3D", 2, dict(method=1, acceleration=1, rgd_stepsize=0.05, restart_interval=7), 30),
    ("sphere2500", 5, dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=20), 45),
    ("sphere2500", 5, dict(method=0, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=1e-2, restart_interval=10), 23),
    ("sphere2500", 5, dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5), 12),
]


@pytest.mark.parametrize("dataset,N,kw,iters", CASES)
def test_loopback_self_sends_carry_the_schedule_bit_for_bit(dataset, N, kw, iters):
    make, mp, n, T, Y = _teams(dataset, N, **kw)
    sels = [k % N for k in range(iters)]
    ref = make()
    _host_driven_schedule(ref, sels)
    Xref = ref.global_X()
    fref = ref.cost()
    ref.close()

    comm = capi.Comm(capi.comm_unique_id(), 0, 1, device=0)
    t = make()
    t.attach_comm(comm, [0] * N, loopback=True)
    t.exchange_all_ranks()
    # K iterations per host call, in uneven pieces (the gate's bookkeeping must carry across calls)
    cut = max(1, iters // 3)
    t.run_ranks(sels[:cut])
    t.run_ranks(sels[cut:])
    t.synchronize()
    c = t.comm_counters()
    X = t.global_X()
    f = comm.global_cost(t)
    assert np.array_equal(X, Xref), "max |dX| = %.3e" % np.abs(X - Xref).max()
    assert abs(f - fref) <= 1e-12 * abs(fref)
    # every iteration moved exactly one message each way through RCCL (one per pair of ranks), plus the full exchange
    # after set_initial
    per_iter = c["messages_sent"] - 1
    if kw["acceleration"]:
        assert per_iter == iters and c["messages_received"] == c["messages_sent"]
    else:
        # plain RBCD: a neighbour that has not moved since it last published is not re-sent (the staleness gate at 0)
        assert 0 < per_iter <= iters
    assert c["bytes_sent"] == c["bytes_received"] > 0
    t.close()
    comm.close()


def test_staleness_gate_withholds_messages_and_changes_the_iterates():
    """max_delayed_iterations (struct default 3, include/dpgo_ros/PGOAgentROS.h:83; 7 here: with five robots in round robin
    a neighbour's copy is 5 iterations old at every turn): a copy up to 7 iterations old is good enough, so fewer slabs
    cross -- and the iterates differ from the fresh-copy schedule while the cost still falls"""
    N = 5
    kw = dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=20)
    make, mp, n, T, Y = _teams("sphere2500", N, **kw)
    sels = [k % N for k in range(40)]
    out = {}
    for delay in (0, 7):
        comm = capi.Comm(capi.comm_unique_id(), 0, 1, device=0)
        t = make()
        f0 = t.cost()
        t.attach_comm(comm, [0] * N, max_delayed_iterations=delay, loopback=True)
        t.exchange_all_ranks()
        t.run_ranks(sels)
        t.synchronize()
        out[delay] = (t.comm_counters()["bytes_sent"], comm.global_cost(t), f0)
        t.close()
        comm.close()
    assert out[7][0] < 0.6 * out[0][0]
    assert out[7][1] < out[7][2] and out[0][1] < out[0][2]
    assert out[7][1] != out[0][1]


def test_world_size_one_without_loopback_is_the_device_resident_schedule():
    """nothing crosses a rank: dpgo_team_run_ranks hands the list to the device-resident schedule (hipGraphs, one-launch
    iterations) -- same bits as dpgo_team_run, no message"""
    N = 5
    kw = dict(method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=20)
    make, mp, n, T, Y = _teams("sphere2500", N, **kw)
    a = make()
    a.run(64)
    a.synchronize()
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, device=0)
    b = make()
    b.attach_comm(comm, [0] * N)
    b.exchange_all_ranks()
    b.run_ranks([k % N for k in range(64)])
    b.synchronize()
    assert np.array_equal(a.global_X(), b.global_X())
    assert b.comm_counters()["messages_sent"] == 0
    assert abs(comm.global_cost(b) - a.cost()) <= 1e-12 * abs(a.cost())
    assert comm.allreduce([1.5, -2.0])[0] == 1.5 and comm.allreduce([1.5, -2.0], op="max")[1] == -2.0
    a.close()
    b.close()
    comm.close()


def test_loopback_with_seven_peers_per_robot_tunnels():
    """BASELINE configs[4]'s graph (MIT tunnels, 8 robots, every robot a neighbour of the seven others, 85 - 137 public
    poses each way): the batches of a full exchange carry more slabs than one pack / unpack launch takes (16), and an
    accelerated iteration moves 14 of them to one receiver.  Same bits as the in-place schedule."""
    from tests.util import load_tunnels
    N = 8
    m = load_tunnels(1)
    nk = [0] * N
    for e in m:
        nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1)
        nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
    Ts = []
    for k in range(N):
        odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
        odo["r1"] = 0
        odo["r2"] = 0
        Ts.append(O.odometry_init(odo, nk[k]))
    T, Y = np.concatenate(Ts), O.fixed_stiefel(5)
    prm = capi.default_params(r=5, num_robots=N, method=1, acceleration=1, rgd_stepsize=0.2, restart_interval=9)

    def make():
        t = capi.Team.from_measurements(m.view(capi.MEAS_DTYPE), prm)
        t.set_initial(T, Y)
        return t
    sels = [(3 * k) % N for k in range(26)]  # (not round robin: the gate's bookkeeping sees uneven gaps)
    ref = make()
    _host_driven_schedule(ref, sels)
    Xref = ref.global_X()
    ref.close()
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, device=0)
    t = make()
    t.attach_comm(comm, [0] * N, loopback=True)
    t.exchange_all_ranks()
    t.run_ranks(sels)
    t.synchronize()
    assert np.array_equal(t.global_X(), Xref)
    c = t.comm_counters()
    assert c["messages_sent"] == 1 + len(sels)
    # the full exchange: every ordered pair of neighbours, both sequences
    npub = sum(len(t.agents[a].public_pose_ids(b)) for a in range(N) for b in t.agents[a].neighbors())
    per_iter = [sum(len(t.agents[b].public_pose_ids(s)) for b in t.agents[s].neighbors()) for s in sels]
    assert c["bytes_sent"] == 8 * 4 * 5 * 2 * (npub + sum(per_iter))
    t.close()
    comm.close()


def test_library_side_ticks_and_colour_classes_in_loopback():
    """the other two multi-rank schedules with their slabs moved by the library: lockstep ASAPP ticks on the tunnels graph
    (dpgo_team_run_simultaneous_ranks) and the classes of a colour-parallel sweep on sphere2500 / 5
    (dpgo_team_run_group_ranks) -- loopback self-sends against the in-place single-team runs, bit for bit"""
    from tests.util import load_tunnels
    # ---- ticks
    N = 8
    m = load_tunnels(1)
    nk = [0] * N
    for e in m:
        nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1)
        nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
    Ts = []
    for k in range(N):
        odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
        odo["r1"] = 0
        odo["r2"] = 0
        Ts.append(O.odometry_init(odo, nk[k]))
    T, Y = np.concatenate(Ts), O.fixed_stiefel(5)
    prm = capi.default_params(r=5, num_robots=N, method=1, acceleration=0, rgd_stepsize=0.2)
    a = capi.Team.from_measurements(m.view(capi.MEAS_DTYPE), prm)
    a.set_initial(T, Y)
    for _ in range(12):
        a.run_simultaneous(1)
    a.synchronize()
    comm = capi.Comm(capi.comm_unique_id(), 0, 1, device=0)
    b = capi.Team.from_measurements(m.view(capi.MEAS_DTYPE), prm)
    b.set_initial(T, Y)
    b.attach_comm(comm, [0] * N, loopback=True)
    b.exchange_all_ranks()
    b.run_simultaneous_ranks(5)
    b.run_simultaneous_ranks(7)
    b.synchronize()
    assert np.array_equal(a.global_X(), b.global_X())
    assert b.comm_counters()["messages_sent"] == 1 + 12
    a.close()
    b.close()
    # ---- colour classes
    N = 5
    kw = dict(method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=1e-2)
    make, mp, n, T, Y = _teams("sphere2500", N, **kw)
    a = make()
    nc, col = a.coloring()
    groups = [[k for k in range(N) if col[k] == c] for c in range(nc)]
    a.run_colored(3)
    a.synchronize()
    b = make()
    b.attach_comm(comm, [0] * N, loopback=True)
    b.exchange_all_ranks()
    b.set_groups(groups)
    for _ in range(3):
        for g, mem in enumerate(groups):
            b.run_group_ranks(g, len(mem))
    b.synchronize()
    assert np.array_equal(a.global_X(), b.global_X())
    assert abs(comm.global_cost(b) - a.cost()) <= 1e-12 * abs(a.cost())
    a.close()
    b.close()
    comm.close()
