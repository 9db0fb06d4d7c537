"""tunnels_gnc_pin.py -- pins the robust path on the ONE solver output the reference tree holds.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).

`/root/reference/data/tunnels/robot{0..7}/measurements.csv` (committed here under data/tunnels/, data files only)
carry a 15th column `weight` (`data/tunnels/robot0/measurements.csv:1`): the GNC-TLS weights a real dpgo run left
behind.  This script re-runs the configuration of `launch/dpgo_gnc_demo.launch:27-43` on the same
measurements -- 8 robots, wrapper weighting kappa = 1e4 / tau = 1e2 (`src/utils.cpp:141-142`), odometry guess,
RTR 3 / 50 / 0.5, GNC-TLS barc 3, mu0 1e-5, mu step 2, 3 weight updates, 50 inner iterations per robot, RoundRobin,
no acceleration -- with the CPU oracle and compares the non-fixed weights after every UPDATE_WEIGHT round
(`src/PGOAgentROS.cpp:1211-1233,1315-1353`) with the file's column.

    python -m oracle.tunnels_gnc_pin                 # the table of the default configuration
    python -m oracle.tunnels_gnc_pin --sweep         # the table under each recalled constant

The same functions are imported by tests/test_tunnels_gnc_pin.py (oracle, CPU) and tests/test_gpu_tunnels_gnc_pin.py
(HIP path through the C-ABI).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402

TUNNELS = os.path.join(ROOT, "data", "tunnels")
NUM_ROBOTS = 8

# launch/dpgo_gnc_demo.launch:27-43 (and PGOAgentROSNode.cpp:216-218: inner iterations = per robot x robots)
DEMO = dict(method=O.METHOD_RTR, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=0.5, acceleration=0,
            rel_change_tol=0.2, robust_cost_type=O.COST_GNC_TLS, gnc_barc=3.0, gnc_mu_step=2.0, gnc_init_mu=1e-5,
            robust_opt_num_weight_updates=3, robust_opt_num_resets=3, robust_opt_inner_iters=50 * NUM_ROBOTS,
            robust_opt_min_convergence_ratio=0.0, max_num_iters=100000)


def edge_key(e):
    return (int(e["r1"]), int(e["p1"]), int(e["r2"]), int(e["p2"]))


def load(reader=O.read_csv, dtype=O.MEAS_DTYPE):
    """(wrapper-weighted edge list, one row per edge; file weights; file is_known_inlier flags; poses per robot).

    The edge list is what the wrapper hands the solver: the codec drops kappa / tau / weight / is_known_inlier and
    re-derives them (`src/utils.cpp:108-152`).  The file's own columns are returned beside it, keyed like the list;
    a shared edge is listed by both of its robots and the two copies must agree."""
    seen, rows, wfile, inl = {}, [], [], []
    for k in range(NUM_ROBOTS):
        path = os.path.join(TUNNELS, "robot%d" % k, "measurements.csv")
        wrapped = reader(path, O.WEIGHT_WRAPPER)
        asfile = reader(path, O.WEIGHT_LIBRARY)
        assert len(wrapped) == len(asfile)
        for e, f in zip(wrapped, asfile):
            key = edge_key(e)
            if key in seen:
                q = seen[key]
                # the two copies differ in the sixth printed digit on 31 of 3548 shared edges: the higher-ID robot holds
                # the owner's weight after the float32 wire (msg/RelativeMeasurementWeights.msg:8); the owner's copy
                # (lower ID, read first) is kept
                assert abs(wfile[q] - f["weight"]) <= 1e-5 * max(wfile[q], 1e-12) and inl[q] == f["is_known_inlier"], key
                continue
            seen[key] = len(rows)
            rows.append(e)
            wfile.append(float(f["weight"]))
            inl.append(int(f["is_known_inlier"]))
    m = np.array(rows, dtype=dtype)
    nk = [0] * NUM_ROBOTS
    for e in m:
        nk[e["r1"]] = max(nk[e["r1"]], int(e["p1"]) + 1)
        nk[e["r2"]] = max(nk[e["r2"]], int(e["p2"]) + 1)
    return m, np.array(wfile), np.array(inl, dtype=bool), nk


def odometry_guess(m, nk):
    """every robot chains its own odometry from the identity (local_initialization_method Odometry,
    `launch/dpgo_gnc_demo.launch:31`); the frames are NOT aligned: the deliberately crude common guess of
    tests/test_gpu_parity.py::test_tunnels_eight_agents"""
    Ts = []
    for k in range(NUM_ROBOTS):
        odo = m[(m["r1"] == k) & (m["r2"] == k) & (m["p1"] + 1 == m["p2"])].copy()
        odo["r1"] = 0
        odo["r2"] = 0
        Ts.append(O.odometry_init(odo.view(O.MEAS_DTYPE), nk[k]))
    return np.concatenate(Ts)


def aligned_odometry_guess(m, nk, min_inliers=3):
    """per-robot odometry chains, then robot k > 0 is moved into robot 0's frame through robust averaging of the
    candidate transforms its shared loop closures with already-aligned robots give (what `initializeInGlobalFrame`
    does on the first public poses it receives, `src/PGOAgentROS.cpp:348-360`, robust_init_min_inliers 3)"""
    T = odometry_guess(m, nk).reshape(-1, 4, 3)  # per pose: 3 x 4 column-major -> [col][row]
    off = np.concatenate([[0], np.cumsum(nk)])
    done = {0}
    while len(done) < NUM_ROBOTS:
        progressed = False
        for k in range(NUM_ROBOTS):
            if k in done:
                continue
            cands = []
            for e in m:
                a, b = int(e["r1"]), int(e["r2"])
                if a == b or k not in (a, b):
                    continue
                other = b if a == k else a
                if other not in done:
                    continue
                R, t = e["R"].reshape(3, 3), e["t"]
                Ta, Tb = T[off[a] + e["p1"]], T[off[b] + e["p2"]]
                Ra, ta, Rb, tb = Ta[:3].T, Ta[3], Tb[:3].T, Tb[3]
                if a == k:  # T_world_b = T_world_k,a * (R,t)  ->  T_align = T_world_b (R,t)^-1 T_k,a^-1
                    Rw = Rb @ R.T @ Ra.T
                    tw = tb - Rw @ ta - Rb @ R.T @ t
                else:      # T_world_k,b = T_world_a (R,t)
                    Rw = Ra @ R @ Rb.T
                    tw = ta + Ra @ t - Rw @ tb
                cands.append(np.concatenate([Rw.T.reshape(-1), tw]))
            if len(cands) < min_inliers:
                continue
            got = O.robust_frame_alignment(np.array(cands), min_inliers=min_inliers)
            if got is None:
                continue
            A = got[0].reshape(4, 3)
            Rw, tw = A[:3].T, A[3]
            for i in range(off[k], off[k + 1]):
                Ri, ti = T[i][:3].T, T[i][3]
                T[i][:3] = (Rw @ Ri).T
                T[i][3] = Rw @ ti + tw
            done.add(k)
            progressed = True
        if not progressed:
            break
    return T.reshape(-1)


def compare(w, wfile, free):
    """deviation of the non-fixed weights `w[free]` from the file's: Spearman rank correlation; median and 90th
    percentile of |w - w_file| / w_file over the edges whose file weight is not zero (the file's weights are all far
    below 1: median 7e-4, i.e. GNC's middle branch w = barc sqrt(mu (mu + 1)) / residual - mu, so this is essentially the
    relative deviation of the residuals); zeros ours / file; share of the file's zero set we also have at zero"""
    a, b = w[free], wfile[free]

    def ranks(x):
        order = np.argsort(x, kind="stable")
        rk = np.empty(len(x))
        rk[order] = np.arange(len(x))
        for v in np.unique(x):  # average the ranks of ties
            sel = x == v
            if sel.sum() > 1:
                rk[sel] = rk[sel].mean()
        return rk

    spearman = float(np.corrcoef(ranks(a), ranks(b))[0, 1])
    za, zb = a == 0.0, b == 0.0
    rel = np.abs(a - b)[~zb] / b[~zb]
    signed = ((a - b)[~zb] / b[~zb])
    overlap = float((za & zb).sum()) / max(1, zb.sum())
    return dict(spearman=spearman, median=float(np.median(rel)), p90=float(np.quantile(rel, 0.9)),
                signed_median=float(np.median(signed)), zeros=int(za.sum()), zeros_file=int(zb.sum()),
                zero_overlap=overlap)


def robot_pair(m):
    lo, hi = np.minimum(m["r1"], m["r2"]), np.maximum(m["r1"], m["r2"])
    return lo * NUM_ROBOTS + hi


# The 139 loop closures between robots 1 and 2 are the one group of the file that NO setting of this path reproduces:
# their file weights correspond to residuals 5 to 30 times ours, falling smoothly from ~430 at robot 2's first poses to
# ~45 beyond its 100th, while each of the other 35 groups (28 robot pairs, 7 private sets) agrees with the oracle to
# about 1 % in the median, Spearman 0.998 and an IDENTICAL zero set (58 of 58).  What robot 1 (their owner) saw of robot
# 2's trajectory when it weighted them in the recorded run cannot be recovered from the file (stale or dropped
# PublicPoses messages are the obvious candidate: that run was 8 ROS processes); the tests report both figures.
ANOMALOUS_PAIR = 1 * NUM_ROBOTS + 2


def per_pair(w, wfile, m, free):
    """median / p90 relative deviation per robot pair (pair id lo * 8 + hi -> (count, median, p90, signed median))"""
    out = {}
    pid = robot_pair(m)
    for g in np.unique(pid):
        sel = free & (pid == g) & (wfile > 0)
        if sel.sum() == 0:
            continue
        d = (w[sel] - wfile[sel]) / wfile[sel]
        out[int(g)] = (int(sel.sum()), float(np.median(np.abs(d))), float(np.quantile(np.abs(d), 0.9)), float(np.median(d)))
    return out


def team_weights(team, m):
    """the team's current weight of every edge of `m` (the owner's copy: the lower-ID endpoint, PGOAgentROS.cpp:732)"""
    index = {edge_key(e): q for q, e in enumerate(m)}
    w = np.full(len(m), np.nan)
    fixed = np.zeros(len(m), dtype=bool)
    agents = team.agents
    for k in range(NUM_ROBOTS):
        for e in agents[k].measurements():
            if min(int(e["r1"]), int(e["r2"])) != k:
                continue
            q = index[edge_key(e)]
            w[q] = e["weight"]
            fixed[q] = bool(e["fixed_weight"])
    assert not np.isnan(w).any()
    return w, fixed


def summarize(team, m, wfile):
    w, fixed = team_weights(team, m)
    free = ~fixed
    res = compare(w, wfile, free)
    rest = compare(w, wfile, free & (robot_pair(m) != ANOMALOUS_PAIR))
    res.update({"rest_" + k: v for k, v in rest.items()})
    res["weights"], res["fixed"] = w, fixed
    return res


def run_rounds(team, m, wfile, rounds=5):
    """The synchronous schedule of the demo through `run_schedule` (the leader's UPDATE_WEIGHT decisions,
    src/PGOAgentROS.cpp:206-214), stopped right after every UPDATE_WEIGHT round: the leader sees 400 inner iterations
    when it holds the token at iteration 401, 801, 1201, ...  `team` is an oracle Team or a capi.Team (same calls).
    The team's robust_opt_num_weight_updates must be >= rounds."""
    out = []
    inner = DEMO["robust_opt_inner_iters"]
    for rnd in range(rounds):
        done, term, wr = team.run_schedule(inner + 1 if rnd == 0 else inner)
        assert (done, term, wr) == (inner + 1 if rnd == 0 else inner, False, 1), (rnd, done, term, wr)
        out.append(summarize(team, m, wfile))
    return out


def run_rounds_manual(team, m, wfile, rounds=5, inner=None, reset_to_guess=False, early=False):
    """oracle only, for the sweep: `inner` block updates per round; `reset_to_guess`: X <- the initial guess after every
    UPDATE_WEIGHT round (one reading of robust_opt_num_resets = num_weight_updates = 3, launch/dpgo_gnc_demo.launch:39-41);
    `early`: the round ends as soon as the leader finds every robot ready to terminate"""
    inner = inner or DEMO["robust_opt_inner_iters"]
    X0 = [a.get_X() for a in team.agents]
    out = []
    for rnd in range(rounds):
        k = 0
        while True:
            sel = team.iterate()
            k += 1
            if sel != 0:
                continue
            if k >= inner or (early and all(a.status().ready_to_terminate for a in team.agents)):
                break
        team.update_weights()
        out.append(summarize(team, m, wfile))
        out[-1]["iterations"] = k
        if reset_to_guess:
            for a, x in zip(team.agents, X0):
                a.set_X(x)
            team.exchange_all()
    return out


def oracle_team(m, nk, T, r=5, **over):
    kw = dict(DEMO)
    kw.update(over)
    p = O.default_params(r=r, num_robots=NUM_ROBOTS, **kw)
    t = O.Team(m, sum(nk), p)
    t.set_initial(T, O.fixed_stiefel(r))
    return t


def table(rows):
    print("| after update # | Spearman | median rel. dev. | p90 | zeros (ours / file) | file's zeros shared |"
          " without robots 1-2: Spearman | median | p90 | zeros |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for k, r in enumerate(rows):
        print("| %d | %.3f | %.2f %% | %.2f %% | %d / %d | %.0f %% | %.4f | %.2f %% | %.2f %% | %d / %d |" % (
            k + 1, r["spearman"], 100 * r["median"], 100 * r["p90"], r["zeros"], r["zeros_file"], 100 * r["zero_overlap"],
            r["rest_spearman"], 100 * r["rest_median"], 100 * r["rest_p90"], r["rest_zeros"], r["rest_zeros_file"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--pairs", action="store_true", help="per robot pair deviations after the third update")
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    m, wfile, inl, nk = load()
    print("edges %d, non-fixed %d, poses per robot %s" % (len(m), int((~inl).sum()), nk))
    T = aligned_odometry_guess(m, nk)
    print("\n### the demo configuration (launch/dpgo_gnc_demo.launch:27-43; recalled constants at their defaults)")
    t = oracle_team(m, nk, T, robust_opt_num_weight_updates=args.rounds)
    rows = run_rounds(t, m, wfile, rounds=args.rounds)
    table(rows)
    if args.pairs:
        pp = per_pair(rows[2]["weights"], wfile, m, ~rows[2]["fixed"])
        print("\n| robots | edges | median | p90 | signed median |\n|---|---|---|---|---|")
        for g, (cnt, med, p90, sg) in sorted(pp.items()):
            print("| %d-%d | %d | %.2f %% | %.2f %% | %+.2f %% |" % (g // NUM_ROBOTS, g % NUM_ROBOTS, cnt, 100 * med, 100 * p90, 100 * sg))
    if not args.sweep:
        return
    T0 = odometry_guess(m, nk)
    variants = [
        ("defaults (manual loop, 400 per round)", dict(), T, dict()),
        ("trust-region radius 30 (max 150)", dict(rtr_initial_radius=30.0, rtr_max_radius=150.0), T, dict()),
        ("trust-region radius 10 (max 50)", dict(rtr_initial_radius=10.0, rtr_max_radius=50.0), T, dict()),
        ("trust-region radius 300 (max 1500)", dict(rtr_initial_radius=300.0, rtr_max_radius=1500.0), T, dict()),
        ("preconditioner shift 1e-3", dict(precond_shift=1e-3), T, dict()),
        ("fp32 weights on the wire", dict(weights_as_float32=1), T, dict()),
        ("status refreshed by every iterate", dict(status_every_iterate=1), T, dict()),
        ("acceleration (restart 50)", dict(acceleration=1, restart_interval=50), T, dict()),
        ("tCG cap 10", dict(rtr_tcg_iterations=10), T, dict()),
        ("RTR 1 outer iteration", dict(rtr_iterations=1), T, dict()),
        ("gradnorm tol 1e-2", dict(gradnorm_tol=1e-2), T, dict()),
        ("rank 3", dict(r=3), T, dict()),
        ("mu stepped BEFORE the weights (first update at 2e-5)", dict(gnc_init_mu=2e-5), T, dict()),
        ("barc 5 (the node's default, PGOAgentROSNode.cpp:205)", dict(gnc_barc=5.0), T, dict()),
        ("frames not aligned (every robot chains from the identity)", dict(), T0, dict()),
        ("200 iterations per round", dict(), T, dict(inner=200)),
        ("800 iterations per round", dict(), T, dict(inner=800)),
        ("2000 iterations per round", dict(), T, dict(inner=2000)),
        ("X reset to the guess after every update", dict(), T, dict(reset_to_guess=True)),
        ("round ends early when every robot is ready", dict(), T, dict(early=True)),
    ]
    print("\n### sweep: the third update (and where the minimum over 5 updates sits)")
    print("| variant | minimum at | Spearman | median | p90 | signed median | without robots 1-2: Spearman | median | p90 | zeros ours / file |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, over, T0_, how in variants:
        t = oracle_team(m, nk, T0_, **over)
        rows = run_rounds_manual(t, m, wfile, rounds=args.rounds, **how)
        best = int(np.argmin([r["median"] for r in rows])) + 1
        r3 = rows[2]
        print("| %s | %d | %.3f | %.2f %% | %.2f %% | %+.2f %% | %.4f | %.2f %% | %.2f %% | %d / %d |" % (
            name, best, r3["spearman"], 100 * r3["median"], 100 * r3["p90"], 100 * r3["signed_median"],
            r3["rest_spearman"], 100 * r3["rest_median"], 100 * r3["rest_p90"], r3["rest_zeros"], r3["rest_zeros_file"]))


if __name__ == "__main__":
    main()
