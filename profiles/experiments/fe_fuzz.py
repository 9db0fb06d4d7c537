"""randomised comparison of the one-launch iteration with the two-launch sequence: sphere2500 split 5 .. 9 ways, r = 3 / 4 / 5,
random extra loop closures (intra- and inter-robot, copies of real measurements between random poses), random restart
interval and step size, GNC re-weighting in between; bitwise equality after every run, and a count of how often the
one-launch form was actually taken"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from dpgo_ros_amd import capi

m0, n = capi.read_g2o(os.path.join(bench.ROOT, "data", "sphere2500.g2o"))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
taken = carried = 0
for case in range(cases):
    robots = int(rng.integers(5, 10))
    r = int(rng.choice([3, 4, 5]))
    extra = int(rng.integers(0, 120))
    m = m0.copy()
    if extra:
        add = m[rng.integers(0, len(m), extra)].copy()
        for e in add:
            i, j = rng.integers(0, n, 2)
            while abs(int(i) - int(j)) < 2:
                i, j = rng.integers(0, n, 2)
            e["p1"], e["p2"] = min(i, j), max(i, j)
        m = np.concatenate([m, add])
    mp = capi.partition(m, n, robots)
    kw = dict(method=1, acceleration=1, rgd_stepsize=float(rng.choice([0.05, 0.1, 0.2])), rgd_use_preconditioner=1,
              restart_interval=int(rng.integers(3, 40)))
    robust = bool(rng.integers(0, 2))
    if robust:
        kw.update(robust_cost_type=5, gnc_barc=5.0)
    T, Y = capi.odometry_init(m0, n), capi.fixed_stiefel(r)
    teams = []
    for fe in (0, 1):
        os.environ["DPGO_FUSED_EVAL"] = str(fe)
        t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=robots, **kw), device=0)
        t.set_initial(T, Y)
        teams.append(t)
    for chunk in rng.integers(1, 400, 4):
        for t in teams:
            t.run(int(chunk)); t.synchronize()
        d = max(float(np.max(np.abs(teams[0].agents[k].get_X() - teams[1].agents[k].get_X()))) for k in teams[0].ids)
        assert d == 0.0, (case, robots, r, extra, kw, int(chunk), d)
        if robust:
            assert teams[0].update_weights() == teams[1].update_weights()
    c7, c8 = teams[1].counters()[7], teams[1].counters()[8]
    taken += c7 > 0
    carried += c8 > 0
    print("case %2d: %d robots r=%d +%3d edges restart %2d step %.2f robust %d -> one-launch iterations %d (%d with carried rows), bitwise equal"
          % (case, robots, r, extra, kw["restart_interval"], kw["rgd_stepsize"], robust, c7, c8), flush=True)
    for t in teams:
        t.close()
print("fuzz ok: %d cases, one-launch form taken in %d, carried rows in %d" % (cases, taken, carried))
