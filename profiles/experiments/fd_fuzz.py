"""randomised comparison of the deep-carried one-launch iteration (csrc/step_deep.hip) -- and of its persistent form
(csrc/step_persist.hip) -- with the two-launch sequence: sphere2500 over 5 robots, r = 3 / 4 / 5, random extra loop closures
INSIDE the robots (longer rows, the private chunks stay private) and between poses that are public already (more shared edges
per pose), random restart interval and step, GNC re-weighting in between; bitwise equality after every run
usage: fd_fuzz.py [seed] [cases]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch  # noqa: F401
import bench
from dpgo_ros_amd import capi

m0, n = capi.read_g2o(os.path.join(bench.ROOT, "data", "sphere2500.g2o"))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
robots, per = 5, 500
deep_taken = 0
for case in range(cases):
    r = int(rng.choice([3, 4, 5]))
    n_in, n_x = int(rng.integers(0, 150)), int(rng.integers(0, 40))
    add = m0[rng.integers(0, len(m0), n_in + n_x)].copy()
    for q, e in enumerate(add):
        if q < n_in:   # inside one robot
            a = int(rng.integers(0, robots))
            i, j = rng.integers(0, per, 2)
            while abs(int(i) - int(j)) < 2:
                i, j = rng.integers(0, per, 2)
            i, j = a * per + int(i), a * per + int(j)
        else:          # between the last 40 poses of robot a and the first 40 of robot a + 1 (public already)
            a = int(rng.integers(0, robots - 1))
            i, j = a * per + per - 1 - int(rng.integers(0, 40)), (a + 1) * per + int(rng.integers(0, 40))
        e["p1"], e["p2"] = min(i, j), max(i, j)
    m = np.concatenate([m0, add]) if len(add) else m0.copy()
    mp = capi.partition(m, n, robots)
    kw = dict(method=1, acceleration=1, rgd_stepsize=float(rng.choice([0.05, 0.1, 0.2])), rgd_use_preconditioner=1,
              restart_interval=int(rng.integers(3, 40)))
    robust = bool(rng.integers(0, 2))
    if robust:
        kw.update(robust_cost_type=5, gnc_barc=5.0)
    T, Y = capi.odometry_init(m0, n), capi.fixed_stiefel(r)
    teams = []
    os.environ["DPGO_FE_MIN_N"] = "32"
    for fe, persist in ((0, 0), (1, 0), (1, 1)):
        os.environ["DPGO_FUSED_EVAL"] = str(fe)
        os.environ["DPGO_FE_PERSIST"] = str(persist)
        t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=robots, **kw), device=0)
        t.set_initial(T, Y)
        teams.append(t)
    for chunk in rng.integers(1, 400, 4):
        for t in teams:
            t.run(int(chunk)); t.synchronize()
        for u in (1, 2):
            d = max(float(np.max(np.abs(teams[0].agents[k].get_X() - teams[u].agents[k].get_X()))) for k in teams[0].ids)
            assert d == 0.0, (case, u, r, n_in, n_x, kw, int(chunk), d)
        if robust:
            assert teams[0].update_weights() == teams[1].update_weights() == teams[2].update_weights()
    c = teams[1].counters()
    deep_taken += c[9] > 0
    print("case %2d: r=%d +%3d inside +%2d across, restart %2d step %.2f robust %d -> one-launch %d, deep-carried %d (persistent form: %d), bitwise equal"
          % (case, r, n_in, n_x, kw["restart_interval"], kw["rgd_stepsize"], robust, c[7], c[9], teams[2].counters()[9]), flush=True)
    for t in teams:
        t.close()
print("fuzz ok: %d cases, deep-carried form taken in %d" % (cases, deep_taken))
