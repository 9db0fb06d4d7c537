import os, sys
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import numpy as np
from dpgo_ros_amd import capi
for ds, N, r in (("cubicle", 10, 5), ("cubicle", 10, 4), ("cubicle", 9, 3)):
    m, n = capi.read_g2o("data/%s.g2o" % ds)
    mp = capi.partition(m, n, N); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(r)
    res = []
    for fused in ("1", "0"):
        os.environ["DPGO_FUSED_RTR"] = fused
        t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=N, method=0, acceleration=0, rtr_iterations=3, rtr_tcg_iterations=12,
                                                                gradnorm_tol=1e-2, precond_mode=1))
        t.set_initial(T, Y)
        t.run(2 * N); t.synchronize()
        X = np.concatenate([t.agents[k].get_X() for k in t.ids])
        cnt = [(t.agents[k].opt_result().rtr_outer_iters, t.agents[k].opt_result().tcg_iters_total, t.agents[k].opt_result().accepted) for k in t.ids]
        sizes = sorted(set(len(t.agents[k].get_X()) // (4 * r) for k in t.ids))
        res.append((X, cnt))
        t.close()
    print(ds, N, r, "agent sizes", sizes, "max |X one-launch - X per-step| %.2e" % np.abs(res[0][0] - res[1][0]).max(), "counts equal", res[0][1] == res[1][1])
