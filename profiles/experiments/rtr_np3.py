"""RTR + Nesterov on torus3D / 8 (625-pose agents): the dense one-launch solve with three poses per workgroup (precond_mode 1)
against the two-level one (precond_mode 3); iterates of both against each other, ms per iteration"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from dpgo_ros_amd import capi
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
ds, N = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("torus3D", 8)
m, n = capi.read_g2o(os.path.join(ROOT, "data", ds + ".g2o"))
mp = capi.partition(m, n, N); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
out = {}
for mode in (1, 3):
    t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, method=0, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50,
                                                            gradnorm_tol=1e-2, restart_interval=50, precond_mode=mode))
    t.set_initial(T, Y)
    t.run(4 * N); t.synchronize()
    X = np.concatenate([t.agents[k].get_X() for k in t.ids])
    r = t.agents[t.ids[0]].opt_result()
    t0 = time.perf_counter(); t.run(200); t.synchronize(); ms = (time.perf_counter() - t0) / 200 * 1e3
    out[mode] = (X, ms, t.cost(), r.tcg_iters_total, r.rtr_outer_iters)
    print("precond_mode %d: %.4f ms per iteration, cost %.12g, last solve of agent 0: %d tCG / %d outer" % (mode, ms, out[mode][2], r.tcg_iters_total, r.rtr_outer_iters))
    t.close()
print("max |X_dense - X_two_level| after %d iterations: %.3e" % (4 * N, np.abs(out[1][0] - out[3][0]).max()))
