"""Phase timeline of the one-launch RTR solve with the two-level preconditioner (torus3D / 8, 625-pose agents): build with
profiles/experiments/build_variant.sh rtrtrace "-DDPGO_RTR_TRACE", run with DPGO_HIP_LIB=.../rtrtrace/libdpgo_hip.so.
Stamps of workgroup 0 after the slab load and after every grid hand-off OUTSIDE the product of the LAST solve of agent 0."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from dpgo_ros_amd import capi
ds, N = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("torus3D", 8)
mode = int(os.environ.get("PRECOND_MODE", "0"))
m, n = capi.read_g2o('data/%s.g2o' % ds)
mp = capi.partition(m, n, N); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, method=0, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=1e-2, restart_interval=50, precond_mode=mode))
t.set_initial(T, Y)
print(t.agents[0].preconditioner_info())
t.run(N * 4 + 1); t.synchronize()
out = (C.c_ulonglong * 448)()
capi.lib().dpgo_agent_read_rtr_handoff(t.h, 0, out, 448)
st = np.array(out[17 * 16 + 2:17 * 16 + 62], dtype=np.int64)
st = st[st > 0]
print("result", t.agents[0].opt_result().tcg_iters_total, "tCG iterations,", t.agents[0].opt_result().rtr_outer_iters, "outer")
print("us since stamp 0:", np.round((st - st[0]) / 100.0, 2).tolist())
print("deltas:", np.round(np.diff(st) / 100.0, 2).tolist())
a0 = time.perf_counter(); t.run(200); t.synchronize(); print("ms per iteration", (time.perf_counter() - a0) / 200 * 1e3)
