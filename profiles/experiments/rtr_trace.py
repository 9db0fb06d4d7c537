"""Phase timeline of the one-launch RTR solve (build libdpgo_hip.so with -DDPGO_RTR_TRACE): wall-clock stamps (100 MHz)
of workgroup 0 after the slab load and after every grid hand-off of the LAST solve of agent 0."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from dpgo_ros_amd import capi
m, n = capi.read_g2o('data/sphere2500.g2o')
mp = capi.partition(m, n, 5); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, method=0, acceleration=1, rtr_iterations=3, rtr_tcg_iterations=50, gradnorm_tol=1e-2, restart_interval=50))
t.set_initial(T, Y)
t.run(int(sys.argv[1]) if len(sys.argv) > 1 else 101); t.synchronize()
out = (C.c_ulonglong * 448)()
capi.lib().dpgo_agent_read_rtr_handoff(t.h, 0, out, 448)
st = np.array(out[17 * 16 + 2:17 * 16 + 62], dtype=np.int64)
st = st[st > 0]
o = t.agents[0].opt_result() if hasattr(t.agents[0], "opt_result") else None
print("stamps", len(st), "result", o)
print("us since stamp 0:", np.round((st - st[0]) / 100.0, 2).tolist())
print("deltas:", np.round(np.diff(st) / 100.0, 2).tolist())
fine = np.array(out[17 * 16 + 2 + 64:17 * 16 + 2 + 64 + 18], dtype=np.int64)
print("fine (tCG iteration 1 of outer 0; us since its start): 0 start, 1 sums, 2 gathers, 3 hess tail, 4 stores, 5 drained, 6 wg barrier, 7 arrived, 8 released, 9 out,")
print("   10 sum, 11 slab apply, 12 tail+stores, 13 drained, 14 wg barrier, 15 arrived, 16 released, 17 out")
print(np.round((fine - fine[0]) / 100.0, 2).tolist())
sl = np.array(out[17 * 16 + 2 + 64 + 20:17 * 16 + 2 + 64 + 23], dtype=np.int64)
print("inside the slab product (us since the iteration's start): vector arrived, products done, quad sums in LDS + barrier:", np.round((sl - fine[0]) / 100.0, 2).tolist())
sc = np.array(out[17 * 16 + 2 + 64 + 23:17 * 16 + 2 + 64 + 26], dtype=np.int64)
print("the same three points by the shader clock (s_memtime): products %d clocks, reduction %d clocks -> %.2f GHz" % (
    sc[1] - sc[0], sc[2] - sc[1], (sc[2] - sc[0]) / max(1, (sl[2] - sl[0])) / 10.0))
