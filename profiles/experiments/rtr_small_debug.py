"""debug: one-launch RTR solve on small agents (smallGrid3D / 3): did it run, what did the hand-off words and the error word say"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
from dpgo_ros_amd import capi
r = int(sys.argv[1]) if len(sys.argv) > 1 else 5
two = len(sys.argv) > 2
m, n = capi.read_g2o('data/smallGrid3D.g2o')
mp = capi.partition(m, n, 3); T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(r)
kw = dict(method=capi.METHOD_RTR, acceleration=1, restart_interval=5, gradnorm_tol=1e-3, rtr_iterations=3, rtr_tcg_iterations=30)
def team(fused):
    os.environ["DPGO_FUSED_RTR"] = "1" if fused else "0"
    return capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=3, **kw))
def show(t, tag):
    for a in range(3):
        out = (C.c_ulonglong * 320)()
        rc = capi.lib().dpgo_agent_read_rtr_handoff(t.h, a, out, 320)
        o = t.agents[a].opt_result()
        print(tag, "agent", a, "rc", rc, "epoch", out[17 * 16], "abort", out[17 * 16 + 1], "shard0", out[0], "outer", o.rtr_outer_iters, "tcg", o.tcg_iters_total,
              "acc", o.accepted, "err:", capi.lib().dpgo_last_error())
    print(tag, "counters", t.counters())
tf = team(True)
ts = team(False) if two else None
tf.set_initial(T, Y)
if ts: ts.set_initial(T, Y)
tf.run(12)
if ts: ts.run(12)
show(tf, "fused")
if ts:
    show(ts, "per-step")
    print("max |X_f - X_s|", np.abs(tf.global_X() - ts.global_X()).max())
