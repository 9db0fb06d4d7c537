"""per-phase timestamps of the one-launch iteration k_step_fe (build with -DDPGO_FE_TRACE: build_variant.sh fe "-DDPGO_FE_TRACE",
run with DPGO_HIP_LIB=profiles/experiments/build/fe/libdpgo_hip.so)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
team = capi.Team.from_measurements(mp, prm, device=0)
team.set_initial(T, Y)
PART_E = 4 * 32768 * 8
sname = ["start", "X staged|edges staged", "G done", "tangent done", "gradient in LDS", "product done", "reduced", "qf done", "V polar", "Y polar", "at A", "past A", "slab issued", "trip1 issued", "rows of the next agent left", "end"]
gname = sname
for rep in range(3):
    team.run(56)   # iterations 0 .. 49 are one-launch iterations; the last of them (49) belongs to agent 4
    team.synchronize()
    buf = np.zeros(128)
    capi.lib().dpgo_agent_read_partials(team.h, 4, PART_E + 4000 * 8, capi._d(buf), 128)
    t0 = buf[0]
    for w in (0, 1, 3, 4, 7):
        t = buf[16 * w:16 * w + 16]
        names = sname if w < 4 else gname
        print("wave %d " % w + " ".join("%s=%.2f" % (names[k], (t[k] - t0) / 100.0) for k in range(16) if t[k] and names[k]))
allb = np.zeros(8 * 2 * 256)
capi.lib().dpgo_agent_read_partials(team.h, 4, PART_E + 4100 * 8, capi._d(allb), allb.size)
allb = allb.reshape(256, 2, 8)[:250]
allb = allb[allb[:, 0, 0] > 0]
t0 = allb[:, 0, 0].min()
st = (allb[:, 0, 0] - t0) / 100.0
e0 = (allb[:, 0, 1] - t0) / 100.0
print("workgroups %d start: min %.2f max %.2f | wave-0 end: min %.2f median %.2f max %.2f" % (len(st), st.min(), st.max(), e0.min(), np.median(e0), e0.max()))
order = np.argsort(e0)
print("latest wave-0 ends:", " ".join("%d:%.2f(start %.2f)" % (i, e0[i], st[i]) for i in order[-8:]))
print("earliest:", " ".join("%d:%.2f" % (i, e0[i]) for i in order[:8]))
ms, b = team.time_kernel(1, 14, reps=500)
print("k_step_fe in this build: %.2f us per launch (HIP events)" % (ms * 1e3))
import ctypes
