"""per-phase timestamps of the deep-carried one-launch iteration k_step_fd (build with -DDPGO_FE_TRACE:
build_variant.sh fd "-DDPGO_FE_TRACE", run with DPGO_HIP_LIB=profiles/experiments/build/fd/libdpgo_hip.so)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
team = capi.Team.from_measurements(mp, prm, device=0)
team.set_initial(T, Y)
PART_E = 4 * 32768 * 8
names = {
    "s": {0: "start", 10: "at A", 11: "past A", 1: "C signalled", 12: "bulk issued", 13: "next vector in LDS (N)", 15: "end (partial sums stored)"},
    "c": {0: "start", 14: "W / X requested", 10: "at A", 11: "past A", 1: "edges in LDS (E)", 3: "G_j + projection", 2: "past C", 12: "rows written", 13: "D signalled", 4: "past D", 5: "product done", 6: "reduced", 7: "qf", 8: "V polar", 9: "Y polar", 15: "end"},
    "w": {0: "start", 9: "edges requested", 10: "at A", 11: "past A", 8: "E signalled", 14: "end"},
    "l": {0: "start", 9: "edges requested", 10: "at A", 11: "past A", 8: "E signalled", 7: "first map", 15: "end"},
}
AG = 1  # the agent of rep 46 of a 56-iteration graph (50 one-launch iterations): every producer flag set
for rep in range(3):
    team.run(56)
    team.synchronize()
    buf = np.zeros(128)
    capi.lib().dpgo_agent_read_partials(team.h, AG, PART_E + 4000 * 8, capi._d(buf), 128)
    t0 = buf[4 * 16]
    for w, kind in ((0, "s"), (3, "s"), (4, "c"), (5, "c"), (6, "w"), (7, "l")):
        t = buf[16 * w:16 * w + 16]
        nm = names[kind]
        print("wave %d " % w + " | ".join("%s %.2f" % (nm[k], (t[k] - t0) / 100.0) for k in sorted(nm, key=lambda k: t[k]) if t[k]))
    print()
allb = np.zeros(8 * 2 * 256)
capi.lib().dpgo_agent_read_partials(team.h, AG, PART_E + 4100 * 8, capi._d(allb), allb.size)
allb = allb.reshape(256, 2, 8)[:250]
allb = allb[allb[:, 0, 0] > 0]
t0 = allb[:, 0, 0].min()
st = (allb[:, 0, 0] - t0) / 100.0
e4 = (allb[:, 0, 1] - t0) / 100.0
e0 = (allb[:, 0, 2] - t0) / 100.0
print("workgroups %d start: min %.2f max %.2f | wave-4 (tail) end: min %.2f median %.2f max %.2f | wave-0 (streamer) end: min %.2f median %.2f max %.2f" % (len(st), st.min(), st.max(), e4.min(), np.median(e4), e4.max(), e0.min(), np.median(e0), e0.max()))
ms, b = team.time_kernel(1, 14, reps=500)
print("k_step_fd in this build: %.2f us per launch (HIP events)" % (ms * 1e3))
