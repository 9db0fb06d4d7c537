"""per-phase timestamps of one iteration of the persistent deep-carried kernel k_step_pd (build with -DDPGO_PD_TRACE:
build_variant.sh pd "-DDPGO_PD_TRACE", run with DPGO_HIP_LIB=profiles/experiments/build/pd/libdpgo_hip.so)"""
import os, sys, time
sys.path.insert(0, ".")
os.environ["DPGO_FE_PERSIST"] = "1"
import numpy as np
import torch  # noqa: F401
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
team = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, **bench.RGD), device=0)
team.set_initial(T, Y)
PART_E = 4 * 32768 * 8
names = {0: "iteration starts", 2: "C", 11: "past RQ (stream released)", 12: "past N+F", 13: "partial sums in LDS", 10: "requested", 1: "past E", 3: "past C", 4: "D signalled",
         5: "F signalled", 6: "past F", 7: "qf", 14: "role done", 15: "past the grid hand-off"}
for rep in range(2):
    team.run(256); team.synchronize()
    buf = np.zeros(128)
    capi.lib().dpgo_agent_read_partials(team.h, 0, PART_E + 4000 * 8, capi._d(buf), 128)
    t0 = buf[4 * 16]
    for w in (0, 3, 4, 5, 6, 7):
        t = buf[16 * w:16 * w + 16]
        print("wave %d " % w + " | ".join("%s %.2f" % (names[k], (t[k] - t0) / 100.0) for k in sorted(names, key=lambda k: t[k]) if t[k]))
    print()
team.prepare(4000); team.synchronize()
for rep in range(3):
    a0 = time.perf_counter(); team.run(4000); team.synchronize(); print("ms/iter %.5f" % ((time.perf_counter() - a0) / 4000 * 1e3))
