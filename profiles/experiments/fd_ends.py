"""when every wave of one workgroup of k_step_fd leaves, in a build that stores nothing else (build_variant.sh fdends "-DDPGO_FD_ENDS")"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch  # noqa: F401
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
team = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=5, **bench.RGD), device=0)
team.set_initial(T, Y)
PART_E = 4 * 32768 * 8
for rep in range(3):
    team.run(58); team.synchronize()
    buf = np.zeros(128)
    capi.lib().dpgo_agent_read_partials(team.h, 1, PART_E + 4000 * 8, capi._d(buf), 128)
    t0 = buf[64]
    print("wave ends (us from wave 4's first instruction): " + "  ".join("%d: %.2f" % (w, (buf[16 * w + 15] - t0) / 100.0) for w in range(8)))
ms, b = team.time_kernel(1, 14, reps=500)
print("k_step_fd in this build: %.2f us per launch" % (ms * 1e3))
