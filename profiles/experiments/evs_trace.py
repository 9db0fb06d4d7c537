"""phase stamps of k_eval_staged in a lockstep ASAPP tick on tunnels (build with -DDPGO_EVS_TRACE): per workgroup of every
agent, us since the workgroup's start: [0] evaluation done (the tile's wave), [1..3] the three helper waves done staging;
[4] shared edges of the tile"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from dpgo_ros_amd import capi
m, nk, T = bench.load_tunnels(capi, 1)
t = capi.Team.from_measurements(m, capi.default_params(r=5, num_robots=8, method=1, rgd_stepsize=0.2, acceleration=0), device=0)
t.set_initial(T, capi.fixed_stiefel(5))
t.run_simultaneous(65); t.synchronize()
PART_E = 4 * 32768 * 8
for k in range(8):
    nb = (nk[k] + 11) // 12
    buf = np.zeros(8 * nb)
    capi.lib().dpgo_agent_read_partials(t.h, k, PART_E + 4000 * 8, capi._d(buf), buf.size)
    b = buf.reshape(nb, 8)
    print("agent %d (%d poses): per tile [done, helpers staged x 3] us, edges" % (k, nk[k]))
    for row in b:
        print("   %5.2f %5.2f %5.2f %5.2f   %3d" % (row[0] / 100, row[1] / 100, row[2] / 100, row[3] / 100, int(row[4])))
