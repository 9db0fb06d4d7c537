"""per-phase timestamps of the two-launch step kernel k_precond<PM_RGD> (build csrc with -DDPGO_PC_TRACE first)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
team = capi.Team.from_measurements(mp, prm, device=0)
team.set_initial(T, Y)
names = ["start", "prologue done", "vector staged", "M requested", "M consumed", "reduced", "sync", "end", "tangent+step", "qf", "V polar", "Y polar"]
if team.agents[0].preconditioner() == capi.PRECOND_TWO_LEVEL:  # stamps of twolevel_dev.h
    names[1:6] = ["pre rows done", "u published (producers)", "exchange passed", "u rows done", "reduced"]
    print(team.agents[0].preconditioner_info())
PART_E = 4 * 32768 * 8
for rep in range(4):
    team.run(36)   # the last step kernel of the run belongs to agent (35 % 5) = 0
    buf = np.zeros(64)
    capi.lib().dpgo_agent_read_partials(team.h, 0, PART_E + 4000 * 8, capi._d(buf), 64)
    for w in (0, 1, 2):
        t = buf[16 * w:16 * w + 12]
        print("wave %d " % w + " ".join("%s=%.2f" % (names[k], (t[k] - buf[0]) / 100.0) for k in (0,1,2,3,4,5,6,8,9,10,11,7) if t[k]))

# start / end of wave 0 and wave 1 of EVERY workgroup of the last step kernel (hardware block order)
allb = np.zeros(8 * 2 * 256)
capi.lib().dpgo_agent_read_partials(team.h, 0, PART_E + 4100 * 8, capi._d(allb), allb.size)
allb = allb.reshape(256, 2, 8)[:250]
t0 = allb[:, 0, 0].min()
st = (allb[:, 0, 0] - t0) / 100.0
e0 = (allb[:, 0, 1] - t0) / 100.0
e1 = (allb[:, 1, 1] - t0) / 100.0
print("workgroup start: min %.2f max %.2f | wave-0 end: min %.2f median %.2f max %.2f | wave-1 end: median %.2f max %.2f"
      % (st.min(), st.max(), e0.min(), np.median(e0), e0.max(), np.median(e1), e1.max()))
print("latest wave-0 ends at blocks", np.argsort(e0)[-6:].tolist(), np.sort(e0)[-6:].round(2).tolist())
