"""Are the slow trailing updates of an UPDATE_WEIGHT round's first factorisation chain (steps 2 and 3: 107 us among ones of
20-38, profiles/r06_update_weight_timeline.txt) a matter of the GPU waking up behind 1.4 ms of host work?  The same round with
a spinning kernel on a side stream while the host works (WARM=1) and without."""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/torus3D.g2o'))
T = capi.odometry_init(m, n); Y = capi.fixed_stiefel(5)
N = 8
mp = capi.partition(m, n, N)
t = capi.Team.from_measurements(mp, capi.default_params(r=5, num_robots=N, method=0, robust_cost_type=5, gnc_barc=5.0, gradnorm_tol=1e-2))
t.set_initial(T, Y); t.run(2 * N); t.synchronize()
side = torch.cuda.Stream()
warm = os.environ.get("WARM") == "1"
ts = []
for k in range(6):
    if warm:
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(4e6))  # ~2 ms of one spinning wave
    t0 = time.perf_counter(); t.update_weights(); t.synchronize(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t.run(N); t.synchronize()
print("WARM=%d update_weights ms:" % warm, ["%.2f" % (x * 1e3) for x in ts])
