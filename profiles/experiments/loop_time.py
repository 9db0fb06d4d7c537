"""Message path in loopback (bench.py loopback_leg): is the host or the GPU the bound?  Prints, per iteration, the host time
of dpgo_team_run_ranks (enqueue only) and the time until the stream has drained; with LOOP_TRACE=1 only runs (for rocprofv3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch's first)
from dpgo_ros_amd import capi
import bench

m, n = capi.read_g2o(os.path.join(bench.ROOT, "data", "sphere2500.g2o"))
NA, r = 5, 5
mp = capi.partition(m, n, NA)
T, Y = capi.odometry_init(m, n), capi.fixed_stiefel(r)
comm = capi.Comm(capi.comm_unique_id(), 0, 1, device=0)
for loop in (True, False):
    t = capi.Team.from_measurements(mp, capi.default_params(r=r, num_robots=NA, **bench.RGD), device=0)
    t.set_initial(T, Y)
    if loop:
        t.attach_comm(comm, [0] * NA, loopback=True)
        t.exchange_all_ranks()
    sel = lambda k0, k: [(k0 + q) % NA for q in range(k)]
    K = int(os.environ.get("LOOP_K", "400"))
    if loop:
        t.run_ranks(sel(0, 50))
    else:
        t.run(50)
    t.synchronize()
    for rep in range(3):
        a0 = time.perf_counter()
        if loop:
            t.run_ranks(sel(50 + rep * K, K))
        else:
            t.run(K)
        a1 = time.perf_counter()
        t.synchronize()
        a2 = time.perf_counter()
        print("%s: host enqueue %.4f ms / iteration, drained %.4f ms / iteration" % ("loopback" if loop else "device-resident", (a1 - a0) / K * 1e3, (a2 - a0) / K * 1e3), flush=True)
    t.close()
