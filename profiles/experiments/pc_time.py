"""the preconditioner-type kernels on the bench configuration: bare apply (k_precond<PM_PLAIN>), the two-launch iteration
(DPGO_FUSED_EVAL=0) and the one-launch iteration, ms per iteration / us per launch"""
import os, sys, time
sys.path.insert(0, ".")
import torch
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
for fe in ("0", "1"):
    os.environ["DPGO_FUSED_EVAL"] = fe
    prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
    t = capi.Team.from_measurements(mp, prm, device=0); t.set_initial(T, Y); t.run(100); t.prepare(4000); t.synchronize()
    best = 1e9
    for rep in range(4):
        a0 = time.perf_counter(); t.run(4000); t.synchronize(); best = min(best, (time.perf_counter() - a0) / 4000 * 1e3)
    print("DPGO_FUSED_EVAL=%s ms/iter %.5f" % (fe, best))
    if fe == "0":
        for which, name in ((0, "k_precond<PM_PLAIN> bare apply"), (9, "k_precond<PM_RGD> back to back"), (1, "k_eval")):
            ms, b = t.time_kernel(1, which, reps=500)
            print("  %s: %.2f us, %.0f GB/s" % (name, ms * 1e3, b / ms / 1e6))
    t.close()
