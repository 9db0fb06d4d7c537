"""The first replay of the K = 20 graph against later ones, split: host time inside dpgo_team_run (enqueue), GPU time between two
events on the team's stream, wall time to the end of the synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from dpgo_ros_amd import capi
import bench

m, mp, n, T, Y = bench.load_problem(capi)
prm = capi.default_params(r=5, num_robots=5, **bench.RGD)
for trial in range(3):
    team = capi.Team.from_measurements(mp, prm, device=0)
    team.set_initial(T, Y)
    team.run(5)
    team.prepare(20)
    team.synchronize()
    st = torch.cuda.ExternalStream(team.stream())
    rows = []
    for rep in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        team.run(20)
        t1 = time.perf_counter()
        e1.record(st)
        team.synchronize()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        rows.append(((t1 - t0) * 1e6, e0.elapsed_time(e1) * 1e3, (t2 - t0) * 1e6))
    print("team %d: " % trial + " | ".join("host %.0f gpu %.0f wall %.0f us" % r for r in rows), flush=True)
    team.close()
