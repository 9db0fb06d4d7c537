"""chordal initialisation of sphere2500 alone (for rocprofv3 --kernel-trace --stats): where its 33 ms go"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from dpgo_ros_amd import capi
m, mp, n, T, Y = bench.load_problem(capi)
capi.chordal_init(m, n)
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    Tc = capi.chordal_init(m, n)
    ts.append((time.perf_counter() - t0) * 1e3)
print("chordal_init ms:", ["%.2f" % x for x in ts])
