"""chordal initialisation of sphere2500 (and torus3D): wall time of repeated calls in one process"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from dpgo_ros_amd import capi
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
for ds in ("sphere2500", "torus3D"):
    m, n = capi.read_g2o(os.path.join(ROOT, "data", ds + ".g2o"))
    capi.chordal_init(m, n)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter(); capi.chordal_init(m, n); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("chordal_init %s (%d poses): min %.2f median %.2f max %.2f ms" % (ds, n, ts[0], ts[len(ts) // 2], ts[-1]))
