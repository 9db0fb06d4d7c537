# wall time of the GPU chordal initialisation (two dense SPD solves: 7500^2 rotations, 2500^2 translations)
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT)
from dpgo_ros_amd import capi
m, n = capi.read_g2o(os.path.join(ROOT, 'data/sphere2500.g2o'))
for rep in range(2):
    t0 = time.perf_counter(); T = capi.chordal_init(m, n); print("chordal_init sphere2500: %.3f s" % (time.perf_counter() - t0))
