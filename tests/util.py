"""Shared helpers of the parity tests: one set of seeded inputs, fed to the HIP path (through the
C-ABI) and to the CPU oracle."""
import os

import numpy as np

from dpgo_ros_amd import capi
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


def load(dataset, num_robots, weight_mode=0):
    m, n = O.read_g2o(os.path.join(DATA, dataset + ".g2o"), weight_mode)
    mp = O.partition(m, n, num_robots, weight_mode) if num_robots > 1 else m
    return m, mp, n


def params_pair(**kw):
    return capi.default_params(**kw), O.default_params(**kw)


def random_point(rng, r, n):
    X = rng.standard_normal(r * 4 * n)
    return O.project_manifold(X, r, n)


def make_pair(dataset, num_robots, r=5, init="odom", **kw):
    """(hip Team, oracle Team, n) on identical inputs and initial guess."""
    m, mp, n = load(dataset, num_robots)
    ph, po = params_pair(r=r, num_robots=num_robots, **kw)
    T = O.odometry_init(m, n) if init == "odom" else O.chordal_init(m, n)
    Y = O.fixed_stiefel(r)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    return th, to, n


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())
