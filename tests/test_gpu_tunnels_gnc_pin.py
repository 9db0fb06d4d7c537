"""HIP half of tests/test_tunnels_gnc_pin.py: the configuration of `launch/dpgo_gnc_demo.launch:27-43` on the tunnels
measurements through `dpgo_team_run_schedule` (the leader's UPDATE_WEIGHT decisions on the device side of the C-ABI),
the non-fixed weights after every UPDATE_WEIGHT round compared with column 15 of the reference's own CSV files -- the
only solver output the reference tree holds -- and with the oracle run beside it."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from oracle import tunnels_gnc_pin as P
from tests.test_tunnels_gnc_pin import check_against_the_file

pytestmark = pytest.mark.gpu


def hip_team(m, nk, T, r=5, **over):
    kw = dict(P.DEMO)
    kw.update(over)
    t = capi.Team.from_measurements(m.view(capi.MEAS_DTYPE), capi.default_params(r=r, num_robots=P.NUM_ROBOTS, **kw))
    t.set_initial(T, capi.fixed_stiefel(r))
    return t


def test_loader_reads_the_file_like_the_oracle():
    a = P.load()
    b = P.load(reader=capi.read_csv, dtype=capi.MEAS_DTYPE)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_hip_path_reproduces_the_reference_weights():
    m, wfile, inl, nk = P.load()
    T = P.aligned_odometry_guess(m, nk)
    th = hip_team(m, nk, T, robust_opt_num_weight_updates=5)
    to = P.oracle_team(m, nk, T, robust_opt_num_weight_updates=5)
    rows_h = P.run_rounds(th, m, wfile, rounds=5)
    rows_o = P.run_rounds(to, m, wfile, rounds=5)
    check_against_the_file(m, wfile, inl, rows_h)
    for rnd, (h, o) in enumerate(zip(rows_h, rows_o)):
        # 400 RTR block updates per round between the comparisons: the weights of the two paths agree far inside the
        # 1 % at which either agrees with the file
        assert np.array_equal(h["fixed"], o["fixed"])
        assert np.abs(h["weights"] - o["weights"]).max() < 1e-7, rnd
        assert (h["weights"] == 0).sum() == (o["weights"] == 0).sum()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-6
    th.close()
