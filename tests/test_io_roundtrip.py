"""Writers of SURVEY 8f-2..4: measurement CSV / g2o / trajectory CSV round trips through the loaders, the
iteration log's column order, and the CMake package the reference's find_package(DPGO) resolves to.
Host-only entry points of libdpgo_hip.so: no GPU needed."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from dpgo_ros_amd import capi
from tests.util import DATA, ROOT as REPO

FIELDS = ["r1", "p1", "r2", "p2", "kappa", "tau", "weight", "fixed_weight", "is_known_inlier"]


def test_measurement_csv_roundtrip(tmp_path):
    m = capi.read_csv(os.path.join(DATA, "tunnels", "robot0", "measurements.csv"))
    assert len(m) > 100 and (m["weight"] < 1).any()
    p = tmp_path / "m.csv"
    capi.write_csv(p, m)
    with open(p) as f:
        assert f.readline().strip() == ("robot_src,pose_src,robot_dst,pose_dst,qx,qy,qz,qw,tx,ty,tz,kappa,tau,"
                                        "is_known_inlier,weight")
    m2 = capi.read_csv(str(p))
    assert len(m2) == len(m)
    for k in FIELDS:
        assert np.array_equal(m[k], m2[k]), k            # integers and %.17g doubles: bit-exact
    assert np.array_equal(m["t"], m2["t"])
    assert np.abs(m["R"] - m2["R"]).max() < 5e-16 * 4    # R -> quaternion -> R
    # a second pass is a fixed point up to the same rounding
    capi.write_csv(tmp_path / "m2.csv", m2)
    m3 = capi.read_csv(str(tmp_path / "m2.csv"))
    assert np.abs(m3["R"] - m2["R"]).max() < 5e-16 * 4


def test_g2o_roundtrip(tmp_path):
    m, n = capi.read_g2o(os.path.join(DATA, "smallGrid3D.g2o"))
    T = capi.odometry_init(m, n)
    p = tmp_path / "g.g2o"
    capi.write_g2o(p, m, T=T, num_poses=n)
    m2, n2 = capi.read_g2o(str(p))
    assert n2 == n and len(m2) == len(m)
    assert np.array_equal(m["p1"], m2["p1"]) and np.array_equal(m["p2"], m2["p2"])
    assert np.array_equal(m["t"], m2["t"])
    assert np.abs(m["R"] - m2["R"]).max() < 2e-15
    # isotropic information blocks reproduce kappa / tau through the reader's trace formula
    assert np.abs(m["kappa"] - m2["kappa"]).max() <= 1e-12 * m["kappa"].max()
    assert np.abs(m["tau"] - m2["tau"]).max() <= 1e-12 * m["tau"].max()
    with open(p) as f:
        assert sum(1 for line in f if line.startswith("VERTEX_SE3:QUAT")) == n
    # partitioned list written back with robot offsets gives the same global graph
    mp = capi.partition(m, n, 2)
    per = n // 2
    capi.write_g2o(tmp_path / "gp.g2o", mp, robot_offsets=[0, per])
    m3, n3 = capi.read_g2o(str(tmp_path / "gp.g2o"))
    assert n3 == n and np.array_equal(m3["p1"], m["p1"]) and np.array_equal(m3["p2"], m["p2"])


def test_trajectory_csv(tmp_path):
    m, n = capi.read_g2o(os.path.join(DATA, "tinyGrid3D.g2o"))
    T = capi.odometry_init(m, n)
    p = tmp_path / "traj.csv"
    capi.write_trajectory_csv(p, T, n)
    rows = np.loadtxt(p, delimiter=",", skiprows=1)
    assert rows.shape == (n, 8)
    Tm = T.reshape(n, 4, 3)                              # column-major 3x4 per pose
    assert np.array_equal(rows[:, 5:8], Tm[:, 3, :])
    q = rows[:, 1:5]
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-15
    x, y, z, w = q.T
    R00 = 1 - 2 * (y * y + z * z)
    assert np.abs(R00 - Tm[:, 0, 0]).max() < 1e-14


def test_iteration_log_columns(tmp_path):
    p = tmp_path / "log.csv"
    log = capi.IterationLog(p)
    log.log(1, 0, 5, 7, 500, 16000, 4.1e-5, 0.5, 0.125, 843.5)
    log.log_string("TERMINATE")
    log.close()
    lines = open(p).read().split("\n")
    # the reference's header verbatim (src/PGOAgentROS.cpp:863-864) with one appended column
    assert lines[0].startswith("robot_id, cluster_id, num_active_robots, iteration, num_poses, bytes_received, "
                               "iter_time_sec, total_time_sec, rel_change")
    assert lines[0].rstrip().endswith("global_cost")
    assert lines[1].split(",")[:6] == ["1", "0", "5", "7", "500", "16000"]
    assert float(lines[1].split(",")[8]) == 0.125 and lines[2] == "TERMINATE"


@pytest.mark.skipif(shutil.which("cmake") is None, reason="cmake not on PATH")
def test_cmake_package_resolves(tmp_path):
    """find_package(DPGO REQUIRED) + target_link_libraries(... DPGO), as in the reference's CMakeLists.txt:6,151-154"""
    src = os.path.join(REPO, "tests", "cpp", "cmake_consumer")
    r = subprocess.run(["cmake", "-S", src, "-B", str(tmp_path), "-DDPGO_DIR=" + os.path.join(REPO, "cmake")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(tmp_path / "mock_wrapper")


# ---- robust inter-robot frame alignment: product (host entry of the C-ABI) vs oracle restatement
def _rand_rot(rng, angle=None):
    v = rng.normal(size=3)
    v /= np.linalg.norm(v)
    th = rng.uniform(0, np.pi) if angle is None else angle
    K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _candidates(seed, n, outlier_frac, rot_noise=0.02, trans_noise=0.05):
    rng = np.random.default_rng(seed)
    R0, t0 = _rand_rot(rng), rng.uniform(-20, 20, size=3)
    Tc, truth = np.zeros((n, 12)), np.ones(n, dtype=bool)
    for i in range(n):
        if rng.uniform() < outlier_frac:
            R, t, truth[i] = _rand_rot(rng, rng.uniform(1.2, np.pi)), rng.uniform(-20, 20, size=3), False
        else:
            R, t = R0 @ _rand_rot(rng, rot_noise * rng.uniform()), t0 + trans_noise * rng.normal(size=3)
        Tc[i, :9] = R.T.reshape(-1)      # column-major
        Tc[i, 9:] = t
    return Tc, truth, R0, t0


@pytest.mark.parametrize("seed,n,frac", [(0, 40, 0.3), (1, 12, 0.5), (2, 100, 0.6), (3, 5, 0.0), (4, 1, 0.0)])
def test_robust_frame_alignment_matches_oracle(seed, n, frac):
    from oracle import oracle
    Tc, truth, R0, t0 = _candidates(seed, n, frac)
    got = capi.robust_frame_alignment(Tc, min_inliers=1)
    ref = oracle.robust_frame_alignment(Tc, min_inliers=1)
    assert got is not None and ref is not None
    assert np.array_equal(got[1], ref[1])                       # inlier sets: exact
    assert np.abs(got[0] - ref[0]).max() < 1e-12                # tolerance: fp64, two independent 3x3 eigen-solvers
    assert np.array_equal(got[1], truth)                        # and they are the planted inliers
    R = got[0][:9].reshape(3, 3).T
    assert np.abs(R.T @ R - np.eye(3)).max() < 1e-13 and np.linalg.det(R) > 0
    assert np.linalg.norm(R - R0) < 0.05 and np.linalg.norm(got[0][9:] - t0) < 0.2


def test_robust_frame_alignment_min_inliers():
    from oracle import oracle
    Tc, truth, _, _ = _candidates(7, 6, 0.0)
    rng = np.random.default_rng(8)
    for i in range(6):   # six mutually inconsistent candidates: no consensus of 3
        Tc[i, :9] = _rand_rot(rng).T.reshape(-1)
        Tc[i, 9:] = rng.uniform(-50, 50, size=3)
    assert capi.robust_frame_alignment(Tc, min_inliers=3) is None
    assert oracle.robust_frame_alignment(Tc, min_inliers=3) is None
    assert capi.robust_frame_alignment(np.zeros((0, 12))) is None


def test_out_of_range_pose_indices_are_rejected():
    """host-side validation: a measurement that names a pose outside [0, n) must fail cleanly, not crash"""
    m, n = capi.read_g2o(os.path.join(DATA, "tinyGrid3D.g2o"))
    T = capi.odometry_init(m, n - 3)          # edges beyond the range are skipped
    assert np.isfinite(T).all() and T.shape == (12 * (n - 3),)
    with pytest.raises(capi.DpgoError):
        capi.chordal_init(m, n - 3)           # validated before any device work
