"""The robust path against the ONE solver output the reference tree holds: the GNC-TLS weights in column 15 of
data/tunnels/robot*/measurements.csv (`data/tunnels/robot0/measurements.csv:1`), left behind by a real dpgo run of
`launch/dpgo_gnc_demo.launch`.  CPU half: the oracle (oracle/tunnels_gnc_pin.py has the script and the sweep); the HIP
half is tests/test_gpu_tunnels_gnc_pin.py."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import tunnels_gnc_pin as P


@pytest.fixture(scope="module")
def demo_rounds():
    m, wfile, inl, nk = P.load()
    T = P.aligned_odometry_guess(m, nk)
    t = P.oracle_team(m, nk, T, robust_opt_num_weight_updates=5)
    return m, wfile, inl, P.run_rounds(t, m, wfile, rounds=5)


def check_against_the_file(m, wfile, inl, rows):
    """the assertions both halves share (round-3 verdict, item 1)"""
    med = [r["median"] for r in rows]
    assert int(np.argmin(med)) == 2, med  # the file is the state after the THIRD update (num_weight_updates 3, launch:39)
    assert min(med[0], med[1], med[3], med[4]) > 10 * med[2]  # and the minimum is sharp
    r3 = rows[2]
    assert r3["median"] <= 0.03 and r3["p90"] <= 0.10 and r3["spearman"] >= 0.93, r3
    assert r3["zero_overlap"] >= 0.90 and abs(r3["zeros"] - r3["zeros_file"]) <= 6
    assert abs(r3["signed_median"]) < 0.005  # no scale error: mu = 4e-5 at the third update, barc = 3
    # the fixed set is the file's is_known_inlier set (odometry rows: src/utils.cpp:147-149)
    assert np.array_equal(r3["fixed"], inl)
    # without the robot 1 - robot 2 group (oracle/tunnels_gnc_pin.py: ANOMALOUS_PAIR) the agreement is much tighter and the
    # zero set is IDENTICAL
    assert r3["rest_median"] <= 0.015 and r3["rest_p90"] <= 0.06 and r3["rest_spearman"] >= 0.997, r3
    assert r3["rest_zeros"] == r3["rest_zeros_file"] == 58 and r3["rest_zero_overlap"] == 1.0
    pairs = P.per_pair(r3["weights"], wfile, m, ~r3["fixed"])
    for g, (cnt, pmed, p90, signed) in pairs.items():
        if g == P.ANOMALOUS_PAIR:
            assert pmed > 1.0  # robots 1-2: the file's residuals are several times anybody's
        elif cnt >= 20:
            assert pmed <= 0.025 and abs(signed) <= 0.01, (g, cnt, pmed, signed)


def test_file_layout_and_duplicates():
    m, wfile, inl, nk = P.load()
    assert len(m) == 4891 and int((~inl).sum()) == 3644 and nk == [105, 138, 149, 148, 168, 175, 191, 181]
    assert np.all(m["kappa"] == 1e4) and np.all(m["tau"] == 1e2) and np.all(m["weight"] == 1.0)
    odo = (m["r1"] == m["r2"]) & (m["p1"] + 1 == m["p2"])
    assert np.array_equal(odo, inl) and np.array_equal(m["fixed_weight"].astype(bool), inl)
    assert int((wfile[~inl] == 0).sum()) == 65 and np.all(wfile[inl] == 1.0)
    assert 0 < np.median(wfile[~inl]) < 1e-3  # GNC's middle branch throughout: w ~ barc sqrt(mu) / residual


def test_oracle_reproduces_the_reference_weights(demo_rounds):
    m, wfile, inl, rows = demo_rounds
    check_against_the_file(m, wfile, inl, rows)


def test_what_the_file_says_about_the_recalled_constants():
    """The sweep of DESIGN.md 0 in brief: the data separate the trust-region radius (100 >> 10), the mu schedule (the
    weights are formed with mu, THEN mu is stepped) and barc; they do not separate the preconditioner shift."""
    m, wfile, inl, nk = P.load()
    T = P.aligned_odometry_guess(m, nk)

    def third(**over):
        t = P.oracle_team(m, nk, T, **over)
        return P.run_rounds_manual(t, m, wfile, rounds=3)[2]

    base = third()
    r10 = third(rtr_initial_radius=10.0, rtr_max_radius=50.0)
    assert r10["rest_median"] > 3 * base["rest_median"]
    mu_first = third(gnc_init_mu=2e-5)
    assert mu_first["rest_median"] > 0.3 and mu_first["signed_median"] > 0.3  # sqrt(2) too large throughout
    barc5 = third(gnc_barc=5.0)
    assert barc5["rest_median"] > 0.5
    shift = third(precond_shift=1e-3)
    assert abs(shift["rest_median"] - base["rest_median"]) < 0.002
