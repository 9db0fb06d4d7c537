"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/dpgo_hip.h declares, its host-side loaders agree bit-for-bit with the oracle's, and the
compute path refuses to run (loudly) without a HIP device -- there is no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import DATA, ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "dpgo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dpgo_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _declared_symbols()
    assert len(declared) >= 55
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared  # the python mirror lists exactly the header's symbols


def test_struct_layouts_match():
    assert C.sizeof(capi.Measurement) == 144 == capi.MEAS_DTYPE.itemsize == O.MEAS_DTYPE.itemsize
    p, q = capi.default_params(r=5, num_robots=3), O.default_params(r=5, num_robots=3)
    assert bytes(p) == bytes(q)  # identical defaults, identical layout


@pytest.mark.parametrize("ds", ["tinyGrid3D", "smallGrid3D", "sphere2500", "torus3D"])
@pytest.mark.parametrize("mode", [capi.WEIGHT_LIBRARY, capi.WEIGHT_WRAPPER])
def test_g2o_loader_and_partition_bit_exact(ds, mode):
    path = os.path.join(DATA, ds + ".g2o")
    m, n = capi.read_g2o(path, mode)
    mo, no = O.read_g2o(path, mode)
    assert n == no and m.tobytes() == mo.tobytes()
    for N in (2, 5, 8):
        assert capi.partition(m, n, N, mode).tobytes() == O.partition(mo, no, N, mode).tobytes()
    assert capi.odometry_init(m, n).tobytes() == O.odometry_init(mo, no).tobytes()
    if mode == capi.WEIGHT_WRAPPER:  # src/utils.cpp:141-149
        assert set(m["kappa"]) == {10000.0} and set(m["tau"]) == {100.0}
        mp = capi.partition(m, n, 2, mode)
        odo = (mp["r1"] == mp["r2"]) & (mp["p1"] + 1 == mp["p2"])
        assert np.array_equal(mp["fixed_weight"].astype(bool), odo)


def test_partition_rule_matches_reference():
    """src/PGODatasetPublisherNode.cpp:84-103: per = n / N, last robot takes the remainder."""
    m, n = capi.read_g2o(os.path.join(DATA, "sphere2500.g2o"))
    mp = capi.partition(m, n, 8)
    per = n // 8
    assert per == 312
    g1 = m["p1"]
    assert np.array_equal(mp["r1"], np.minimum(g1 // per, 7))
    assert np.array_equal(mp["p1"], g1 - mp["r1"] * per)
    assert max(mp["p1"][mp["r1"] == 7].max(), mp["p2"][mp["r2"] == 7].max()) == 2500 - 7 * 312 - 1  # 316 poses on the last robot


@pytest.mark.parametrize("robot", range(8))
def test_csv_loader_bit_exact(robot):
    path = os.path.join(DATA, "tunnels", "robot%d" % robot, "measurements.csv")
    for mode in (0, 1):
        assert capi.read_csv(path, mode).tobytes() == O.read_csv(path, mode).tobytes()
    m = capi.read_csv(path, 0)
    assert len(m) > 100 and set(m["kappa"]) == {10000.0}
    assert np.array_equal(m["is_known_inlier"] == 1, (m["r1"] == m["r2"]) & (m["p1"] + 1 == m["p2"]))


def test_quantile_threshold():
    # chi2inv(0.9, 3) = 6.251388631170325
    assert abs(capi.error_threshold_at_quantile(0.9, 3) ** 2 - 6.251388631170325) < 1e-9


def test_compute_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.DpgoError, match="no HIP device"):
        capi.Team(capi.default_params(r=5, num_robots=1), [0])


def test_bad_rank_rejected():
    p = capi.default_params(r=2, num_robots=1)
    with pytest.raises(capi.DpgoError):
        capi.Team(p, [0])


def test_robust_cost_parameters_are_validated():
    """the six cost types of src/PGOAgentROSNode.cpp:178-188 have C-ABI values in the facade's enum order; their thresholds
    default to the library's ([UPSTREAM-RECALL] 10 / 3); anything else is refused before a device is looked for"""
    assert (capi.COST_L2, capi.COST_L1, capi.COST_HUBER, capi.COST_TLS, capi.COST_GM, capi.COST_GNC_TLS) == (0, 1, 2, 3, 4, 5)
    p = capi.default_params(r=5, num_robots=1)
    assert p.tls_threshold == 10.0 and p.huber_threshold == 3.0
    for bad in (dict(robust_cost_type=6), dict(robust_cost_type=-1), dict(robust_cost_type=capi.COST_TLS, tls_threshold=0.0),
                dict(robust_cost_type=capi.COST_HUBER, huber_threshold=-1.0)):
        with pytest.raises(capi.DpgoError) as e:
            capi.Team(capi.default_params(r=5, num_robots=1, **bad), [0])
        assert "robust_cost_type" in str(e.value) or "threshold" in str(e.value)


def test_rank_exchange_fails_loudly_without_a_device():
    """no CPU fallback in the multi-rank path either: without a HIP device a communicator cannot be created (the error says
    why), and the planning layer -- host arithmetic -- rejects a rank outside its world"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.DpgoError, match="no HIP device|RCCL"):
        capi.Comm(b"\0" * capi.COMM_ID_BYTES, 0, 1, device=0)
    with pytest.raises(capi.DpgoError, match="bad arguments"):
        capi.rank_plan_simulate([0, 1], [[0, 3], [3, 0]], 5, 2, [0, 1])
    lib_path, version = capi.comm_library()   # RCCL is bound at run time, also here
    assert version > 20000 and "rccl" in lib_path
