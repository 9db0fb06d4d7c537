"""The dissection behind the two-level preconditioner (csrc/twolevel_plan.cpp) on the reference's datasets, host
arithmetic only (dpgo_two_level_plan through the C-ABI; no GPU): subdomains touch only through the separator, every pose
is placed exactly once, the plan is deterministic, and one apply streams a fraction of the dense inverse's bytes."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load, merged_graph, synthetic_chain


def _pattern(mp, rid, n):
    adj = [set([i]) for i in range(n)]
    e = mp[(mp["r1"] == rid) & (mp["r2"] == rid)]
    for a, b in zip(e["p1"], e["p2"]):
        adj[int(a)].add(int(b))
        adj[int(b)].add(int(a))
    rp, col = [0], []
    for s in adj:
        col += sorted(s)
        rp.append(len(col))
    return np.array(rp, dtype=np.int32), np.array(col, dtype=np.int32)


def _check(rp, col, n, max_ratio):
    sub_of, info = capi.two_level_plan(rp, col)
    again, info2 = capi.two_level_plan(rp, col)
    assert np.array_equal(sub_of, again) and info == info2              # deterministic
    assert sub_of.shape == (n,) and sub_of.min() >= -1 and sub_of.max() == info["subdomains"] - 1
    assert (sub_of == -1).sum() == info["separator_poses"]
    for j in range(n):                                                  # no edge between two different subdomains
        for i in col[rp[j]:rp[j + 1]]:
            assert sub_of[i] == sub_of[j] or sub_of[i] < 0 or sub_of[j] < 0, (i, j)
    assert info["producer_workgroups"] == info["separator_poses"]
    dense = 128.0 * n * n
    assert info["bytes_per_apply"] <= max_ratio * dense, (info, dense)
    return info


@pytest.mark.parametrize("dataset,robots,max_ratio", [("sphere2500", 5, 0.32), ("torus3D", 8, 0.30), ("cubicle", 1, 0.25),
                                                      ("parking-garage", 1, 0.2), ("sphere2500", 1, 0.2)])
def test_plan_on_the_bundled_datasets(dataset, robots, max_ratio):
    m, mp, n = load(dataset, robots)
    for k in range(robots):
        sel = mp[(mp["r1"] == k) | (mp["r2"] == k)]
        nk = int(max(sel["p1"][sel["r1"] == k].max(initial=0), sel["p2"][sel["r2"] == k].max(initial=0))) + 1
        rp, col = _pattern(mp, k, nk)
        info = _check(rp, col, nk, max_ratio)
        if dataset == "sphere2500" and robots == 5:
            assert info["separator_poses"] <= 80 and info["bytes_per_apply"] < 1e7 and not info["worthwhile"]
        if robots == 1 and dataset != "parking-garage":
            assert info["worthwhile"]       # the automatic mode takes the two-level form beyond 256 MB of dense inverse


def test_plan_on_random_loop_closures_and_on_a_long_chain():
    mo, n = merged_graph()
    mp = O.partition(mo, n, 8)
    rp, col = _pattern(mp, 7, n - 7 * (n // 8))   # cubicle tail + garage + 10 % random loop closures: poor separators
    _check(rp, col, n - 7 * (n // 8), 0.5)
    m, n = synthetic_chain(60000)
    rp, col = _pattern(m, 0, n)
    info = _check(rp, col, n, 0.02)
    assert info["bytes_per_apply"] < 8e9


def test_plan_of_tiny_and_disconnected_graphs():
    # 3 poses, no edges at all: every pose its own component
    rp, col = np.array([0, 1, 2, 3], dtype=np.int32), np.array([0, 1, 2], dtype=np.int32)
    sub_of, info = capi.two_level_plan(rp, col)
    assert (sub_of >= 0).all() and info["separator_poses"] == 0
    # a single pose
    sub_of, info = capi.two_level_plan(np.array([0, 1], dtype=np.int32), np.array([0], dtype=np.int32))
    assert sub_of.tolist() == [0] and info["subdomains"] == 1
