"""The two-level (nested-dissection / Schur-complement) form of the preconditioner (csrc/twolevel.h): the same
operator (Q + shift I)^-1 as the dense inverse and as the oracle's sparse Cholesky solve, to
round-off -- apply, RGD and RTR iterates, weight updates."""
import numpy as np
import pytest

from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import make_pair, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dataset,robots", [("smallGrid3D", 2), ("sphere2500", 5), ("torus3D", 8), ("sphere2500", 1)])
def test_two_level_apply_matches_sparse_cholesky(dataset, robots):
    th, to, n = make_pair(dataset, robots, precond_mode=capi.PRECOND_TWO_LEVEL)
    rng = np.random.default_rng(3)
    for k in range(robots):
        ah, ao = th.agents[k], to.agents[k]
        ah.build_problem(False)
        ao.build_problem(False)
        info = ah.preconditioner_info()
        assert info["mode"] == capi.PRECOND_TWO_LEVEL
        X = ah.get_X()
        V = rng.standard_normal(X.shape)
        zh, zo = ah.precondition(X, V), ao.precondition(X, V)
        assert relerr(zh, zo) < 1e-9, (dataset, ah.id, info, relerr(zh, zo))


def test_automatic_mode_by_agent_size():
    """the dense inverse where its stream is cheaper than the exchange inside a two-level apply (measured: up to ~1250
    poses), the two-level form beyond"""
    th, _, _ = make_pair("sphere2500", 5)
    assert all(a.preconditioner() == capi.PRECOND_DENSE for a in th.agents.values())
    th.close()
    th, _, _ = make_pair("sphere2500", 1)
    info = th.agents[0].preconditioner_info()
    assert info["mode"] == capi.PRECOND_TWO_LEVEL
    assert info["bytes_per_apply"] < 0.25 * info["dense_bytes"] == 0.25 * 8e8
    th.close()


def test_two_level_bytes_on_the_bench_agents():
    """sphere2500 / 5: 11 subdomains + 71 separator poses, 9.5 MB per apply against the dense inverse's 32 MB"""
    th, _, _ = make_pair("sphere2500", 5, precond_mode=capi.PRECOND_TWO_LEVEL)
    for ah in th.agents.values():
        info = ah.preconditioner_info()
        assert info["mode"] == capi.PRECOND_TWO_LEVEL and info["bytes_per_apply"] < 1e7 and info["dense_bytes"] == 32e6
    th.close()


@pytest.mark.parametrize("method", [O.METHOD_RGD, O.METHOD_RTR])
@pytest.mark.parametrize("accel", [0, 1])
def test_two_level_iterates_match_the_oracle(method, accel):
    kw = dict(method=method, acceleration=accel, restart_interval=7, rgd_stepsize=0.1, gradnorm_tol=1e-2,
              precond_mode=capi.PRECOND_TWO_LEVEL)
    th, to, n = make_pair("sphere2500", 5, **kw)
    th.run(25)
    for _ in range(25):
        to.iterate()
    assert np.abs(th.global_X() - to.global_X()).max() < 1e-7
    assert abs(th.cost() - to.cost()) <= 1e-9 * abs(to.cost())
    th.close()


@pytest.mark.parametrize("mode", [capi.PRECOND_DENSE, capi.PRECOND_TWO_LEVEL])
def test_preconditioner_residual_of_both_exact_forms(mode):
    """|z (Q + 0.1 I) - v| / |v| of the operator the kernels apply, bench agents (cond ~ 1e5): round-off either way"""
    th, _, _ = make_pair("sphere2500", 5, precond_mode=mode)
    for ah in th.agents.values():
        assert ah.preconditioner() == mode
        assert ah.preconditioner_residual() < 1e-12
    th.close()
