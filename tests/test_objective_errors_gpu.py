"""Objective error paths on the device route, one optimizer per family (ActiveCMAES, SHADE, APSO).

The reference hands a Python exception raised inside `f` back to the caller through its C++ loop
(/root/reference/py/multivariate_py.cpp:385-388: pybind's error_already_set) and has undefined
behaviour on a NaN fitness (`std::sort` over `_fitness`, /root/reference/src/multivariate/cma/
base_cmaes.cpp:221).  Here: a NaN fitness ranks as +inf, +-Inf are ordinary
values of the ordering, an exception of the callable leaves the C loop with its ORIGINAL type, and
the handle serves a later, well-behaved problem as if nothing had happened."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 6
LO, UP = -5. * np.ones(N), 5. * np.ones(N)
GUESS = np.random.default_rng(11).uniform(-3, 3, N)


def _sphere(x):
    return float(np.dot(x, x))


def _make(hip, family, mfev=4000):
    if family == "ActiveCMAES":
        return hip.ActiveCMAES(mfev=mfev, tol=1e-8, np=12, seed=3)
    if family == "SHADE":
        return hip.SHADE(mfev=mfev, npinit=24, tol=1e-8, seed=3)
    return hip.APSO(mfev=mfev, tol=1e-8, np=16, seed=3)


class _Counting:
    """f(x) = |x|^2, except what `rule(call index, x)` returns when it is not None"""

    def __init__(self, rule):
        self.calls, self.rule = 0, rule

    def __call__(self, x):
        k = self.calls
        self.calls += 1
        v = self.rule(k, x)
        return _sphere(x) if v is None else v


FAMILIES = ["ActiveCMAES", "SHADE", "APSO"]


@pytest.mark.parametrize("family", FAMILIES)
def test_nan_for_some_candidates_ranks_as_plus_infinity(hip, family):
    """every third evaluation returns NaN: the run returns, and no candidate whose fitness was NaN
    is ever the incumbent (CMA: they close the ranking; DE / PSO: they never replace a parent or a
    personal best)"""
    f = _Counting(lambda k, x: float("nan") if k % 3 == 1 else None)
    g = _make(hip, family)
    g.initialize(f, LO, UP, GUESS)
    for _ in range(6):
        g.iterate()
        if family == "ActiveCMAES":
            fit = g.get_state("fit_val")
            lam = 12
            nbad = int(np.isinf(fit[:lam]).sum())
            assert nbad >= 1 and not np.isnan(fit).any()
            assert np.all(np.diff(fit[:lam][np.isfinite(fit[:lam])]) >= 0)
            assert np.isinf(fit[lam - nbad:lam]).all()          # +inf closes the ranking
        elif family == "SHADE":
            assert not np.isnan(g.get_state("f")).any()
        else:
            assert not np.isnan(g.get_state("fb")).any()
    sol = g.solution()
    assert np.isfinite(sol.x).all() and np.isfinite(_sphere(sol.x))
    # and the whole loop still terminates by its own rules
    g2 = _make(hip, family, mfev=3000)
    sol = g2.optimize(_Counting(lambda k, x: float("nan") if k % 3 == 1 else None), LO, UP, GUESS)
    assert sol.n_evals <= 3000 + 64 and np.isfinite(sol.x).all()


@pytest.mark.parametrize("family", FAMILIES)
def test_nan_for_all_candidates_returns(hip, family):
    """an objective that is NaN everywhere: optimize() returns within its budget, x* stays inside
    the box / finite, and the handle then solves a well-formed problem"""
    g = _make(hip, family, mfev=4000)
    sol = g.optimize(lambda x: float("nan"), LO, UP, GUESS)
    assert sol.n_evals <= 4000 + 64
    assert sol.x.shape == (N,)
    sol = g.optimize(_sphere, LO, UP, GUESS)              # the same handle, reused
    assert np.isfinite(sol.x).all() and _sphere(sol.x) < 1e-2


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("value", [float("inf"), float("-inf")])
def test_infinite_fitness_values(hip, family, value):
    """+Inf on a region (a common 'infeasible' marker) is simply never chosen; -Inf for ONE
    candidate makes it the best of its generation.  Neither stops the run from returning."""
    if value > 0:
        f = _Counting(lambda k, x: value if x[0] > 1. else None)
        g = _make(hip, family, mfev=3000)
        sol = g.optimize(f, LO, UP, GUESS)
        assert sol.n_evals <= 3000 + 64 and np.isfinite(sol.x).all()
        assert sol.x[0] <= 1. + 1e-12 and _sphere(sol.x) < 1e-2
    else:
        f = _Counting(lambda k, x: value if k == 30 else None)
        g = _make(hip, family, mfev=1200)
        sol = g.optimize(f, LO, UP, GUESS)
        assert sol.n_evals <= 1200 + 64 and sol.x.shape == (N,)
        assert not np.isnan(sol.x).any()


class _Boom(ValueError):
    pass


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("entry", ["optimize", "iterate", "run"])
def test_exception_in_the_objective_propagates_and_the_handle_survives(hip, family, entry):
    """multivariate_py.cpp:385-388: the exception raised by the k-th call comes out of optimize()
    / iterate() / run() with its own type and message, the callable is not called again after it
    failed, and the same object then optimizes a well-formed problem"""
    k_fail = 40

    def rule(k, x):
        if k == k_fail:
            raise _Boom("call %d" % k)
        return None

    f = _Counting(rule)
    g = _make(hip, family, mfev=4000)
    with pytest.raises(_Boom, match="call %d" % k_fail):
        if entry == "optimize":
            g.optimize(f, LO, UP, GUESS)
        else:
            g.initialize(f, LO, UP, GUESS)
            if entry == "iterate":
                for _ in range(50):
                    g.iterate()
            else:
                g.run(50)
    assert f.calls == k_fail + 1
    sol = g.optimize(_sphere, LO, UP, GUESS)
    assert _sphere(sol.x) < 1e-2, str(sol)


@pytest.mark.parametrize("family", FAMILIES)
def test_exception_in_a_vectorized_objective(hip, family):
    """the batch-callback form (one call per generation) fails the same way"""
    state = {"calls": 0}

    def fv(X):
        state["calls"] += 1
        if state["calls"] == 3:
            raise KeyError("batch 3")
        return (X * X).sum(axis=1)
    fv._bbo_vectorized = True
    g = _make(hip, family)
    with pytest.raises(KeyError, match="batch 3"):
        g.optimize(fv, LO, UP, GUESS)
    sol = g.optimize(_sphere, LO, UP, GUESS)
    assert _sphere(sol.x) < 1e-2
