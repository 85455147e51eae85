"""GPU parity: one CCPSO2 generation (regrouping, the 2 nswarm np context-vector evaluations,
personal / swarm / ring bests, the yhat re-evaluation, the Cauchy rate, the resampling) against
the oracle in Philox mode.  The oracle's mt19937 form is pinned bit for bit to the compiled
reference; the Philox form differs only in where the random numbers come from and in regrouping
with the keyed Feistel bijection instead of std::shuffle."""
import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if a.size == 0:
        return
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert err <= rtol, "%s: rel err %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("n,npp,obj,pps,kw", [
    (12, 8, "rastrigin", [2, 3, 6], {}),
    (20, 10, "rosenbrock", [5, 10], dict(correct=False)),
    (16, 6, "sphere", [1, 2, 4, 8, 16], dict(pcauchy=0.3)),
    (300, 24, "ellipsoid", [5, 10, 50], {}),             # ld > 256: 64 lanes per team
    (600, 7, "rosenbrock", [300, 600], {}),              # swarms wider than the 256 coordinates a
                                                         # team holds in registers; 2 np = 14 is
                                                         # not a multiple of the teams per swarm
    (40, 5, "sphere", [20, 40], {}),                     # the same with 16 lanes per team
])
def test_generations_match_oracle(hip, oracle_lib, n, npp, obj, pps, kw):
    seed = 555
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=pps, seed=seed, **kw)
    o = po.ccpso(oracle_lib, 10 ** 8, 1e-12, npp, pps, **kw)
    o.set_mode(True, po.RNG_PHILOX, seed)
    g.initialize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o.init(obj, lo, up, np.zeros(n))
    np.testing.assert_array_equal(g.get_state("x"), o.get("x"))
    np.testing.assert_array_equal(g.get_state("yhat"), o.get("yhat"))
    for gen in range(15 if n > 100 else 30):
        g.iterate()
        o.iterate()
        tag = "gen %d" % gen
        for k in ("is", "nswarm", "cpswarm", "fev", "improved"):
            assert int(g.get_state(k)[0]) == int(o.scalar(k)), tag + " " + k
        np.testing.assert_array_equal(g.get_state("k"), o.get("k"), err_msg=tag + " grouping")
        np.testing.assert_array_equal(g.get_state("ibest"), o.get("ibest"), err_msg=tag + " ibest")
        np.testing.assert_array_equal(g.get_state("strat"), o.get("strat"), err_msg=tag + " strat")
        _close(g.get_state("fx"), o.get("fx"), 1e-11, tag + " fx")
        _close(g.get_state("fy"), o.get("fy"), 1e-11, tag + " fy")
        _close(g.get_state("y"), o.get("y"), 1e-13, tag + " y")
        _close(g.get_state("x"), o.get("x"), 1e-10, tag + " x")
        _close(g.get_state("yhat"), o.get("yhat"), 1e-13, tag + " yhat")
        _close(g.get_state("fyhat"), [o.scalar("fyhat")], 1e-11, tag + " fyhat")
        assert abs(g.get_state("phat")[0] - o.scalar("phat")) <= 1e-12, tag + " phat"


def test_whole_run_same_seed_matches_oracle(hip, oracle_lib):
    n = 20
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.CCPSO(mfev=150000, sigmatol=1e-6, np=12, pps=[2, 5, 10], seed=8)
    sol = g.optimize(hip.objectives.sphere, lo, up, np.zeros(n))
    o = po.ccpso(oracle_lib, 150000, 1e-6, 12, [2, 5, 10])
    o.set_mode(True, po.RNG_PHILOX, 8)
    xo, fevo, convo = o.optimize("sphere", lo, up, np.zeros(n))
    assert sol.n_evals == fevo and sol.converged == convo
    np.testing.assert_allclose(sol.x, xo, rtol=0, atol=1e-9)


def test_python_callback_objective(hip):
    n = 6
    calls = [0]

    def fx(x):
        calls[0] += 1
        return float(np.sum((x - 1.) ** 2))

    g = hip.CCPSO(mfev=20000, sigmatol=1e-5, np=10, pps=[2, 3], seed=3)
    sol = g.optimize(fx, -5. * np.ones(n), 5. * np.ones(n), np.zeros(n))
    assert calls[0] == sol.n_evals
    assert np.abs(sol.x - 1.).max() < 0.1


def test_invalid_swarm_size_is_rejected(hip):
    g = hip.CCPSO(mfev=1000, sigmatol=1e-5, np=10, pps=[4], seed=3)
    with pytest.raises(Exception):
        g.initialize(hip.objectives.sphere, -np.ones(6), np.ones(6), np.zeros(6))
    with pytest.raises(TypeError):         # a local optimizer must offer optimize()
        hip.CCPSO(mfev=1000, sigmatol=1e-5, np=10, pps=[2], local=object())


@pytest.mark.parametrize("n,npp,cps,obj,variant,lf", [
    (24, 8, 4, "ellipsoid", "cmaes", 2),
    (20, 10, 5, "sphere", "active", 3)])
def test_local_optimizer_hook_matches_oracle(hip, oracle_lib, n, npp, cps, obj, variant, lf):
    """CCPSO with its local optimizer (ccpso.cpp:116-118, 371-435): the device runs the
    generations, the Python class drives localSearch with a device CMA-ES on the swarm weights
    (objective evaluated on the host); the oracle -- pinned bit for bit to the reference for this
    path -- does the same with its own CMA-ES drawing the same Philox normals (search q: seed
    base + q, started from B = C = I on both sides).  Same evaluation counts, same context
    vector after every generation."""
    seed, lseed = 31, 77
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    # (lambda = 16 >= 2 nswarm: with mu < n - 1 the first covariance matrices of plain CMA-ES
    # have a repeated eigenvalue, whose eigenvectors -- and with them the samples -- are decided
    # by rounding inside the eigensolver: same distribution, different trajectory)
    Local = hip.CMAES if variant == "cmaes" else hip.ActiveCMAES
    loc = Local(mfev=320, tol=1e-9, np=16, seed=lseed)
    g = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=[cps], local=loc, localfreq=lf,
                  seed=seed)
    o = po.ccpso(oracle_lib, 10 ** 8, 1e-12, npp, [cps],
                 local=po.cma(oracle_lib, variant, 320, 1e-9, 16), localfreq=lf, local_seed=lseed,
                 local_fresh=True)
    o.set_mode(True, po.RNG_PHILOX, seed)
    # the objective as a host callable that evaluates the oracle's own formula: every f value --
    # in the generations and inside the local searches -- is then the same double on both
    # sides, and the two CMA-ES runs rank their candidates identically
    g.initialize(lambda x: oracle_lib.objective(obj, x), lo, up, np.zeros(n))
    o.init(obj, lo, up, np.zeros(n))
    improved_by_local = 0
    for gen in range(10):
        f_before = float(g.get_state("fyhat")[0])
        g.iterate()
        o.iterate()
        tag = "gen %d" % gen
        assert int(g.get_state("fev")[0]) == int(o.scalar("fev")), tag + " fev"
        assert int(g.get_state("improved")[0]) == int(o.scalar("improved")), tag + " improved"
        _close(g.get_state("yhat"), o.get("yhat"), 1e-8, tag + " yhat")
        _close(g.get_state("fyhat"), [o.scalar("fyhat")], 1e-8, tag + " fyhat")
        _close(g.get_state("x"), o.get("x"), 1e-8, tag + " x")
        if gen % lf == 0 and float(g.get_state("fyhat")[0]) < f_before:
            improved_by_local += 1
    assert improved_by_local >= 1


def test_local_optimizer_whole_run(hip):
    """optimize() with a local optimizer: the reference's loop (generation, local search every
    localfreq generations, budget test, spread test) driven from the Python class"""
    n = 12
    loc = hip.CMAES(mfev=200, tol=1e-8, np=8, seed=5)
    g = hip.CCPSO(mfev=40000, sigmatol=1e-6, np=10, pps=[3], local=loc, localfreq=5, seed=9)
    sol = g.optimize(hip.objectives.sphere, -5. * np.ones(n), 5. * np.ones(n), np.zeros(n))
    assert sol.n_evals > 0 and float(np.sum(sol.x ** 2)) < 1e-3
    with pytest.raises(TypeError):
        hip.CCPSO(mfev=100, sigmatol=1e-6, np=4, pps=[2], local=object())
