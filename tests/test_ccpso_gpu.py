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


class _ForeignLocal:
    """any object with optimize(f, lower, upper, guess): what the Python class drives itself"""

    def __init__(self, inner):
        self._inner, self._params = inner, inner._params

    def reseed(self, seed):
        self._inner.reseed(seed)

    def optimize(self, f, lower, upper, guess):
        return self._inner.optimize(f, lower, upper, guess)


def test_local_search_inside_the_library_equals_the_python_driven_one(hip, oracle_lib):
    """bbo_ccpso_set_local (a CMA-ES object of this package: the whole loop behind bbo_iterate,
    what a C caller of INTEGRATION.md section B gets) against the same search driven from the
    Python class through bbo_get / bbo_set (any foreign object with optimize()): the same
    evaluation counts, context vector and swarm after every generation, bit for bit -- and the C
    ABI used directly, without the Python class"""
    import ctypes as C
    from bboptpy_amd import _ffi
    n, npp, cps, lf = 24, 8, 4, 2
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    f = lambda x: oracle_lib.objective("ellipsoid", x)
    runs = []
    for native in (True, False):
        loc = hip.ActiveCMAES(mfev=320, tol=1e-9, np=16, seed=77)
        g = hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=npp, pps=[cps], localfreq=lf, seed=31,
                      local=loc if native else _ForeignLocal(loc))
        assert g._local_native == native
        g.initialize(f, lo, up, np.zeros(n))
        trace = []
        for _ in range(7):
            g.iterate()
            trace.append((int(g.get_state("fev")[0]), int(g.get_state("improved")[0]),
                          g.get_state("yhat").copy(), g.get_state("x").copy(),
                          float(g.get_state("fyhat")[0])))
        runs.append(trace)
    for a, b in zip(*runs):
        assert a[0] == b[0] and a[1] == b[1] and a[4] == b[4]
        np.testing.assert_array_equal(a[2], b[2])
        np.testing.assert_array_equal(a[3], b[3])
    # the C ABI by hand: two handles, bbo_ccpso_set_local, bbo_optimize on a built-in objective
    L = _ffi.lib()
    pl = _ffi.default_params(_ffi.ALGO_CMAES)
    pl.mfev, pl.tol, pl.np, pl.seed = 200, 1e-8, 8, 5
    pc = _ffi.default_params(_ffi.ALGO_CCPSO)
    pc.mfev, pc.tol, pc.np, pc.npps, pc.seed = 40000, 1e-6, 10, 1, 9
    pc.pps[0] = 3
    hl, hc = C.c_void_p(), C.c_void_p()
    _ffi.check(L.bbo_create(C.byref(pl), C.byref(hl)))
    _ffi.check(L.bbo_create(C.byref(pc), C.byref(hc)))
    _ffi.check(L.bbo_ccpso_set_local(hc, hl, 5), hc)
    obj = _ffi.Objective()
    obj.kind, obj.builtin = _ffi.OBJ_BUILTIN, _ffi.BUILTIN_IDS["sphere"]
    m = 12
    x, fev, conv = np.zeros(m), C.c_int(), C.c_int()
    _ffi.check(L.bbo_optimize(hc, m, -5. * np.ones(m), 5. * np.ones(m), np.zeros(m), C.byref(obj),
                              x, C.byref(fev), C.byref(conv)), hc)
    assert 0 < fev.value <= 40000 + 400 and float(np.sum(x * x)) < 1e-3
    assert L.bbo_ccpso_set_local(hc, hc, 5) < 0           # itself: refused
    assert L.bbo_ccpso_set_local(hl, hc, 5) < 0           # not a CCPSO handle: refused
    _ffi.check(L.bbo_ccpso_set_local(hc, None, 0), hc)    # detach before the local one goes
    L.bbo_destroy(hl)
    L.bbo_destroy(hc)


def test_run_searches_locally_in_the_generation_that_spends_the_budget(hip, oracle_lib):
    """(advisor, round 4) the reference's iterate() runs localSearch before optimize() looks at the
    budget (ccpso.cpp:112-147); bbo_run used to skip the search of the generation that raised the
    stop flag, so run() / optimize() and an iterate()-driven loop could end at different points
    whenever the last generation was a local-search generation.  Both are held together here for
    budgets that end on either kind of generation."""
    n, npp, cps, lf = 24, 8, 4, 2
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    kinds = set()
    for mfev in (150, 250, 350, 450, 560, 700):
        out = []
        for driver in ("iterate", "run"):
            loc = hip.ActiveCMAES(mfev=60, tol=1e-9, np=8, seed=77)
            g = hip.CCPSO(mfev=mfev, sigmatol=1e-300, np=npp, pps=[cps], localfreq=lf, seed=31,
                          local=loc)
            assert g._local_native
            g.initialize(hip.objectives.ellipsoid, lo, up, np.zeros(n))
            if driver == "iterate":
                gens = 0
                while True:                      # the reference's optimize() loop
                    g.iterate()
                    gens += 1
                    if int(g.get_state("fev")[0]) >= mfev:
                        break
                kinds.add((gens - 1) % lf == 0)
            else:
                g.run(10 ** 6)
            out.append((int(g.get_state("fev")[0]), float(g.get_state("fyhat")[0]),
                        g.get_state("yhat").copy(), int(g.get_state("it")[0])))
        a, b = out
        assert a[0] == b[0] and a[1] == b[1] and a[3] == b[3], (mfev, a[0], b[0], a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])
    assert kinds == {True, False}, kinds         # both kinds of last generation were exercised


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
