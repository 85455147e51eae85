"""SepCMAES (diagonal covariance) on the device against the CPU oracle, through the C ABI.

The oracle's SepCmaes restatement is pinned bit for bit against the compiled reference
(tests/test_oracle_vs_reference.py, tests/golden/sep_runs.json); here every phase of a device
generation is compared with it under the device's own normals."""
import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu


def _close(a, b, rtol=1e-12, atol=0., what=""):
    a, b = np.asarray(a, float), np.asarray(b, float)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * max(np.abs(b).max(), 1e-300), "%s: err %.3e" % (what, err)


@pytest.mark.parametrize("n,lam,obj,bound,adjustlr", [
    (10, 20, "ellipsoid", False, False),
    (37, 50, "rastrigin", True, True),        # ragged n and lambda, box, learning-rate adjustment
    (128, 256, "rosenbrock", False, False),
    (1500, 64, "sphere", False, True),        # n > 1024: 64 lanes per candidate
    (24, 32, "ellipsoid", False, None),       # None: keyword omitted -> the reference's default
])
def test_generation_matches_oracle(hip, oracle_lib, n, lam, obj, bound, adjustlr):
    from bboptpy_amd import _ffi
    rng = np.random.default_rng(n)
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = rng.uniform(-4, 4, n)
    kw = {} if adjustlr is None else {"adjustlr": adjustlr}
    g = hip.SepCMAES(mfev=10 ** 8, tol=1e-14, np=lam, sigma0=1.5, bound=bound, seed=99, **kw)
    g.initialize(getattr(hip.objectives, obj), lo, up, guess)
    g.set_state("record_normals", [1.0])
    # the default path (py/multivariate_py.cpp:131-135 binds adjustlr=true): the oracle is told
    # True explicitly, the device gets no keyword -- it must pick the same variant by itself
    o = po.cma(oracle_lib, "sep", 10 ** 8, 1e-14, lam, sigma0=1.5, bound=bound,
               adjustlr=True if adjustlr is None else adjustlr)
    if adjustlr is None:
        off = po.cma(oracle_lib, "sep", 10 ** 8, 1e-14, lam, sigma0=1.5, bound=bound,
                     adjustlr=False)
        off.init(obj, lo, up, guess)
        o.init(obj, lo, up, guess)
        assert off.scalar("ccov") != o.scalar("ccov")      # the variants do differ here
    o.set_rng(po.RNG_INJECT)
    o.init(obj, lo, up, guess)
    for key in ("mueff", "cc", "cs", "ccov", "damps", "chi"):
        assert g.get_state(key)[0] == o.scalar(key), key
    for gen in range(8):
        it = int(g.get_state("it")[0])
        g.phase(_ffi.PHASE_SAMPLE_EVALUATE)
        z = g.get_state("zlast")
        want = np.zeros(lam * n)
        oracle_lib.f("philox_normals")(99, it, lam, n, want)
        np.testing.assert_array_equal(z, want)          # same normals as the oracle's statement
        o.inject_z(z)
        o.step("sample")
        o.step("evaluate_sort")
        _close(g.get_state("arx"), o.get("arx"), rtol=1e-14, what="arx")
        g.phase(_ffi.PHASE_RANK)
        fo = o.get("fit_val")
        _close(g.get_state("fit_val"), fo, rtol=1e-11, what="sorted fitness")
        gi, oi = g.get_state("fit_idx").astype(int), o.get("fit_idx").astype(int)
        if not np.array_equal(gi, oi):
            pytest.skip("fitness tie to rounding changed the ranking")
        g.phase(_ffi.PHASE_UPDATE)
        g.phase(_ffi.PHASE_EIGEN)          # no-op for a diagonal covariance
        o.step("update_distribution")
        for key, tol in (("xmean", 1e-13), ("ps", 1e-11), ("pc", 1e-11), ("csep", 1e-11),
                         ("D", 1e-11)):
            _close(g.get_state(key), o.get(key), rtol=tol, what="%s gen %d" % (key, gen))
        _close(g.get_state("sigma"), [o.scalar("sigma")], rtol=1e-11, what="sigma")
        g.phase(_ffi.PHASE_HISTORY_STOP)
        o.step("update_history")           # (includes it++, like the device phase)
        assert int(g.get_state("flag")[0]) == o.converged()
        # keep the two on the same trajectory to the last bit for the next generation
        for key in ("xmean", "ps", "pc", "csep"):
            o.set(key, g.get_state(key))
        o.set("D", g.get_state("D"))
        o.set("sigma", g.get_state("sigma"))


def test_uncapped_learning_rate_matches_the_reference(hip, oracle_lib):
    """The reference multiplies c_cov by (n + 2)/3 without a cap (sep_cmaes.cpp:56-58, on by
    default): for lambda large against n, c_cov > 1, the diagonal covariance turns negative and the
    reference's state is NaN within two generations.  A drop-in does the same: same c_cov, and
    non-finite from the same generation under the same normals -- not a silent repair."""
    n, lam = 7, 300
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.full(n, 1.5)
    g = hip.SepCMAES(mfev=10 ** 7, tol=1e-12, np=lam, seed=5)
    g.initialize(hip.objectives.sphere, lo, up, guess)
    o = po.cma(oracle_lib, "sep", 10 ** 7, 1e-12, lam, adjustlr=True)
    o.set_rng(po.RNG_PHILOX, 5)
    o.init("sphere", lo, up, guess)
    assert g.get_state("ccov")[0] == o.scalar("ccov") > 1.
    first = {}
    for gen in range(6):
        g.iterate()
        o.iterate()
        for who, x in (("device", g.get_state("xmean")), ("oracle", o.get("xmean", n))):
            if who not in first and not np.all(np.isfinite(x)):
                first[who] = gen
    assert "oracle" in first and first.get("device") == first["oracle"], first


def test_sep_rejects_full_covariance_keys(hip):
    g = hip.SepCMAES(mfev=1000, tol=1e-8, np=8)
    g.initialize(hip.objectives.sphere, -np.ones(4), np.ones(4), np.zeros(4))
    with pytest.raises(Exception):
        g.get_state("B")


@pytest.mark.parametrize("obj,n", [("ellipsoid", 64), ("discus", 30)])
def test_sep_whole_run_matches_oracle(hip, oracle_lib, obj, n):
    """End to end with the SAME random numbers: the oracle draws the device's Philox normals
    (same seed, same counter layout), so both take the same trajectory: same number of
    evaluations (within a generation or two of rounding luck), same stop flag, same optimum.
    (Smooth objectives only: on Rastrigin the last-bit difference between a 16-lane tree sum and
    a serial sum of f flips a ranking sooner or later and the two runs drift apart.)"""
    lam = 4 * (4 + int(3 * np.log(n)))
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(1).uniform(-4, 4, n)
    g = hip.SepCMAES(mfev=400000, tol=1e-10, np=lam, sigma0=2., adjustlr=True, seed=3)
    sol = g.optimize(getattr(hip.objectives, obj), lo, up, guess)
    o = po.cma(oracle_lib, "sep", 400000, 1e-10, lam, sigma0=2., adjustlr=True)
    o.set_rng(po.RNG_PHILOX, 3)
    xo, fevo, convo = o.optimize(obj, lo, up, guess)
    assert sol.converged and convo
    assert abs(sol.n_evals - fevo) <= 2 * lam
    assert int(g.get_state("flag")[0]) == int(o.scalar("flag"))
    f_dev, f_cpu = oracle_lib.objective(obj, sol.x), oracle_lib.objective(obj, xo)
    assert abs(f_dev - f_cpu) <= 1e-5 * max(abs(f_cpu), 1e-300) + 1e-300
