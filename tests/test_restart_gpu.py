"""GPU: the IPOP / BIPOP restart drivers (host logic of libbbopt_hip over device-side CMA-ES
runs).  Inner trajectories are chaotic over hundreds of generations, so the schedule is checked
through the reference's own rules (bipop_cmaes.cpp:109-267, ipop_cmaes.cpp:112-162) applied to
the quantities the driver reports, and the outcome through the optimum it finds; the rules
themselves are pinned bit-for-bit on the CPU (tests/test_oracle_golden.py::test_restart_schedule).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _max_evals(n, lam, mfev, fev):
    maxit = int(100. + 50. * (n + 3) * (n + 3) / math.sqrt(1. * lam))
    return min(maxit * lam, mfev - fev)


def test_bipop_schedule_follows_the_reference_rules(hip):
    n, mfev = 6, 60000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    base = hip.ActiveCMAES(mfev=1, tol=1e-6, np=4)
    drv = hip.BiPopCMAES(base, mfev=mfev, seed=21)
    drv.initialize(hip.objectives.rastrigin, lo, up, np.random.default_rng(1).uniform(-5, 5, n))
    lamdef = 4 + int(3. * math.log(n))
    assert int(drv.get_state("lambdadef")[0]) == lamdef
    fev = int(drv.get_state("fev")[0])
    assert fev == int(drv.get_state("last_inner_fev")[0]) + 1     # the +1 re-evaluation
    large = small = 0
    nl = ns = 0
    best_regime, fbest = 1, drv.get_state("fxbest")[0]
    large_lambda = None
    for _ in range(40):
        if nl >= 9 or fev >= mfev:
            break
        want = (1 if large <= small * 2. else 2) if best_regime == 1 else \
               (2 if small <= 2. * large else 1)
        drv.iterate()
        regime = int(drv.get_state("last_regime")[0])
        lam = int(drv.get_state("last_lambda")[0])
        sig = drv.get_state("last_sigma")[0]
        used = int(drv.get_state("last_inner_fev")[0])
        assert regime == want
        if regime == 1:
            assert lam == int(lamdef * 2 ** (nl + 1))
            assert sig == max(2. * (1. / 1.6) ** (nl + 1), 0.02)
            assert used <= max(_max_evals(n, lam, mfev, fev), 0) + lam
            large += used
            nl += 1
            large_lambda = lam
        else:
            assert lamdef <= lam <= max(lamdef, large_lambda // 2)
            assert 2e-2 * (1 - 1e-12) <= sig <= 2.
            cap = min(_max_evals(n, lam, mfev, fev), large >> 1)
            assert used <= max(cap, 0) + lam
            small += used
            ns += 1
        fev += used + 1
        assert int(drv.get_state("fev")[0]) == fev
        fx = drv.get_state("fx")[0]
        if fx < fbest:
            fbest, best_regime = fx, regime
        assert drv.get_state("fxbest")[0] == fbest
        assert int(drv.get_state("bestregime")[0]) == best_regime
        assert (int(drv.get_state("largebudget")[0]), int(drv.get_state("smallbudget")[0])) \
            == (large, small)
    sol = drv.solution()
    assert not sol.converged                       # bipop_cmaes.cpp:166-168
    assert hip.objectives.rastrigin(sol.x) == pytest.approx(fbest, rel=1e-9, abs=1e-9)


def test_bipop_finds_the_rastrigin_optimum(hip):
    """statistical end-to-end check.  The reference itself (oracle/_ref, seeds 1..10, same
    configuration) reaches f < 1e-6 in 6 of 10 runs and f = 0.995 (the first local minimum)
    in the others; the device path must do as well as the lower end of that band."""
    n = 6
    vals = []
    for seed in range(1, 9):
        base = hip.ActiveCMAES(mfev=1, tol=1e-8, np=4)
        drv = hip.BiPopCMAES(base, mfev=200000, seed=seed)
        sol = drv.optimize(hip.objectives.rastrigin, -5. * np.ones(n), 5. * np.ones(n),
                           np.random.default_rng(seed).uniform(-5, 5, n))
        assert sol.n_evals <= 200000 + 20000
        vals.append(hip.objectives.rastrigin(sol.x))
    assert sum(v < 1e-6 for v in vals) >= 3, vals
    assert max(vals) < 2.1, vals


def test_ipop_doubles_lambda_and_shrinks_sigma(hip):
    n, mfev = 5, 30000
    base = hip.CMAES(mfev=1, tol=1e-6, np=4)
    drv = hip.IPopCMAES(base, mfev=mfev, seed=5)
    drv.initialize(hip.objectives.rastrigin, -5. * np.ones(n), 5. * np.ones(n), np.zeros(n))
    lam = 4 + int(3. * math.log(n))
    sig = 2.
    fev = int(drv.get_state("fev")[0])
    for _ in range(12):
        if fev >= mfev:
            break
        drv.iterate()
        lam <<= 1
        if lam > 10 * n * n:
            lam = 10 * n * n if lam - 10 * n * n < 10 * n * n - (lam >> 1) else 4 + int(3. * math.log(n))
        sig = max(sig / 1.6, 0.02)
        assert int(drv.get_state("lambda")[0]) == lam
        assert drv.get_state("sigma")[0] == sig
        fev += int(drv.get_state("last_inner_fev")[0]) + 1
        assert int(drv.get_state("fev")[0]) == fev
    sol = drv.solution()
    assert sol.n_evals == fev and not sol.converged


# ---- the device drivers against the oracle's Restart (itself pinned bit for bit to the
# reference's BiPopCmaes / IPopCmaes under mt19937, tests/test_oracle_vs_reference.py) -------
def _schedule_row(h, get):
    return (int(get(h, "last_regime")), int(get(h, "last_lambda")), get(h, "last_sigma"),
            int(get(h, "last_inner_fev")), int(get(h, "fev")))


@pytest.mark.parametrize("driver,variant,n,obj,seed", [
    ("bipop", "active", 6, "rastrigin", 21),
    ("bipop", "active", 10, "rosenbrock", 22),
    ("bipop", "cmaes", 5, "ellipsoid", 23),
    ("ipop", "active", 6, "rastrigin", 24),
    ("ipop", "cmaes", 8, "sphere", 25),
])
def test_restart_schedule_matches_oracle_restart(hip, oracle_lib, driver, variant, n, obj, seed):
    """Same Philox key on both sides: the oracle's Restart draws the driver's restart points,
    u and u' from the RESTART stream and runs restart r's inner CMA-ES under the key
    seed + golden * (r + 1), exactly the device's rule (bbo_restart.hip); n <= 16, so the
    device eigensolver has the reference's sign conventions and whole inner runs coincide.
    Compared per restart: (regime, lambda, sigma, evaluations used, budget) exactly, the
    returned f* to 1e-7 relative (hundreds of generations of rounding), the bookkeeping exactly."""
    import pyoracle as po
    mfev = 40000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    cls = hip.ActiveCMAES if variant == "active" else hip.CMAES
    base = cls(mfev=1, tol=1e-6, np=4)
    drv = (hip.BiPopCMAES if driver == "bipop" else hip.IPopCMAES)(base, mfev=mfev, seed=seed)
    drv.initialize(getattr(hip.objectives, obj), lo, up, guess)
    ob = po.cma(oracle_lib, variant, 1, 1e-6, 4)
    o = getattr(po, driver)(oracle_lib, ob, mfev)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init(obj, lo, up, guess)
    gd = lambda h, k: h.get_state(k)[0]
    go = lambda h, k: h.scalar(k)
    assert int(gd(drv, "fev")) == int(go(o, "fev"))
    assert gd(drv, "fx") == pytest.approx(go(o, "fx"), rel=1e-7, abs=1e-12)
    rows = 0
    for _ in range(14):
        if go(o, "fev") >= mfev or (driver == "bipop" and go(o, "largerestarts") >= 9):
            break
        drv.iterate()
        o.iterate()
        np.testing.assert_array_equal(drv.get_state("x0"), o.get("x0"))      # same restart point
        if driver == "bipop":
            assert _schedule_row(drv, gd) == _schedule_row(o, go)
            for k in ("largebudget", "smallbudget", "largerestarts", "smallrestarts",
                      "bestregime"):
                assert int(gd(drv, k)) == int(go(o, k)), k
        else:
            for k in ("lambda", "last_inner_fev", "fev"):
                assert int(gd(drv, k)) == int(go(o, k)), k
            assert gd(drv, "sigma") == go(o, "sigma")
        assert gd(drv, "fx") == pytest.approx(go(o, "fx"), rel=1e-7, abs=1e-12)
        assert gd(drv, "fxbest") == pytest.approx(go(o, "fxbest"), rel=1e-7, abs=1e-12)
        rows += 1
    assert rows >= 4
    np.testing.assert_allclose(drv.get_state("xbest"), o.get("xbest"), rtol=0, atol=1e-7)


# ---- print=True: the Tabular rows (tabular.hpp:65-77) -----------------------------------------
def _cells(line):
    assert line.startswith(" | ") and line.endswith(" | "), repr(line)
    return line[3:-3].split(" | ")


@pytest.mark.parametrize("driver", ["bipop", "ipop"])
def test_print_rows_have_the_reference_format(hip, capfd, driver):
    """header, rule and row layout against the text the compiled reference printed
    (tests/golden/restart_print.json); the numbers in each row against the driver's own state,
    formatted the reference's way (toStringFull: max_digits10 significant digits, %g style)"""
    from _golden import load
    from _tabular import fmt_cell, WIDTHS
    rec = [r for r in load("restart_print.json") if r["driver"] == driver][0]
    n = rec["n"]
    base = hip.ActiveCMAES(mfev=1, tol=1e-6, np=4)
    cls = hip.BiPopCMAES if driver == "bipop" else hip.IPopCMAES
    drv = cls(base, mfev=rec["mfev"], print=True, seed=3)
    capfd.readouterr()
    drv.initialize(getattr(hip.objectives, rec["objective"]), -5. * np.ones(n), 5. * np.ones(n),
                   np.random.default_rng(1).uniform(-5, 5, n))
    want = []

    def expect():
        g = lambda k: drv.get_state(k)[0]
        if driver == "bipop":
            reg = int(g("last_regime"))
            vals = [int(g("it")), reg, int(g("largerestarts")), int(g("smallrestarts")),
                    int(g("largebudget")), int(g("smallbudget")), int(g("fev")),
                    int(g("last_lambda")), g("last_sigma"), g("fx"), g("fxbest")]
        else:
            vals = [int(g("it")), int(g("fev")), int(g("last_lambda")), g("last_sigma"),
                    g("fx"), g("fxbest")]
        want.append(" | " + " | ".join(fmt_cell(v, w) for v, w in zip(vals, WIDTHS[driver]))
                    + " | ")

    expect()
    for _ in range(5):
        drv.iterate()
        expect()
    out = capfd.readouterr().out.split("\n")
    assert out[0] == rec["lines"][0]          # header, character for character
    assert out[1] == rec["lines"][1]          # the rule under it
    assert out[2:2 + len(want)] == want
    for line in out[2:2 + len(want)]:
        assert [len(c) for c in _cells(line)] == [max(w, len(c.strip())) for w, c in
                                                  zip(WIDTHS[driver], _cells(line))]
