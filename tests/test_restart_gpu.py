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


@pytest.mark.parametrize("driver,nflag,n,obj,seed", [
    ("bipop", True, 6, "rastrigin", 21),
    ("bipop", False, 6, "ellipsoid", 22),
    ("bipop", True, 10, "rosenbrock", 25),
    ("ipop", True, 6, "rastrigin", 24),
    ("ipop", False, 8, "schwefel12", 25),
])
def test_restart_decisions_match_oracle_restart(hip, oracle_lib, driver, nflag, n, obj, seed):
    """Every DECISION of the device's restart drivers against the oracle's Restart
    (bipop_cmaes.cpp:109-267 / ipop_cmaes.cpp:112-162 restated, pinned bit for bit to the
    compiled reference): before each restart the oracle takes over the device driver's
    bookkeeping (budgets, counters, incumbent value, best regime), both plan and run the restart,
    and the plan -- restart point (same Philox counter (run, k) on both sides), regime, lambda,
    sigma, evaluation cap -- must be IDENTICAL; so must the budget arithmetic applied to what the
    device's own inner run reported.

    The inner runs themselves are compared exactly only for the first run and the first restart
    (evaluations used; f* to 1e-6 relative + 1e-7 absolute; for the first restart that holds for
    the seeds used here and for 19 of 20 consecutive seeds of the Rosenbrock case,
    scripts/dev_first_restart_sweep.py -- the twentieth parts the way described next, one
    restart early).  Beyond that they legitimately part, and fast -- measured on this path (scripts/dev_restart_divergence.py): a restarted
    run samples its first generation through the B the previous run left behind
    (cmaes.cpp:53-59); CMA-ES is translation invariant, so the difference e that B carries
    into the mean (1e-11 after one restart) persists unchanged while both sides rank their
    candidates identically, and once sigma has shrunk to where e matters (tol = 1e-6 drives f to
    1e-9) rankings flip: the run ends with C different at the 1e-2 level, and the NEXT run,
    started through that B, is a different random run.  On (nearly) isotropic optima
    (Rastrigin, sphere) the carried-over eigenvectors are not even a function of C (clustered
    eigenvalues: C equal to 6e-8, B different by O(1)).  Whole inner runs are held against the
    oracle where that is well defined: tests/test_cma_gpu.py::test_whole_run_same_seed_*."""
    import pyoracle as po
    mfev = 40000
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    base = hip.ActiveCMAES(mfev=1, tol=1e-6, np=4)
    ob = po.cma(oracle_lib, "active", 1, 1e-6, 4)
    if driver == "bipop":
        drv = hip.BiPopCMAES(base, mfev=mfev, nbipop=nflag, seed=seed)
        o = po.bipop(oracle_lib, ob, mfev, nbipop=nflag)
        book = ("fev", "it", "largelambda", "largebudget", "smallbudget", "largerestarts",
                "smallrestarts", "bestregime", "fxbest")
    else:
        drv = hip.IPopCMAES(base, mfev=mfev, nipop=nflag, seed=seed)
        o = po.ipop(oracle_lib, ob, mfev, nipop=nflag)
        book = ("fev", "it", "lambda", "sigma", "fxbest")
    drv.initialize(getattr(hip.objectives, obj), lo, up, guess)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init(obj, lo, up, guess)
    gd = lambda k: drv.get_state(k)[0]
    go = lambda k: o.scalar(k)
    same_f = lambda a, b: a == pytest.approx(b, rel=1e-6, abs=1e-7)
    plan = ("last_regime", "last_lambda", "last_sigma", "last_maxfev")
    # the first default run: nothing inherited, whole run comparable
    assert [gd(k) for k in plan] == [go(k) for k in plan]
    assert int(gd("last_inner_fev")) == int(go("last_inner_fev"))
    assert int(gd("fev")) == int(go("fev")) and same_f(gd("fx"), go("fx"))
    rows = 0
    for _ in range(14):
        if gd("fev") >= mfev or (driver == "bipop" and gd("largerestarts") >= 9):
            break
        for k in book:                       # the oracle continues from the DEVICE's history
            o.rset(k, gd(k))
        before = {k: gd(k) for k in book}
        drv.iterate()
        o.iterate()
        np.testing.assert_array_equal(drv.get_state("x0"), o.get("x0"))
        assert [gd(k) for k in plan] == [go(k) for k in plan], "restart %d plan" % (rows + 1)
        used, fx = int(gd("last_inner_fev")), gd("fx")
        if rows == 0:                        # first restart: the inner run itself, too
            assert used == int(go("last_inner_fev"))
            assert same_f(fx, go("fx"))
        # the arithmetic on the device's own inner result (bipop_cmaes.cpp:223-236,254-267)
        assert int(gd("fev")) == int(before["fev"]) + used + 1
        assert int(gd("it")) == int(before["it"]) + 1
        better = fx < before["fxbest"]
        assert gd("fxbest") == (fx if better else before["fxbest"])
        if driver == "bipop":
            reg = int(gd("last_regime"))
            assert int(gd("largebudget")) == int(before["largebudget"]) + (used if reg == 1 else 0)
            assert int(gd("smallbudget")) == int(before["smallbudget"]) + (used if reg == 2 else 0)
            assert int(gd("largerestarts")) == int(before["largerestarts"]) + (reg == 1)
            assert int(gd("smallrestarts")) == int(before["smallrestarts"]) + (reg == 2)
            assert int(gd("bestregime")) == (reg if better else int(before["bestregime"]))
        rows += 1
    assert rows >= 4
    sol = drv.solution()
    assert not sol.converged and sol.n_evals == int(gd("fev"))
    assert getattr(hip.objectives, obj)(sol.x) == pytest.approx(gd("fxbest"), rel=1e-9, abs=1e-12)


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
