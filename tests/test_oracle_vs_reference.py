"""The CPU oracle against the REAL reference, live (development container only: needs
oracle/_ref/libbbo_ref.so; skipped elsewhere).  Wider than the committed fixtures: several
shapes, objectives and seeds, every generation compared bit for bit."""
import numpy as np
import pytest

import pyoracle as po


def _same(a, b, keys, tag):
    for k in keys:
        np.testing.assert_array_equal(a.get(k), b.get(k), err_msg="%s: %s" % (tag, k))


@pytest.mark.parametrize("variant", ["active", "cmaes"])
@pytest.mark.parametrize("n,lam,obj", [(10, 20, "rosenbrock"), (5, 8, "sphere"),
                                       (24, 40, "rastrigin"), (13, 17, "ackley")])
def test_cma_bit_exact(oracle_lib, ref_lib, variant, n, lam, obj):
    oracle_lib.seed(21)
    ref_lib.seed(21)
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(n).uniform(-5, 5, n)
    o = po.cma(oracle_lib, variant, 4000 * lam, 1e-8, lam)
    r = po.cma(ref_lib, variant, 4000 * lam, 1e-8, lam)
    o.init(obj, lo, up, guess)
    r.init(obj, lo, up, guess)
    for it in range(150):
        o.iterate()
        r.iterate()
        _same(o, r, ("xmean", "sigma", "C", "B", "D", "invsqrtC", "pc", "ps", "arx", "fit_val",
                     "fit_idx"), "%s n=%d it=%d" % (variant, n, it))
        fo, fr = o.converged(), r.converged()
        assert fo == fr
        if fo:
            break


@pytest.mark.parametrize("n,lam,obj,bound,adjustlr", [
    (10, 20, "ellipsoid", False, False), (37, 50, "rastrigin", True, True),
    (64, 128, "rosenbrock", False, False), (5, 6, "cigar", True, False)])
def test_sep_cma_bit_exact(oracle_lib, ref_lib, n, lam, obj, bound, adjustlr):
    """SepCmaes (sep_cmaes.cpp), every generation and the stop decision"""
    oracle_lib.seed(77)
    ref_lib.seed(77)
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(n).uniform(-4, 4, n)
    hs = [po.cma(L, "sep", 10 ** 7, 1e-10, lam, sigma0=1.5, bound=bound, adjustlr=adjustlr)
          for L in (oracle_lib, ref_lib)]
    for h in hs:
        h.init(obj, lo, up, guess)
    for k in ("cc", "cs", "ccov", "damps", "mueff", "chi"):
        assert hs[0].scalar(k) == hs[1].scalar(k), k
    for it in range(300):
        for h in hs:
            h.iterate()
        _same(hs[0], hs[1], ("xmean", "sigma", "csep", "D", "pc", "ps", "arx", "fit_val",
                             "fit_idx"), "sep n=%d it=%d" % (n, it))
        fo, fr = hs[0].converged(), hs[1].converged()
        assert fo == fr
        if fo:
            break


def test_cma_bound_and_lazy_eigen(oracle_lib, ref_lib):
    """bound=True clipping (cmaes.cpp:74-77,93-95) and the lazy eigen schedule of plain CMAES
    at small lambda (cmaes.cpp:48,233)"""
    n, lam = 20, 8
    oracle_lib.seed(2)
    ref_lib.seed(2)
    lo, up = -1. * np.ones(n), 2. * np.ones(n)
    guess = 1.9 * np.ones(n)
    o = po.cma(oracle_lib, "cmaes", 40000, 1e-10, lam, sigma0=1., bound=True)
    r = po.cma(ref_lib, "cmaes", 40000, 1e-10, lam, sigma0=1., bound=True)
    o.init("rosenbrock", lo, up, guess)
    r.init("rosenbrock", lo, up, guess)
    skipped = 0
    for it in range(120):
        o.iterate()
        r.iterate()
        _same(o, r, ("xmean", "sigma", "C", "B", "D", "arx", "eigenlastev"), "it=%d" % it)
        skipped += int(o.scalar("eigen_done") == 0)
    assert skipped > 0
    assert o.get("arx").max() <= 2. and o.get("arx").min() >= -1.


@pytest.mark.parametrize("algo,kw,keys", [
    ("shade", dict(mfev=6000, npinit=30, tol=1e-8), ("x", "f", "arch", "MCR", "MF", "k", "np", "fev")),
    ("shade", dict(mfev=3000, npinit=20, tol=1e-8, archive=False, repaircr=False, h=5, npmin=6),
     ("x", "f", "MCR", "MF", "k", "np", "fev")),
    ("jade", dict(mfev=6000, np_=25, tol=1e-8), ("x", "f", "arch", "mucr", "muf", "fev")),
    ("jade", dict(mfev=3000, np_=20, tol=1e-8, archive=False, repaircr=False, pelite=0.2),
     ("x", "f", "mucr", "muf", "fev")),
    ("apso", dict(mfev=6000, tol=1e-8, np_=15), ("x", "v", "xb", "f", "fb", "xbest", "fbest", "w",
                                                 "c1", "c2", "state", "it", "fev")),
    ("apso", dict(mfev=6000, tol=1e-8, np_=10, correct=False), ("x", "v", "f", "xbest", "fbest",
                                                                "state", "fev")),
])
@pytest.mark.parametrize("obj", ["rastrigin", "rosenbrock", "griewank"])
def test_de_pso_bit_exact(oracle_lib, ref_lib, algo, kw, keys, obj):
    n = 9
    oracle_lib.seed(33)
    ref_lib.seed(33)
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    o = getattr(po, algo)(oracle_lib, **kw)
    r = getattr(po, algo)(ref_lib, **kw)
    o.init(obj, lo, up, np.zeros(n))
    r.init(obj, lo, up, np.zeros(n))
    _same(o, r, keys, "%s init" % algo)
    for it in range(100):
        o.iterate()
        r.iterate()
        if algo == "apso" and not (0 <= r.scalar("state") <= 4):
            break   # the reference indexed past its rule table (apso.cpp:384): undefined
        _same(o, r, keys, "%s it=%d" % (algo, it))
        if r.scalar("fev") >= kw["mfev"]:
            break
    xo, fo, co = o.solution()
    xr, fr, cr = r.solution()
    if algo != "apso" or 0 <= r.scalar("state") <= 4:
        np.testing.assert_array_equal(xo, xr)
        assert (fo, co) == (fr, cr)


@pytest.mark.parametrize("n,npp,obj,kw", [
    (8, 16, "rastrigin", {}),
    (13, 24, "rosenbrock", dict(repaircr=False, crref=3, pupdate=7, crupdate=5)),
    (5, 10, "sphere", dict(pupdate=10, crupdate=4)),
    (21, 40, "griewank", dict(crref=1))])
def test_sansde_bit_exact(oracle_lib, ref_lib, n, npp, obj, kw):
    """SaNSDESearch, every generation: swarm, per-individual CR and all adaptation counters"""
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    hs = []
    for L in (oracle_lib, ref_lib):
        L.seed(31)
        h = po.sansde(L, 200000, npp, 1e-9, **kw)
        h.init(obj, lo, up, np.zeros(n))
        hs.append(h)
    for it in range(200):
        for h in hs:
            h.iterate()
        for k in ("x", "f", "cr", "p", "fp", "crm", "crrec", "crdeltaf", "pns", "pnf", "fpns",
                  "fpnf", "fev", "it"):
            np.testing.assert_array_equal(hs[0].get(k), hs[1].get(k),
                                          err_msg="sansde n=%d it=%d %s" % (n, it, k))
    (xa, fa, ca), (xb, fb, cb) = hs[0].solution(), hs[1].solution()
    np.testing.assert_array_equal(xa, xb)
    assert (fa, ca) == (fb, cb)


@pytest.mark.parametrize("n,npp,obj,kw", [
    (8, 12, "rastrigin", {}),
    (8, 13, "rosenbrock", dict(pcompete=2)),
    (5, 30, "sphere", dict(pcompete=4, ring=True)),
    (11, 200, "ackley", dict(pcompete=2, ring=True, correct=False, vmax=0.1)),
    (6, 500, "sphere", dict(pcompete=5)),
    (4, 70000, "sphere", dict(pcompete=2))])      # np^2 > 2^32: one swap per draw in std::shuffle
def test_cso_bit_exact(oracle_lib, ref_lib, n, npp, obj, kw):
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    hs = []
    for L in (oracle_lib, ref_lib):
        L.seed(57)
        h = po.cso(L, 10 ** 8, 1e-9, npp, **kw)
        h.init(obj, lo, up, np.zeros(n))
        hs.append(h)
    keys = ["x", "v", "f", "mean", "meanw", "xbest", "fbest", "fev", "phil", "phih"] + \
        (["pmean", "home"] if kw.get("ring") else [])
    for it in range(3 if npp > 10000 else 40):
        for h in hs:
            h.iterate()
        for k in keys:
            np.testing.assert_array_equal(hs[0].get(k), hs[1].get(k),
                                          err_msg="cso n=%d it=%d %s" % (n, it, k))


@pytest.mark.parametrize("n,npp,obj,pps,kw", [
    (12, 8, "rastrigin", [2, 3, 6], {}),
    (20, 10, "rosenbrock", [5, 10], dict(correct=False)),
    (30, 12, "ackley", [5], {}),
    (64, 20, "ellipsoid", [4, 8, 16, 32], {})])
def test_ccpso_bit_exact(oracle_lib, ref_lib, capfd, n, npp, obj, pps, kw):
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    hs = []
    for L in (oracle_lib, ref_lib):
        L.seed(91)
        h = po.ccpso(L, 10 ** 8, 1e-9, npp, pps, **kw)
        h.init(obj, lo, up, np.zeros(n))
        hs.append(h)
    for it in range(40):
        for h in hs:
            h.iterate()
        for k in ("x", "y", "yhat", "fx", "fy", "k", "ibest", "strat", "fyhat", "phat", "fev", "is",
                  "nswarm", "cpswarm", "improved"):
            np.testing.assert_array_equal(hs[0].get(k), hs[1].get(k),
                                          err_msg="ccpso n=%d it=%d %s" % (n, it, k))
    capfd.readouterr()      # the reference prints _fyhat every generation (ccpso.cpp:121)


@pytest.mark.parametrize("n,npp,obj,cps,variant,lf", [
    (24, 8, "rosenbrock", 4, "cmaes", 3),
    (20, 10, "rastrigin", 5, "active", 2),
    (30, 12, "ellipsoid", 6, "cmaes", 1),
    (16, 6, "griewank", 2, "active", 5)])
def test_ccpso_with_local_optimizer_bit_exact(oracle_lib, ref_lib, capfd, n, npp, obj, cps,
                                              variant, lf):
    """CCPSO's localSearch (ccpso.cpp:371-435) with a CMA-ES variant as `local`, one swarm size
    (the reference's Cmaes::init has undefined behaviour when the dimension grows between two
    optimize() calls on the same object, cmaes.cpp:53-54)"""
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    hs = []
    for L in (oracle_lib, ref_lib):
        L.seed(17)
        h = po.ccpso(L, 10 ** 8, 1e-9, npp, [cps], local=po.cma(L, variant, 300, 1e-8, 8),
                     localfreq=lf)
        h.init(obj, lo, up, np.zeros(n))
        hs.append(h)
    for it in range(25):
        for h in hs:
            h.iterate()
        for k in ("x", "y", "yhat", "fx", "fy", "k", "ibest", "strat", "fyhat", "phat", "fev", "is",
                  "nswarm", "cpswarm", "improved"):
            np.testing.assert_array_equal(hs[0].get(k), hs[1].get(k),
                                          err_msg="ccpso+local n=%d it=%d %s" % (n, it, k))
    capfd.readouterr()


@pytest.mark.parametrize("driver", ["bipop", "ipop"])
@pytest.mark.parametrize("variant", ["active", "cmaes"])
def test_restart_drivers_bit_exact(oracle_lib, ref_lib, driver, variant):
    n = 5
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(8).uniform(-5, 5, n)
    oracle_lib.seed(44)
    ref_lib.seed(44)
    o = getattr(po, driver)(oracle_lib, po.cma(oracle_lib, variant, 1, 1e-6, 4), 40000)
    r = getattr(po, driver)(ref_lib, po.cma(ref_lib, variant, 1, 1e-6, 4), 40000)
    xo, fo, _ = o.optimize("rastrigin", lo, up, guess)
    xr, fr, _ = r.optimize("rastrigin", lo, up, guess)
    np.testing.assert_array_equal(xo, xr)
    assert fo == fr
