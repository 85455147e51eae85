"""GPU <-> REFERENCE directly, no oracle in between (tests/golden/cma_runs.json was written by
the compiled reference, oracle/gen_golden.py): the reference's own normals of its first three
generations are injected into the device (bbo_cma_inject_normals) and the device's state after
each generation is held against the reference's recorded state.  All golden runs have n <= 16,
where the device's eigensolver is the reference's tql2 with its sign conventions, so B and with
it x = m + sigma B D z are comparable from generation 2 on as well.

Second half: crafted states for every stop test of Cmaes::converged (cmaes.cpp:151-227), the
device's cma_history_stop against the oracle's converged() and against the flag the state was
built to raise.

Tolerance: 1e-10 relative to the largest entry (generation 1: B = I, pure GEMM/reduction
rounding; generations 2-3 go through an eigendecomposition of C = I + O(0.1))."""
import numpy as np
import pytest

import pyoracle as po
from _golden import load, unhex

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / scale
    assert err <= rtol, "%s: rel err %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("idx", range(5))
def test_injected_reference_normals_reproduce_reference_states(hip, idx):
    rec = load("cma_runs.json")[idx]
    n, lam, box = rec["n"], rec["lambda"], rec["box"]
    cls = hip.ActiveCMAES if rec["variant"] == "active" else hip.CMAES
    g = cls(mfev=rec["mfev"], tol=rec["tol"], np=lam, seed=1)
    g.initialize(getattr(hip.objectives, rec["objective"]), -box * np.ones(n), box * np.ones(n),
                 unhex(rec["guess"]))
    zs = unhex(rec["normals_first3"]).reshape(3, lam * n)
    states = {s["gen"]: s for s in rec["states"]}
    for gen in (1, 2, 3):
        g.inject_normals(zs[gen - 1])
        g.iterate()
        st = states[gen]
        tol = 1e-11 if gen == 1 else 1e-10
        for key in ("arx", "xmean", "sigma", "pc", "ps", "D"):
            _close(g.get_state(key), unhex(st[key]), tol, "gen %d %s" % (gen, key))
        _close(g.get_state("fit_val"), unhex(st["fit_val"]), 1e-10, "gen %d fit_val" % gen)
        np.testing.assert_array_equal(g.get_state("fit_idx"), unhex(st["fit_idx"]))
        Cg = np.tril(g.get_state("C").reshape(n, n))
        Cr = np.tril(unhex(st["C"]).reshape(n, n))      # the reference maintains the lower half
        _close(Cg, Cr, tol, "gen %d C" % gen)
        _close(g.get_state("invsqrtC"), unhex(st["invsqrtC"]), 1e-9, "gen %d invsqrtC" % gen)
        assert int(g.get_state("it")[0]) == int(unhex(st["it"])[0])
        assert int(g.get_state("fev")[0]) == int(unhex(st["fev"])[0])
        # B itself is comparable only while the eigenvalues are simple: with lambda < n the first
        # covariance matrices are I + (rank < n) and the basis of the repeated eigenvalue is
        # decided by rounding inside the eigensolver -- the next generation's x = m + sigma B D z
        # then differs legitimately, and the comparison ends here
        ev = unhex(st["D"]) ** 2
        if np.diff(ev).min() <= 1e-6 * ev.max():
            assert gen >= 1
            break
        _close(g.get_state("B"), unhex(st["B"]), 1e-8, "gen %d B" % gen)
    g.inject_normals(None)


def _crafted(hip, oracle_lib, variant, n, lam, tol, mfev):
    """one real generation on both sides (so f / order / history hold real values), device
    normals fed to the oracle"""
    from bboptpy_amd import _ffi
    cls = hip.ActiveCMAES if variant == "active" else hip.CMAES
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(n).uniform(-3, 3, n)
    g = cls(mfev=mfev, tol=tol, np=lam, seed=3)
    g.initialize(hip.objectives.ellipsoid, lo, up, guess)
    g.set_state("record_normals", [1.0])
    o = po.cma(oracle_lib, variant, mfev, tol, lam)
    o.set_rng(po.RNG_INJECT)
    o.init("ellipsoid", lo, up, guess)
    g.phase(_ffi.PHASE_SAMPLE_EVALUATE)
    o.inject_z(g.get_state("zlast"))
    o.step("sample")
    o.step("evaluate_sort")
    g.phase(_ffi.PHASE_RANK)
    g.phase(_ffi.PHASE_UPDATE)
    g.phase(_ffi.PHASE_EIGEN)
    o.step("update_distribution")
    # identical starting point for the crafted part: the oracle takes the device's state
    for key in ("xmean", "pc", "ps", "C", "B", "D", "invsqrtC", "sigma"):
        o.set(key, g.get_state(key))
    return g, o


def _both(g, o, key, value):
    g.set_state(key, value)
    o.set(key, value)


def _stop_flags(g, o):
    from bboptpy_amd import _ffi
    g.phase(_ffi.PHASE_HISTORY_STOP)
    o.step("update_history")
    assert int(g.get_state("it")[0]) == int(o.scalar("it"))
    return int(g.get_state("flag")[0]), o.converged()


@pytest.mark.parametrize("variant", ["active", "cmaes"])
def test_crafted_stop_states_raise_every_flag(hip, oracle_lib, variant):
    n, lam, tol, mfev = 6, 12, 1e-12, 12 * 1000
    hlen = 10 + int(np.ceil(30. * n / lam))
    fresh = lambda: _crafted(hip, oracle_lib, variant, n, lam, tol, mfev)

    # no crafted state: the run goes on
    g, o = fresh()
    assert _stop_flags(g, o) == (0, 0)

    # 1 MaxIter (cmaes.cpp:154): it reaches mit = mfev / lambda
    g, o = fresh()
    _both(g, o, "it", [mfev // lam - 1])
    assert _stop_flags(g, o) == (1, 1)
    assert int(g.get_state("stop")[0]) == 1

    # 2 TolHistFun (:160): a full history of best values within tol of each other
    g, o = fresh()
    f0 = g.get_state("fit_val")[0]
    ring = f0 + 1e-14 * np.arange(hlen)
    g.set_state("best_hist", ring); o.set("best_hist", ring)
    g.set_state("kth_hist", ring + 1.); o.set("kth_hist", ring + 1.)
    _both(g, o, "best_len", [hlen])
    _both(g, o, "best_buffer", [hlen - 1])
    _both(g, o, "it", [hlen + 3])
    assert _stop_flags(g, o) == (2, 2)

    # 3 EqualFunVals (:166-177): best == k-th best in at least n/3 of the last n generations
    g, o = fresh()
    ring = 100. + 7. * np.arange(hlen)          # spread >> tol: TolHistFun stays quiet
    g.set_state("best_hist", ring); o.set("best_hist", ring)
    g.set_state("kth_hist", ring); o.set("kth_hist", ring)
    _both(g, o, "best_len", [hlen])
    _both(g, o, "best_buffer", [4])
    _both(g, o, "it", [hlen + 3])
    assert _stop_flags(g, o) == (3, 3)

    # 4 TolX (:180-190): sigma so small that every coordinate's step is below tol
    g, o = fresh()
    _both(g, o, "sigma", [1e-20])
    assert _stop_flags(g, o) == (4, 4)

    # 5 TolUpSigma (:193): sigma / sigma0 > 1e20 D_max
    g, o = fresh()
    _both(g, o, "sigma", [1e25])
    assert _stop_flags(g, o) == (5, 5)

    # 7 ConditionCov (:199): D_max > 1e7 D_min
    g, o = fresh()
    D = np.logspace(-8.5, 0, n)
    g.set_state("D", D); o.set("D", D)
    assert _stop_flags(g, o) == (7, 7)

    # 8 NoEffectAxis (:205-217): 0.1 sigma D_axis b_axis vanishes against the mean
    g, o = fresh()
    _both(g, o, "xmean", 1e10 * np.ones(n))
    _both(g, o, "sigma", [1e-8])
    _both(g, o, "pc", np.ones(n))                 # keeps TolX quiet: pc sigma / sigma0 >= tol
    assert _stop_flags(g, o) == (8, 8)

    # 9 NoEffectCoor (:220-225): one coordinate of the mean does not feel 0.2 sigma sqrt(C_ii),
    # while the axis of this generation (not that coordinate's) still moves the mean
    g, o = fresh()
    eye = np.eye(n)
    g.set_state("B", eye); o.set("B", eye.ravel())
    ones = np.ones(n)
    g.set_state("D", ones); o.set("D", ones)
    o.set("invsqrtC", eye.ravel())
    g.set_state("C", eye); o.set("C", eye.ravel())
    it_now = int(g.get_state("it")[0])           # the phase increments it before the tests
    iaxis = n - 1 - (it_now % n)
    stuck = (iaxis + 1) % n
    xm = np.ones(n)
    xm[stuck] = 1e10
    _both(g, o, "xmean", xm)
    _both(g, o, "sigma", [1e-8])
    _both(g, o, "pc", np.ones(n))
    assert _stop_flags(g, o) == (9, 9)
