"""GPU parity: the fused SaNSDE generation against the oracle's generation-synchronous
restatement (Sansde::iterate_sync, itself derived from the reference-pinned async form) fed by
the same Philox draws.  Element-wise arithmetic is IEEE on both sides; the tolerances leave room
for device tan vs glibc tan in the Cauchy F and for tree-ordered sums of the adaptation tallies."""
import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert err <= rtol, "%s: rel err %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("n,npp,obj,kw", [
    (8, 16, "rastrigin", {}),
    (33, 50, "rosenbrock", dict(repaircr=False, crref=3, pupdate=7, crupdate=5)),   # odd n, ragged np
    (12, 20, "ackley", dict(crref=1, pupdate=4, crupdate=2)),
    (64, 256, "sphere", dict(pupdate=10, crupdate=5)),
    (201, 24, "ellipsoid", dict(pupdate=6, crupdate=3)),       # > 128 columns: second pass of the loop
    (600, 20, "sphere", dict(pupdate=6, crupdate=3)),          # > 512 columns: 8 rows per workgroup
    (2048, 16, "rosenbrock", dict(pupdate=3, crupdate=2)),     # the largest accepted n: 4 rows
])
def test_generations_match_sync_oracle(hip, oracle_lib, n, npp, obj, kw):
    seed = 123
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.SANSDE(mfev=10 ** 7, np=npp, tol=1e-12, seed=seed, **kw)
    o = po.sansde(oracle_lib, 10 ** 7, npp, 1e-12, **kw)
    o.set_mode(True, po.RNG_PHILOX, seed)
    g.initialize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o.init(obj, lo, up, np.zeros(n))
    np.testing.assert_array_equal(np.sort(g.get_state("x"), axis=None),
                                  np.sort(o.get("x"), axis=None))
    for gen in range(30):
        g.iterate()
        o.iterate()
        # the oracle keeps its swarm in last generation's order until it sorts at the start of
        # the next one; the device reports sorted order: compare as sorted sets
        og = {k: o.get(k) for k in ("x", "f", "cr")}
        idx = np.argsort(og["f"], kind="stable")
        xs = og["x"].reshape(npp, n)[idx].ravel()
        _close(g.get_state("f"), og["f"][idx], 1e-11, "gen %d f" % gen)
        _close(g.get_state("x"), xs, 1e-12, "gen %d x" % gen)
        _close(g.get_state("cr"), og["cr"][idx], 1e-12, "gen %d cr" % gen)
        assert int(g.get_state("fev")[0]) == int(o.scalar("fev"))
        np.testing.assert_array_equal(g.get_state("pns"), o.get("pns"))
        np.testing.assert_array_equal(g.get_state("pnf"), o.get("pnf"))
        for k in ("fpns", "fpnf"):
            a, b = g.get_state(k), o.get(k)
            assert np.abs(a - b).max() <= 1e-10 * max(np.abs(b).max(), 1.), (gen, k)
        for k in ("p", "fp", "crm", "crrec", "crdeltaf"):
            a, b = g.get_state(k)[0], o.scalar(k)
            assert (np.isnan(a) and np.isnan(b)) or \
                abs(a - b) <= 1e-10 * max(abs(b), 1e-300) + 1e-300, (gen, k, a, b)


@pytest.mark.parametrize("obj,seed", [("sphere", 1), ("ellipsoid", 2)])
def test_whole_run_same_seed_matches_oracle(hip, oracle_lib, obj, seed):
    n = 10
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.SANSDE(mfev=80000, np=30, tol=1e-8, seed=seed)
    sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o = po.sansde(oracle_lib, 80000, 30, 1e-8)
    o.set_mode(True, po.RNG_PHILOX, seed)
    xo, fevo, convo = o.optimize(obj, lo, up, np.zeros(n))
    assert sol.converged and convo
    assert sol.n_evals == fevo
    np.testing.assert_allclose(sol.x, xo, rtol=0, atol=1e-10)


def test_python_callback_objective(hip):
    """the host-callback path: same optimizer, a Python f"""
    n = 6
    calls = [0]

    def fx(x):
        calls[0] += 1
        return float(np.sum((x - 0.5) ** 2))

    g = hip.SANSDE(mfev=6000, np=20, tol=1e-6, seed=9)
    sol = g.optimize(fx, -5. * np.ones(n), 5. * np.ones(n), np.zeros(n))
    assert calls[0] == sol.n_evals
    assert np.abs(sol.x - 0.5).max() < 1e-2
