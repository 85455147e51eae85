"""GPU parity: the fused L-SHADE / JADE generation against the oracle's generation-synchronous
restatement (De::iterate_sync) fed by the same Philox draws.

Everything except tan / log / sincos is element-wise IEEE arithmetic built with
-ffp-contract=off on both sides, so positions agree to a few ulp; the tolerances below leave
room for the libm differences (device tan vs glibc tan in F, Box-Muller in CR) and for the
tree-ordered sums of the success memories.
"""
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


def _pair(hip, oracle_lib, algo, n, np_, obj, seed, **kw):
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    if algo == "shade":
        g = hip.SHADE(seed=seed, **kw)
        okw = dict(kw)
        o = po.shade(oracle_lib, **okw)
    else:
        g = hip.JADE(seed=seed, **kw)
        okw = dict(kw)
        okw["np_"] = okw.pop("np")
        o = po.jade(oracle_lib, **okw)
    o.set_mode(True, po.RNG_PHILOX, seed)
    g.initialize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o.init(obj, lo, up, np.zeros(n))
    return g, o


def _compare(g, o, algo, tag):
    assert int(g.get_state("np")[0]) == int(o.scalar("np")), tag
    assert int(g.get_state("fev")[0]) == int(o.scalar("fev")), tag
    _close(g.get_state("f"), o.get("f"), 1e-11, tag + " f")
    _close(g.get_state("x"), o.get("x"), 1e-12, tag + " x")
    assert int(g.get_state("larch")[0]) == int(o.scalar("larch")), tag
    _close(g.get_state("arch"), o.get("arch"), 1e-12, tag + " archive")
    if algo == "shade":
        _close(g.get_state("MCR"), o.get("MCR"), 1e-10, tag + " MCR")
        _close(g.get_state("MF"), o.get("MF"), 1e-10, tag + " MF")
        assert int(g.get_state("k")[0]) == int(o.scalar("k")), tag
    else:
        _close(g.get_state("mucr"), o.get("mucr"), 1e-10, tag + " mucr")
        _close(g.get_state("muf"), o.get("muf"), 1e-10, tag + " muf")


@pytest.mark.parametrize("algo,n,kw,obj", [
    ("shade", 8, dict(mfev=100000, npinit=16, tol=1e-12), "rastrigin"),
    ("shade", 33, dict(mfev=100000, npinit=50, tol=1e-12), "rosenbrock"),      # odd n, ragged np
    ("shade", 16, dict(mfev=2500, npinit=64, tol=1e-12, npmin=8), "sphere"),   # LPSR + archive trim
    ("shade", 12, dict(mfev=100000, npinit=20, tol=1e-12, archive=False, repaircr=False, h=3),
     "ackley"),
    ("jade", 8, dict(mfev=100000, np=16, tol=1e-12), "rastrigin"),
    ("jade", 21, dict(mfev=100000, np=40, tol=1e-12, pelite=0.2, archive=False), "griewank"),
    ("shade", 150, dict(mfev=100000, npinit=24, tol=1e-12), "ellipsoid"),   # > 128 columns: the
    ("jade", 301, dict(mfev=100000, np=20, tol=1e-12), "sphere"),           # loop's second pass
    # rows of more than 512 doubles: 8 / 4 individuals per workgroup (rows_per_wg16)
    ("shade", 513, dict(mfev=100000, npinit=24, tol=1e-12), "rastrigin"),
    ("jade", 1024, dict(mfev=100000, np=20, tol=1e-12), "rosenbrock"),
    ("shade", 2048, dict(mfev=100000, npinit=18, tol=1e-12), "sphere"),
])
def test_generations_match_sync_oracle(hip, oracle_lib, algo, n, kw, obj):
    g, o = _pair(hip, oracle_lib, algo, n, kw.get("npinit", kw.get("np")), obj, 99, **kw)
    # identical initial population: same Philox words, same affine map -> bit-exact
    np.testing.assert_array_equal(np.sort(g.get_state("x"), axis=None),
                                  np.sort(o.get("x"), axis=None))
    _compare(g, o, algo, "init")
    for gen in range(25 if n <= 512 else 6):
        g.iterate()
        o.iterate()
        _compare(g, o, algo, "gen %d" % gen)
        if o.scalar("fev") >= kw["mfev"]:
            break


def test_c2_full_size_generations_match_sync_oracle(hip, oracle_lib):
    """C2 itself: L-SHADE n = 128, np = 4096, Rastrigin, three generations against the
    generation-synchronous oracle (~25 ms per generation there)"""
    n, kw = 128, dict(mfev=10 ** 8, npinit=4096, tol=1e-12)
    g, o = _pair(hip, oracle_lib, "shade", n, 4096, "rastrigin", 2024, **kw)
    _compare(g, o, "shade", "init")
    for gen in range(3):
        g.iterate()
        o.iterate()
        _compare(g, o, "shade", "gen %d" % gen)


def test_shade_solves_sphere_and_stops(hip):
    n = 16
    alg = hip.SHADE(mfev=200000, npinit=100, tol=1e-6, seed=5)
    sol = alg.optimize(hip.objectives.sphere, -10 * np.ones(n), 10 * np.ones(n), np.zeros(n))
    assert sol.converged and np.abs(sol.x).max() < 1e-3
    assert sol.n_evals < 200000


def test_jade_solves_rosenbrock(hip):
    n = 10
    alg = hip.JADE(mfev=100000, np=50, tol=1e-8, seed=6)
    sol = alg.optimize(hip.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n), np.zeros(n))
    assert hip.objectives.rosenbrock(sol.x) < 1e-6


def test_python_callback_objective(hip):
    """the host-callback path (multivariate_py.cpp:385-388): same optimizer, Python f"""
    n = 6
    calls = []

    def f(x):
        calls.append(1)
        return float(np.sum((x - 0.5) ** 2))

    alg = hip.SHADE(mfev=6000, npinit=30, tol=1e-9, seed=7)
    sol = alg.optimize(f, -2 * np.ones(n), 2 * np.ones(n), np.zeros(n))
    assert np.abs(sol.x - 0.5).max() < 1e-3
    assert len(calls) == sol.n_evals


@pytest.mark.parametrize("algo", ["shade", "jade"])
@pytest.mark.parametrize("obj,seed", [("sphere", 1), ("ellipsoid", 2)])
def test_whole_run_same_seed_matches_oracle(hip, oracle_lib, algo, obj, seed):
    """optimize() to the algorithm's own stop (radius-spread test, L-SHADE's population
    reduction on the way) on the device and by the generation-synchronous oracle drawing the same
    Philox numbers: the same number of evaluations, the same x* (smooth objectives: on Rosenbrock
    a last-bit difference in f flips a `<=` selection late in the run and the counts differ by
    a few dozen evaluations, with the same optimum)."""
    n = 10
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    if algo == "shade":
        g = hip.SHADE(mfev=60000, npinit=40, tol=1e-8, seed=seed)
        o = po.shade(oracle_lib, 60000, 40, 1e-8)
    else:
        g = hip.JADE(mfev=60000, np=30, tol=1e-8, seed=seed)
        o = po.jade(oracle_lib, 60000, 30, 1e-8)
    o.set_mode(True, po.RNG_PHILOX, seed)
    sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    xo, fevo, convo = o.optimize(obj, lo, up, np.zeros(n))
    assert sol.converged and convo
    assert sol.n_evals == fevo
    np.testing.assert_allclose(sol.x, xo, rtol=0, atol=1e-10)
