"""GPU parity: one CSO generation (means, shuffle, group sort, winners' mean, the losers'
learning step, incumbent) against the oracle in generation-synchronous Philox mode.  The oracle's
reference-mode form is pinned bit for bit to the compiled reference (std::shuffle replica and the
birth-slot ring neighbourhood included); the synchronous form differs only in where the random
numbers come from (Philox words keyed by slot / coordinate / generation) and in shuffling with a
keyed Feistel bijection of the slots instead of std::shuffle."""
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
    (8, 12, "rastrigin", {}),
    (33, 50, "rosenbrock", dict(pcompete=2)),                       # odd n, np rounded up? (50 % 2 == 0)
    (7, 31, "sphere", dict(pcompete=4, ring=True)),                 # np rounded up to 32, ring
    (12, 202, "ackley", dict(pcompete=2, ring=True, correct=False, vmax=0.1)),   # np > 100: phi > 0
    (16, 9000, "sphere", dict(pcompete=3)),                         # a large swarm
    (200, 45, "rosenbrock", dict(pcompete=3)),                      # 32 lanes per group (fused swarm mean)
    (301, 30, "ellipsoid", dict(pcompete=3)),                       # 64 lanes per group
    (513, 30, "sphere", dict(pcompete=3)),                          # > 512 columns: 8 groups per workgroup
    (1024, 24, "rastrigin", dict(pcompete=2, ring=True)),
])
def test_generations_match_sync_oracle(hip, oracle_lib, n, npp, obj, kw):
    seed = 321
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.CSO(mfev=10 ** 8, stol=1e-12, np=npp, seed=seed, **kw)
    o = po.cso(oracle_lib, 10 ** 8, 1e-12, npp, **kw)
    o.set_mode(True, po.RNG_PHILOX, seed)
    g.initialize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o.init(obj, lo, up, np.zeros(n))
    assert int(g.get_state("np")[0]) == int(o.scalar("np"))
    np.testing.assert_array_equal(g.get_state("x"), o.get("x"))     # same Philox words, same map
    for gen in range(12 if npp > 1000 else 25):
        g.iterate()
        o.iterate()
        tag = "gen %d" % gen
        np.testing.assert_array_equal(g.get_state("home"), o.get("home"), err_msg=tag + " order")
        _close(g.get_state("x"), o.get("x"), 1e-12, tag + " x")
        _close(g.get_state("v"), o.get("v"), 1e-11, tag + " v")
        _close(g.get_state("f"), o.get("f"), 1e-11, tag + " f")
        _close(g.get_state("meanw"), o.get("meanw"), 1e-12, tag + " meanw")
        if kw.get("ring"):
            _close(g.get_state("pmean"), o.get("pmean"), 1e-13, tag + " pmean")
        else:
            _close(g.get_state("mean"), o.get("mean"), 1e-12, tag + " mean")
        _close(g.get_state("xbest"), o.get("xbest"), 1e-12, tag + " xbest")
        assert int(g.get_state("fev")[0]) == int(o.scalar("fev"))
    assert abs(g.get_state("phil")[0] - o.get("phil")[0]) == 0
    assert abs(g.get_state("phih")[0] - o.get("phih")[0]) == 0


@pytest.mark.parametrize("obj,kw", [("sphere", {}), ("ellipsoid", dict(pcompete=2, ring=True))])
def test_whole_run_same_seed_matches_oracle(hip, oracle_lib, obj, kw):
    n = 10
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.CSO(mfev=60000, stol=1e-7, np=60, seed=4, **kw)
    sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o = po.cso(oracle_lib, 60000, 1e-7, 60, **kw)
    o.set_mode(True, po.RNG_PHILOX, 4)
    xo, fevo, convo = o.optimize(obj, lo, up, np.zeros(n))
    assert sol.n_evals == fevo and sol.converged == convo
    np.testing.assert_allclose(sol.x, xo, rtol=0, atol=1e-9)


def test_python_callback_objective(hip):
    n = 5
    calls = [0]

    def fx(x):
        calls[0] += 1
        return float(np.sum((x + 1.) ** 2))

    g = hip.CSO(mfev=8000, stol=1e-5, np=30, seed=2)
    sol = g.optimize(fx, -5. * np.ones(n), 5. * np.ones(n), np.zeros(n))
    assert calls[0] == sol.n_evals
    assert np.abs(sol.x + 1.).max() < 5e-2
