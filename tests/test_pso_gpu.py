"""GPU parity: the APSO generation against the oracle's generation-synchronous restatement
(Apso with sync = True) fed by the same Philox draws.

The evolutionary factor is the one quantity computed differently: the device forms the
all-pairs distances from a Gram matrix of the CENTRED swarm on the matrix cores, the oracle
subtracts coordinates like apso.cpp:316-321 -- tolerance 1e-9 on f, and the fuzzy state must
be identical.
"""
import numpy as np
import pytest

import pyoracle as po

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, what):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, what
    err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
    assert err <= rtol, "%s: rel err %.3e > %.1e" % (what, err, rtol)


@pytest.mark.parametrize("n,np_,obj,correct", [
    (8, 12, "rastrigin", True),
    (8, 12, "rosenbrock", True),
    (33, 70, "sphere", True),         # odd n, np not a multiple of 16 or 64
    (16, 64, "ackley", False),
    (64, 130, "griewank", True),
    (24, 1100, "rosenbrock", True),   # 16 refreshes of the swarm's best per generation (chunks of 80)
    # rows of more than 512 doubles: 8 / 4 particles per workgroup (rows_per_wg16)
    (513, 40, "sphere", True),
    (1024, 24, "rastrigin", True),
    (2048, 20, "sphere", False),
])
def test_generations_match_sync_oracle(hip, oracle_lib, n, np_, obj, correct):
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    seed = 4242
    g = hip.APSO(mfev=10 ** 7, tol=1e-12, np=np_, correct=correct, seed=seed)
    o = po.apso(oracle_lib, 10 ** 7, 1e-12, np_, correct)
    o.set_mode(True, po.RNG_PHILOX, seed)
    g.initialize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o.init(obj, lo, up, np.zeros(n))
    # (the swarm moves in chunks with the best refreshed in between: the oracle is told the chunk)
    chunk = int(g.get_state("chunk")[0])
    assert chunk == np_ if np_ <= 64 else (chunk % 16 == 0 and np_ / 16 <= chunk < np_)
    o.set_chunk(chunk)
    np.testing.assert_array_equal(g.get_state("x"), o.get("x"))   # same Philox words
    _close(g.get_state("f"), o.get("f"), 1e-12, "init f")
    _close(g.get_state("fbest"), o.get("fbest"), 1e-12, "init fbest")
    states = []
    for gen in range(40 if n <= 512 else 6):
        g.iterate()
        o.iterate()
        tag = "gen %d" % gen
        _close(g.get_state("evof"), o.get("evof"), 1e-9, tag + " evolutionary factor")
        assert int(g.get_state("state")[0]) == int(o.scalar("state")), tag
        states.append(int(o.scalar("state")))
        for k, tol in (("w", 1e-10), ("c1", 1e-12), ("c2", 1e-12)):
            _close(g.get_state(k), o.get(k), tol, tag + " " + k)
        assert int(g.get_state("fev")[0]) == int(o.scalar("fev")), tag
        _close(g.get_state("x"), o.get("x"), 1e-11, tag + " x")
        _close(g.get_state("v"), o.get("v"), 1e-10, tag + " v")
        _close(g.get_state("xb"), o.get("xb"), 1e-11, tag + " pbest")
        _close(g.get_state("f"), o.get("f"), 1e-10, tag + " f")
        _close(g.get_state("xbest"), o.get("xbest"), 1e-11, tag + " gbest")
        _close(g.get_state("fbest"), o.get("fbest"), 1e-10, tag + " fbest")
    if n <= 512 and np_ < 1000:
        assert len(set(states)) >= 2   # the fuzzy state machine actually moved


@pytest.mark.parametrize("n,np_", [
    (8, 12),          # one block, mostly padding
    (40, 300),        # NB = 3 (odd): every block sweeps one partner, the last block is partial
    (130, 700),       # NB = 6 (even): the opposite pairs belong to the smaller index; n % 16 != 0
    (20, 1024),       # NB = 8, all blocks full: the test-free epilogue of pso_ese_sym
    (16, 1152),       # NB = 9, all blocks full
    (515, 260),       # rows longer than 512 doubles (33 staged chunks), NB = 3
])
def test_mean_distance_matches_numpy(hip, n, np_):
    """getf's d_i = mean_j ||x_i - x_j|| (apso.cpp:300-339) from pso_nrm + pso_ese_sym +
    pso_ese_finish against the plain double loop, at block counts that exercise the cyclic pair
    cover (odd / even NB), partial blocks and the unguarded tile path."""
    alg = hip.APSO(mfev=10 ** 9, tol=0., np=np_, seed=5)
    alg.initialize(hip.objectives.sphere, -5. * np.ones(n), 5. * np.ones(n), np.zeros(n))
    X0 = alg.get_state("x").reshape(np_, n).copy()
    alg.iterate()
    ws = alg.get_state("ws")
    Xc = X0 - X0.mean(0)
    Gc = Xc @ Xc.T
    sqc = np.diag(Gc)
    D = np.sqrt(np.maximum(sqc[:, None] + sqc[None, :] - 2. * Gc, 0.))
    np.fill_diagonal(D, 0.)
    # (the centred Gram form loses ~1e-13 relative next to the direct differences; hold the
    # device against the direct form on a sample of rows and against the Gram form on all)
    want = D.sum(1) / (np_ - 1.)
    assert np.abs(ws - want).max() <= 1e-11 * want.max()
    rows = np.random.default_rng(1).choice(np_, size=min(np_, 16), replace=False)
    direct = np.array([np.sqrt(((X0[i] - X0) ** 2).sum(1)).sum() / (np_ - 1.) for i in rows])
    assert np.abs(ws[rows] - direct).max() <= 1e-11 * direct.max()


def test_chunk_of_the_whole_swarm_is_the_synchronous_form(hip, oracle_lib):
    """bbo_set "chunk" 0: every particle sees the best of the generation start (rounds 1-4); the
    oracle's sync mode without a chunk is that form"""
    n, np_, seed = 12, 200, 7
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.APSO(mfev=10 ** 7, tol=1e-12, np=np_, seed=seed)
    o = po.apso(oracle_lib, 10 ** 7, 1e-12, np_)
    o.set_mode(True, po.RNG_PHILOX, seed)
    g.initialize(hip.objectives.rosenbrock, lo, up, np.zeros(n))
    g.set_state("chunk", [0.])
    assert int(g.get_state("chunk")[0]) == np_
    o.init("rosenbrock", lo, up, np.zeros(n))
    for gen in range(10):
        g.iterate()
        o.iterate()
        _close(g.get_state("x"), o.get("x"), 1e-11, "gen %d x" % gen)
        _close(g.get_state("xbest"), o.get("xbest"), 1e-11, "gen %d gbest" % gen)


@pytest.mark.parametrize("np_", [30, 64, 65, 200, 1024, 5000, 40000])
def test_bench_prices_a_launch_with_the_engines_chunk(hip, np_):
    """bench.py prices ONE pso_update launch with the chunk of particles it moves: its rule and
    the engine's must be the same"""
    import bench
    n = 4
    g = hip.APSO(mfev=10 ** 7, tol=1e-12, np=np_, seed=1)
    g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
    assert int(g.get_state("chunk")[0]) == bench.pso_chunk(np_)


def test_python_objective_sees_the_same_chunks(hip):
    """a Python callable and the built-in objective of the same function: the same swarm after
    every generation (the host path evaluates chunk by chunk, the best refreshed in between)"""
    n, np_, seed = 6, 150, 3
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    a = hip.APSO(mfev=10 ** 7, tol=1e-12, np=np_, seed=seed)
    b = hip.APSO(mfev=10 ** 7, tol=1e-12, np=np_, seed=seed)
    a.initialize(hip.objectives.sphere, lo, up, np.zeros(n))
    b.initialize(lambda x: float(np.sum(np.asarray(x) ** 2)), lo, up, np.zeros(n))
    assert int(a.get_state("chunk")[0]) == int(b.get_state("chunk")[0]) == 64   # three chunks: 64, 64, 22
    for gen in range(8):
        a.iterate()
        b.iterate()
        _close(a.get_state("x"), b.get_state("x"), 1e-12, "gen %d x" % gen)
        _close(a.get_state("xbest"), b.get_state("xbest"), 1e-12, "gen %d gbest" % gen)
        assert int(a.get_state("fev")[0]) == int(b.get_state("fev")[0])


def test_apso_solves_sphere(hip):
    n = 8
    alg = hip.APSO(mfev=200000, tol=1e-6, np=40, seed=3)
    sol = alg.optimize(hip.objectives.sphere, -10 * np.ones(n), 10 * np.ones(n), np.zeros(n))
    assert hip.objectives.sphere(sol.x) < 1e-4


def test_apso_python_callback(hip):
    n = 5

    def f(x):
        return float(np.sum((x + 1.0) ** 2))

    alg = hip.APSO(mfev=8000, tol=1e-7, np=20, seed=9)
    sol = alg.optimize(f, -4 * np.ones(n), 4 * np.ones(n), np.zeros(n))
    assert np.abs(sol.x + 1.0).max() < 0.05


@pytest.mark.parametrize("obj,seed", [("rosenbrock", 1), ("rosenbrock", 2)])
def test_whole_run_same_seed_matches_oracle(hip, oracle_lib, obj, seed):
    """APSO run to its evaluation budget on the device and by the synchronous oracle with the
    same Philox numbers: same evaluation count (incl. the elitist-learning extras), same best
    point"""
    n = 10
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    g = hip.APSO(mfev=60000, tol=1e-8, np=30, seed=seed)
    sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
    o = po.apso(oracle_lib, 60000, 1e-8, 30)
    o.set_mode(True, po.RNG_PHILOX, seed)
    o.set_chunk(30)
    xo, fevo, convo = o.optimize(obj, lo, up, np.zeros(n))
    assert sol.n_evals == fevo and sol.converged == convo
    np.testing.assert_allclose(sol.x, xo, rtol=0, atol=1e-8)
