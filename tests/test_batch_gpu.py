"""Batched handles (`populations=P`): the populations of one handle are independent runs.
Population 0 uses Philox sub-stream 0, exactly what a single-population handle with the same
seed uses, so it must reproduce that run bit for bit while the other populations go their own
way (different sub-streams) without disturbing it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(hip, algo, P, seed):
    if algo == "active":
        return hip.ActiveCMAES(mfev=10 ** 8, tol=1e-12, np=24, seed=seed, populations=P)
    if algo == "cmaes":
        return hip.CMAES(mfev=10 ** 8, tol=1e-12, np=24, seed=seed, populations=P)
    if algo == "sep":
        return hip.SepCMAES(mfev=10 ** 8, tol=1e-12, np=24, seed=seed, populations=P)
    if algo == "shade":
        return hip.SHADE(mfev=10 ** 8, npinit=32, tol=1e-12, seed=seed, populations=P)
    if algo == "jade":
        return hip.JADE(mfev=10 ** 8, np=32, tol=1e-12, seed=seed, populations=P)
    if algo == "sansde":
        return hip.SANSDE(mfev=10 ** 8, np=32, tol=1e-12, seed=seed, populations=P)
    if algo == "ccpso":
        return hip.CCPSO(mfev=10 ** 8, sigmatol=1e-12, np=12, pps=[2, 3, 4], seed=seed,
                         populations=P)
    if algo == "cso":
        return hip.CSO(mfev=10 ** 8, stol=1e-12, np=33, seed=seed, populations=P)
    return hip.APSO(mfev=10 ** 8, tol=1e-12, np=32, seed=seed, populations=P)


@pytest.mark.parametrize("algo", ["active", "cmaes", "sep", "shade", "jade", "sansde", "apso", "cso", "ccpso"])
def test_population_zero_is_the_single_run(hip, algo):
    n, P, seed, gens = 12, 5, 77, 30
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    rng = np.random.default_rng(4)
    guess = rng.uniform(-4, 4, (P, n))
    batch = _make(hip, algo, P, seed)
    single = _make(hip, algo, 1, seed)
    batch.initialize(hip.objectives.rosenbrock, lo, up, guess)
    single.initialize(hip.objectives.rosenbrock, lo, up, guess[0])
    assert batch.run(gens) == gens and single.run(gens) == gens
    key = "xmean" if algo in ("active", "cmaes", "sep") else "x"
    np.testing.assert_array_equal(batch.get_state(key, 0), single.get_state(key))
    assert batch.get_state("fev", 0)[0] == single.get_state("fev")[0]
    s0, s1 = batch.solution(0), single.solution()
    np.testing.assert_array_equal(s0.x, s1.x)
    # the other populations are different runs, each of them sane
    seen = [batch.get_state(key, p).copy() for p in range(P)]
    for p in range(1, P):
        assert np.isfinite(seen[p]).all()
        assert not np.array_equal(seen[p], seen[0])
        assert not np.array_equal(seen[p], seen[p - 1])


def test_batch_stops_populations_independently(hip):
    """each population stops on its own flag; a stopped population is frozen while the others
    continue (device-resident stop flags)"""
    n, P = 6, 4
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(1).uniform(-4, 4, (P, n))
    g = hip.ActiveCMAES(mfev=10 ** 7, tol=1e-6, np=12, seed=3, populations=P)
    g.initialize(hip.objectives.sphere, lo, up, guess)
    done = g.run(5000)
    assert done < 5000
    its = [int(g.get_state("it", p)[0]) for p in range(P)]
    flags = [int(g.get_state("flag", p)[0]) for p in range(P)]
    assert all(f != 0 for f in flags)
    assert len(set(its)) > 1            # they did not all stop in the same generation
    assert max(its) <= done
    for p in range(P):
        assert g.solution(p).converged
