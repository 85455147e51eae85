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


@pytest.mark.parametrize("variant,n,lam,P,bound,obj", [
    ("active", 10, 20, 1, False, "rosenbrock"),        # C1
    ("active", 10, 20, 37, False, "rosenbrock"),
    ("cmaes", 5, 8, 3, False, "sphere"),
    ("active", 16, 64, 5, True, "rastrigin"),          # the largest fused shape, clamped samples
    ("cmaes", 2, 6, 2, False, "ellipsoid"),
    ("active", 13, 33, 4, False, "ackley"),            # ragged lambda (lambda_pad = 48)
])
def test_fused_small_generations_equal_the_kernel_sequence(hip, variant, n, lam, P, bound, obj):
    """n <= 16, lambda <= 64: run() / iterate() execute whole generations in ONE launch
    (cma_small_generations: the bodies of the nine kernels back to back inside one workgroup per
    population).  Same code, same arithmetic: the state after 25 generations -- several of them
    per launch -- is BIT-IDENTICAL to the nine-kernel sequence (diagnostic bit 64 keeps it)."""
    cls = hip.ActiveCMAES if variant == "active" else hip.CMAES
    lo, up = -3. * np.ones(n), 3. * np.ones(n)
    guess = np.random.default_rng(n).uniform(-2, 2, (P, n))

    def make(dbg):
        g = cls(mfev=10 ** 7, tol=1e-12, np=lam, seed=99, populations=P, bound=bound)
        g.initialize(getattr(hip.objectives, obj), lo, up, guess)
        if dbg:
            g.set_state("dbg", [float(dbg)])
        return g

    a, b = make(0), make(64)
    a.run(13)                  # 8 + 5 generations: two launches
    b.run(13)
    for _ in range(12):        # and one generation per launch
        a.iterate()
        b.iterate()
    for p in (0, P // 2, P - 1):
        for key in ("xmean", "sigma", "pc", "ps", "C", "B", "D", "invsqrtC", "arx", "fitness",
                    "fit_idx", "it", "fev", "flag", "best_hist", "kth_hist", "fbest", "fworst"):
            np.testing.assert_array_equal(a.get_state(key, p), b.get_state(key, p), err_msg=key)
    assert int(a.get_state("it")[0]) == 25


def test_fused_small_run_freezes_stopped_populations(hip):
    """a population that stops inside a multi-generation launch stays exactly at its stopping
    generation (the in-kernel loop honours the stop flag like the kernel sequence does)"""
    n, lam, P = 4, 8, 6
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(1).uniform(-3, 3, (P, n))
    runs = []
    for dbg in (0, 64):
        g = hip.ActiveCMAES(mfev=4000, tol=1e-8, np=lam, seed=5, populations=P)
        g.initialize(hip.objectives.sphere, lo, up, guess)
        if dbg:
            g.set_state("dbg", [64.])
        g.run(10000)
        runs.append([(g.solution(p).n_evals, g.solution(p).converged, g.solution(p).x.tolist(),
                      int(g.get_state("it", p)[0])) for p in range(P)])
    assert runs[0] == runs[1]
    assert len({r[3] for r in runs[0]}) > 1          # they did stop at different generations


@pytest.mark.parametrize("lam,P", [(4096, 8), (1024, 32)])
def test_lean_sampler_build_equals_the_general_one(hip, lam, P):
    """cma_sample_eval128 has a lean build of its tile loop for n = 128 exactly, no box, lambda a
    multiple of 16 (M, C3): no bounds tests, clamps or per-store branches.  Same arithmetic:
    X, f and everything downstream are BIT-IDENTICAL to the general build (diagnostic bit 256)."""
    n = 128
    lo, up = -10. * np.ones(n), 10. * np.ones(n)
    guess = np.random.default_rng(4).uniform(-10, 10, (P, n))
    runs = []
    for dbg in (0, 256):
        g = hip.ActiveCMAES(mfev=10 ** 9, tol=0., np=lam, seed=77, populations=P)
        g.initialize(hip.objectives.rosenbrock, lo, up, guess)
        if dbg:
            g.set_state("dbg", [float(dbg)])
        g.run(5)
        runs.append([(g.get_state("arx", p), g.get_state("fitness", p), g.get_state("C", p),
                      g.get_state("sigma", p)) for p in (0, P - 1)])
    for a, b in zip(runs[0], runs[1]):
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)


def test_lean_separable_sampler_equals_the_general_one(hip):
    """sep_sample_eval has the same kind of lean build (n == ld, no box, lambda == lambda_pad: the
    benchmark's SEP shape): X, f and the state downstream BIT-IDENTICAL to the general build
    (diagnostic bit 1048576 keeps the row-in-LDS form where the sum-on-draw kernel would run)."""
    n, lam, P = 1024, 256, 4
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(6).uniform(-5, 5, (P, n))
    runs = []
    for dbg in (1048576, 256):
        g = hip.SepCMAES(mfev=10 ** 9, tol=0., np=lam, seed=78, populations=P)
        g.initialize(hip.objectives.ellipsoid, lo, up, guess)
        g.set_state("dbg", [float(dbg)])
        g.run(5)
        runs.append([(g.get_state("arx", p), g.get_state("fitness", p), g.get_state("D", p), g.get_state("csep", p),
                      g.get_state("sigma", p)) for p in (0, P - 1)])
    for a, b in zip(runs[0], runs[1]):
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)


@pytest.mark.parametrize("obj", ["sphere", "ellipsoid", "rastrigin", "cigar", "discus", "diffpow"])
@pytest.mark.parametrize("n,lam", [(1024, 256), (512, 64), (2048, 48)])
def test_sum_on_draw_separable_sampler(hip, obj, n, lam):
    """sep_sample_sum (objectives that are sums of per-coordinate terms, lean shapes): no row in
    LDS, a lane adds the terms of the coordinates it draws.  X is BIT-IDENTICAL to the row-in-LDS
    kernel's (same Philox counters, same ziggurat, same x = m + sigma d z); f equals the objective
    of the stored row (numpy, and the row-in-LDS kernel's value) to rounding -- the terms are added
    in another order."""
    from bboptpy_amd import _ffi
    P = 3
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(n + lam).uniform(-4, 4, (P, n))
    out = []
    for dbg in (0, 1048576):
        g = hip.SepCMAES(mfev=10 ** 9, tol=0., np=lam, seed=5, populations=P)
        g.initialize(getattr(hip.objectives, obj), lo, up, guess)
        if dbg:
            g.set_state("dbg", [float(dbg)])
        g.run(2)
        g.phase(_ffi.PHASE_SAMPLE_EVALUATE)
        out.append([(g.get_state("arx", p).reshape(lam, n), g.get_state("fitness", p)[:lam]) for p in range(P)])
    t = np.arange(n) / (n - 1.)
    for (Xa, fa), (Xb, fb) in zip(*out):
        np.testing.assert_array_equal(Xa, Xb)
        np.testing.assert_allclose(fa, fb, rtol=2e-13, atol=0)
        want = {"sphere": lambda X: (X * X).sum(1),
                "ellipsoid": lambda X: ((10. ** (6. * t)) * X * X).sum(1),
                "rastrigin": lambda X: 10. * n + (X * X - 10. * np.cos(2. * np.pi * X)).sum(1),
                "cigar": lambda X: X[:, 0] ** 2 + 1e6 * (X[:, 1:] ** 2).sum(1),
                "discus": lambda X: 1e6 * X[:, 0] ** 2 + (X[:, 1:] ** 2).sum(1),
                "diffpow": lambda X: (np.abs(X) ** (2. + 4. * t)).sum(1)}[obj](Xa)
        np.testing.assert_allclose(fa, want, rtol=1e-12, atol=0)


@pytest.mark.parametrize("variant,n,lam,P", [("ActiveCMAES", 128, 4096, 8), ("ActiveCMAES", 128, 1000, 16),
                                             ("CMAES", 120, 300, 40)])
def test_streaming_gram_equals_the_lds_staged_one(hip, variant, n, lam, P):
    """cma_gram128s (every wavefront loads its own MFMA fragments from X, no LDS tile, no
    barrier) against cma_gram128 (diagnostic bit 512): the same products in the same order, so C,
    the mean and everything downstream are BIT-IDENTICAL -- full and ragged shapes (lambda not a
    multiple of the chunk, n < ld)."""
    lo, up = -10. * np.ones(n), 10. * np.ones(n)
    guess = np.random.default_rng(4).uniform(-10, 10, (P, n))
    runs = []
    for dbg in (0, 512):
        g = getattr(hip, variant)(mfev=10 ** 9, tol=0., np=lam, seed=79, populations=P)
        g.initialize(hip.objectives.rosenbrock, lo, up, guess)
        if dbg:
            g.set_state("dbg", [float(dbg)])
        g.run(4)
        runs.append([(g.get_state("C", p), g.get_state("xmean", p), g.get_state("sigma", p),
                      g.get_state("arx", p)) for p in (0, P - 1)])
    for a, b in zip(runs[0], runs[1]):
        for u, v in zip(a, b):
            np.testing.assert_array_equal(u, v)
