"""GPU, at BASELINE.json's full sizes (C2: L-SHADE n = 128, np = 4096; C4: APSO n = 512,
np = 65536; M: ActiveCMAES n = 128, lambda = 4096): properties that do not depend on the size.
(The oracle itself follows M, C3 and C2 for a few generations -- ~0.15 s per generation at M --
and does: tests/test_cma_headline_gpu.py, tests/test_de_gpu.py::test_c2_full_size_*.  Only C4's
O(np^2 n) reference generation, about half an hour, is out of its reach.)

* the stored fitness IS the objective of the stored position (recomputed on the host),
* selection never loses ground (sorted fitness / personal bests are element-wise non-increasing),
* box and velocity limits hold, evaluation counters advance by the reference's amounts,
* APSO's all-pairs mean distances (the MFMA Gram kernel) equal a direct numpy evaluation for
  sampled particles, the evolutionary factor is the reference's function of them,
* CMA-ES: ranks are a permutation that sorts f, B is orthonormal, B D^2 B^T reproduces C,
  C^-1/2 C C^-1/2 = I.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rastrigin(X):
    return 10. * X.shape[1] + (X * X - 10. * np.cos(2. * np.pi * X)).sum(axis=1)


def _rosenbrock(X):
    return (100. * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1. - X[:, :-1]) ** 2).sum(axis=1)


def test_c2_lshade_full_size_properties(hip):
    n, npop = 128, 4096
    lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
    g = hip.SHADE(mfev=10 ** 8, npinit=npop, tol=0., npmin=npop, seed=11)
    g.initialize(hip.objectives.rastrigin, lo, up, np.zeros(n))
    f_prev = g.get_state("f")
    assert f_prev.shape == (npop,) and np.all(np.diff(f_prev) >= 0)     # sorted, like _swarm
    fev = int(g.get_state("fev")[0])
    assert fev == npop
    for gen in range(6):
        g.iterate()
        x = g.get_state("x").reshape(npop, n)
        f = g.get_state("f")
        assert np.all(x >= lo) and np.all(x <= up)
        np.testing.assert_allclose(f, _rastrigin(x), rtol=1e-12, atol=1e-9)
        assert np.all(np.diff(f) >= 0)
        # every individual keeps its fitness or improves: the sorted vectors dominate
        assert np.all(f <= f_prev)
        f_prev = f
        assert int(g.get_state("fev")[0]) == fev + npop
        fev += npop
        assert 0 <= int(g.get_state("larch")[0]) <= npop
        mcr, mf = g.get_state("MCR"), g.get_state("MF")
        assert np.all((mcr >= 0) & (mcr <= 1)) and np.all((mf > 0) & (mf <= 1))
    assert f_prev[0] < _rastrigin(np.zeros((1, n)) + 5.12)[0]


def test_c4_apso_full_size_properties(hip):
    n, npop = 512, 65536
    lo, up = -10. * np.ones(n), 10. * np.ones(n)
    g = hip.APSO(mfev=2 ** 31 - 1, tol=0., np=npop, seed=5)
    g.initialize(hip.objectives.sphere, lo, up, np.zeros(n))
    fb_prev = g.get_state("fb")
    fev = int(g.get_state("fev")[0])
    rng = np.random.default_rng(0)
    for gen in range(2):
        x_before = g.get_state("x").reshape(npop, n)
        f_before = g.get_state("f")
        g.iterate()
        # the evolutionary factor of this generation was computed on x_before
        ws = g.get_state("ws")
        sample = np.concatenate([rng.integers(0, npop, 3), [int(np.argmin(f_before))]])
        for i in sample:
            d = np.sqrt(((x_before - x_before[i]) ** 2).sum(axis=1)).sum() / (npop - 1.)
            assert abs(ws[i] - d) <= 1e-9 * d
        ig = int(np.argmin(f_before))
        evof = (ws[ig] - ws.min()) / (ws.max() - ws.min())
        assert abs(float(g.get_state("evof")[0]) - evof) <= 1e-9
        assert 0. <= evof <= 1.
        x = g.get_state("x").reshape(npop, n)
        v = g.get_state("v").reshape(npop, n)
        f, fb = g.get_state("f"), g.get_state("fb")
        assert np.all(x >= lo) and np.all(x <= up)
        assert np.all(np.abs(v) <= 0.2 * (up - lo) * (1 + 1e-15))
        np.testing.assert_allclose(f, (x * x).sum(axis=1), rtol=1e-12)
        assert np.all(fb <= fb_prev) and np.all(fb <= f)
        fb_prev = fb
        xb = g.get_state("xb").reshape(npop, n)
        np.testing.assert_allclose(fb, (xb * xb).sum(axis=1), rtol=1e-12)
        # the incumbent is at least as good as every personal best (elitist learning may
        # have improved it beyond them)
        assert float(g.get_state("fbest")[0]) <= fb.min()
        new_fev = int(g.get_state("fev")[0])
        assert new_fev - fev in (npop, npop + 1)        # + 1: the elitist-learning evaluation
        fev = new_fev


def test_m_active_cma_full_size_properties(hip):
    n, lam = 128, 4096
    lo, up = -10. * np.ones(n), 10. * np.ones(n)
    g = hip.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=lam, seed=3)
    g.initialize(hip.objectives.rosenbrock, lo, up, np.random.default_rng(1).uniform(-10, 10, n))
    for gen in range(4):
        g.iterate()
        x = g.get_state("arx").reshape(lam, n)
        f = g.get_state("fitness")
        np.testing.assert_allclose(f, _rosenbrock(x), rtol=1e-11)
        rank = g.get_state("rank").astype(int)
        assert sorted(rank.tolist()) == list(range(lam))
        order = np.argsort(rank)
        assert np.all(np.diff(f[order]) >= 0)
        B = g.get_state("B").reshape(n, n)
        D = g.get_state("D")
        C = g.get_state("C").reshape(n, n)
        Cs = np.tril(C) + np.tril(C, -1).T                   # only the lower triangle is live
        assert np.abs(B.T @ B - np.eye(n)).max() <= 1e-13 * n
        assert np.linalg.norm(B @ np.diag(D * D) @ B.T - Cs) <= 1e-13 * np.linalg.norm(Cs) * n
        S = g.get_state("invsqrtC").reshape(n, n)
        assert np.abs(S @ Cs @ S - np.eye(n)).max() <= 1e-10
        assert np.all(np.diff(D) >= 0) and D[0] > 0
        assert int(g.get_state("fev")[0]) == (gen + 1) * lam
