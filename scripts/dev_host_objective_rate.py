"""The PCIe-inclusive rate of the metric configuration: ActiveCMAES n=128, lambda=4096, ONE
population, the objective a numpy-vectorised Python callable on the host (X leaves HBM once per
generation, f comes back) against the same run with the built-in device objective."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b

n, lam, gens = 128, 4096, 100


def rosen(X):
    return (100. * (X[:, 1:] - X[:, :-1] ** 2) ** 2 + (1. - X[:, :-1]) ** 2).sum(1)


rosen._bbo_vectorized = True


def scalar_rosen(x):
    return float((100. * (x[1:] - x[:-1] ** 2) ** 2 + (1. - x[:-1]) ** 2).sum())


g = np.random.default_rng(0).uniform(-10, 10, n)
for name, f, gg in (("device built-in", b.objectives.rosenbrock, gens),
                    ("host, vectorised callable", rosen, gens),
                    ("host, scalar callable", scalar_rosen, 10)):
    alg = b.ActiveCMAES(mfev=2**31 - 1, tol=1e-30, np=lam, seed=1)
    alg.initialize(f, -10 * np.ones(n), 10 * np.ones(n), g)
    for _ in range(3):
        alg.iterate()
    t = time.perf_counter()
    for _ in range(gg):
        alg.iterate()
    alg.get_state("sigma")
    dt = time.perf_counter() - t
    print("%-28s %8.3f ms/generation  %.3e evaluations/s" % (name, 1e3 * dt / gg, lam * gg / dt))
