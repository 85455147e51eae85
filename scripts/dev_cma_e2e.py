import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import bboptpy_amd as hip
import pyoracle as po
L = po.oracle()
for variant, cls in (("active", hip.ActiveCMAES), ("cmaes", hip.CMAES)):
    for seed in (1, 2, 3):
        n, lam = 10, 20
        lo, up = -10. * np.ones(n), 10. * np.ones(n)
        guess = np.random.default_rng(seed).uniform(-10, 10, n)
        g = cls(mfev=10000, tol=1e-4, np=lam, seed=seed)
        sol = g.optimize(hip.objectives.rosenbrock, lo, up, guess)
        o = po.cma(L, variant, 10000, 1e-4, lam)
        o.set_rng(po.RNG_PHILOX, seed)
        xo, fevo, convo = o.optimize("rosenbrock", lo, up, guess)
        print(variant, seed, "device evals", sol.n_evals, "flag", g.get_state("flag")[0], "f %.3e" % L.objective("rosenbrock", sol.x), "| oracle evals", fevo, "flag", o.scalar("flag"), "f %.3e" % L.objective("rosenbrock", xo), "dx %.2e" % np.abs(sol.x - xo).max())
