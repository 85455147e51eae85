import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import bboptpy_amd as hip
import pyoracle as po
L = po.oracle()
for obj, n in (("ellipsoid", 64), ("rastrigin", 20)):
    lam = 4 * (4 + int(3 * np.log(n)))
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(1).uniform(-4, 4, n)
    g = hip.SepCMAES(mfev=400000, tol=1e-10, np=lam, sigma0=2., adjustlr=True, seed=5)
    sol = g.optimize(getattr(hip.objectives, obj), lo, up, guess)
    print(obj, "device: evals", sol.n_evals, "conv", sol.converged, "flag", g.get_state("flag")[0], "f", L.objective(obj, sol.x), "sigma", g.get_state("sigma")[0])
    o = po.cma(L, "sep", 400000, 1e-10, lam, sigma0=2., adjustlr=True)
    o.set_rng(po.RNG_PHILOX, 5)
    xo, fevo, convo = o.optimize(obj, lo, up, guess)
    print(obj, "oracle(philox): evals", fevo, "conv", convo, "flag", o.scalar("flag"), "f", L.objective(obj, xo), "sigma", o.scalar("sigma"))
    L.seed(5)
    o = po.cma(L, "sep", 400000, 1e-10, lam, sigma0=2., adjustlr=True)
    xo, fevo, convo = o.optimize(obj, lo, up, guess)
    print(obj, "oracle(mt): evals", fevo, "conv", convo, "flag", o.scalar("flag"), "f", L.objective(obj, xo))
