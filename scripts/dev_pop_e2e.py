import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import bboptpy_amd as hip
import pyoracle as po
L = po.oracle()
n = 10
lo, up = -5. * np.ones(n), 5. * np.ones(n)
for obj in ("sphere", "rosenbrock", "ellipsoid"):
    for seed in (1, 2):
        g = hip.SHADE(mfev=60000, npinit=40, tol=1e-8, seed=seed)
        sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
        o = po.shade(L, 60000, 40, 1e-8); o.set_mode(True, po.RNG_PHILOX, seed)
        xo, fo, co = o.optimize(obj, lo, up, np.zeros(n))
        print("SHADE", obj, seed, sol.n_evals, fo, sol.converged, co, "%.3e %.3e" % (L.objective(obj, sol.x), L.objective(obj, xo)), "dx %.2e" % np.abs(sol.x - xo).max())
        g = hip.JADE(mfev=60000, np=30, tol=1e-8, seed=seed)
        sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
        o = po.jade(L, 60000, 30, 1e-8); o.set_mode(True, po.RNG_PHILOX, seed)
        xo, fo, co = o.optimize(obj, lo, up, np.zeros(n))
        print("JADE ", obj, seed, sol.n_evals, fo, sol.converged, co, "%.3e %.3e" % (L.objective(obj, sol.x), L.objective(obj, xo)), "dx %.2e" % np.abs(sol.x - xo).max())
        g = hip.APSO(mfev=60000, tol=1e-8, np=30, seed=seed)
        sol = g.optimize(getattr(hip.objectives, obj), lo, up, np.zeros(n))
        o = po.apso(L, 60000, 1e-8, 30); o.set_mode(True, po.RNG_PHILOX, seed)
        xo, fo, co = o.optimize(obj, lo, up, np.zeros(n))
        print("APSO ", obj, seed, sol.n_evals, fo, sol.converged, co, "%.3e %.3e" % (L.objective(obj, sol.x), L.objective(obj, xo)), "dx %.2e" % np.abs(sol.x - xo).max())
