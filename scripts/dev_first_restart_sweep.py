#!/usr/bin/env python3
"""Dev script (GPU box): how often does the FIRST restart's inner run of the device's BIPOP driver
use exactly the evaluations the oracle's does under the same seed?  (The first restart samples its
first generation through the B the first run left behind -- see
tests/test_restart_gpu.py::test_restart_decisions_match_oracle_restart.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po          # noqa: E402
import bboptpy_amd as hip      # noqa: E402

L = po.oracle()
n, obj, mfev = 10, "rosenbrock", 40000
lo, up = -5. * np.ones(n), 5. * np.ones(n)
book = ("fev", "it", "largelambda", "largebudget", "smallbudget", "largerestarts",
        "smallrestarts", "bestregime", "fxbest")
same = 0
seeds = list(range(23, 23 + int(sys.argv[1]) if len(sys.argv) > 1 else 43))
for seed in seeds:
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    base = hip.ActiveCMAES(mfev=1, tol=1e-6, np=4)
    ob = po.cma(L, "active", 1, 1e-6, 4)
    drv = hip.BiPopCMAES(base, mfev=mfev, nbipop=True, seed=seed)
    o = po.bipop(L, ob, mfev, nbipop=True)
    drv.initialize(getattr(hip.objectives, obj), lo, up, guess)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init(obj, lo, up, guess)
    gd = lambda k: drv.get_state(k)[0]
    first = (int(gd("last_inner_fev")), int(o.scalar("last_inner_fev")))
    for k in book:
        o.rset(k, gd(k))
    drv.iterate()
    o.iterate()
    second = (int(gd("last_inner_fev")), int(o.scalar("last_inner_fev")))
    same += second[0] == second[1]
    print(seed, "first run", first, "first restart", second, "fx %.3e %.3e" % (gd("fx"), o.scalar("fx")),
          flush=True)
print("first restart identical in %d of %d" % (same, len(seeds)))
