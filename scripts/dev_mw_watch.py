"""dev (GPU box): per-generation wall time of an n = 256 run (a stall of ~1 s = a wavefront of
cma_tred_mw gave up waiting)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bboptpy_amd as bb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
alg = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=20, seed=1, populations=P, poll_every=1)
g = np.random.default_rng(0).uniform(-10, 10, (P, n)) if P > 1 else np.random.default_rng(0).uniform(-10, 10, n)
alg.initialize(bb.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n), g)
ts = []
for k in range(60):
    t = time.perf_counter()
    alg.run(1)
    ts.append(time.perf_counter() - t)
print("ms per generation:", " ".join("%.2f" % (1e3 * t) for t in ts))
