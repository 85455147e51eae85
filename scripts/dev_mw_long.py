"""dev (GPU box): long runs through the spread reduction -- does a wavefront ever give up?
    python scripts/dev_mw_long.py n P generations"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bboptpy_amd import _ffi
if os.environ.get("BBO_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["BBO_LIB"])
import bboptpy_amd as bb
n, P, gens = (int(a) for a in sys.argv[1:4])
alg = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=20, seed=1, populations=P, poll_every=50)
g = np.random.default_rng(0).uniform(-5, 5, (P, n)) if P > 1 else np.random.default_rng(0).uniform(-5, 5, n)
alg.initialize(bb.objectives.rastrigin, -5.12 * np.ones(n), 5.12 * np.ones(n), g)
alg.set_state("stop_off", [1023.])
t = time.perf_counter()
done = alg.run(gens)
dt = time.perf_counter() - t
fails = [int(alg.get_state("eig_mw_fail", p)[0]) for p in range(P)]
print("n %d P %d: %d generations in %.2f s (%.3f ms each), gave up: %d, switched off: %d, f finite: %s"
      % (n, P, done, dt, 1e3 * dt / max(done, 1), sum(fails), int(alg.get_state("eig_mw_off")[0]),
         bool(np.all(np.isfinite(alg.get_state("xmean"))))))
