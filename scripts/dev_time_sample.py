# dev helper: time the CMA sampling kernel alone
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
from bboptpy_amd import _ffi
n, lam = 128, 4096
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
alg = b.ActiveCMAES(mfev=2**31-1, tol=1e-30, np=lam, seed=1, populations=P)
g = np.random.default_rng(0).uniform(-10, 10, (P, n))
alg.initialize(b.objectives.rosenbrock, -10*np.ones(n), 10*np.ones(n), g)
alg.run(3)
for rw in (4, 8, 16, 32):
    for prio in (0, 1, 2):
        alg.set_state("dbg", [float(prio + 256 * rw)])
        alg.set_state("profile", [1.0])
        for _ in range(6):
            alg.phase(_ffi.PHASE_SAMPLE_EVALUATE)
        r = alg.get_state("profile")
        print("P", P, "rw", rw * 128, "prio mode", prio, "sample_eval avg us: %.1f (%d calls)" % (1e3 * r[0] / max(r[1], 1), r[1]))
