"""A/B of cma_gram128s with and without its pacing barrier (diagnostic bit 65536 = off):
per-kernel HIP-event averages of the M step, 256 populations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
from bboptpy_amd import _ffi
if os.environ.get("BBO_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["BBO_LIB"])   # a variant build (scripts/_variants)
P, n, lam = 256, 128, 4096
bits = [int(a) for a in sys.argv[1:]] or [0, 65536]
for dbg in bits:
    alg = b.ActiveCMAES(mfev=2**31 - 1, tol=1e-30, np=lam, seed=1, populations=P)
    g = np.random.default_rng(0).uniform(-10, 10, (P, n))
    alg.initialize(b.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n), g)
    alg.set_state("dbg", [float(dbg)])
    alg.run(10)
    alg.set_state("profile", [1.0])
    alg.run(40)
    prof = alg.get_state("profile")
    print("dbg", dbg, " ".join("%.1f" % (1e3 * prof[2 * i] / max(prof[2 * i + 1], 1)) for i in range(len(prof) // 2)), "us per launch (kernel slots in engine order)")
