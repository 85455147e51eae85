"""dev (GPU box): a plain run of a CMA workload (timers off) for a rocprofv3 --kernel-trace timeline:
    python scripts/dev_plain_run.py [n lam P gens [dbg [objective]]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bboptpy_amd import _ffi   # noqa: E402
if os.environ.get("BBO_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["BBO_LIB"])
import bboptpy_amd as bb   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lam = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 1
gens = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
obj = sys.argv[6] if len(sys.argv) > 6 else "rosenbrock"
alg = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=lam, seed=1, populations=P, poll_every=gens)
g = np.random.default_rng(0).uniform(-10, 10, (P, n))
alg.initialize(getattr(bb.objectives, obj), -10 * np.ones(n), 10 * np.ones(n), g)
if dbg:
    alg.set_state("dbg", [float(dbg)])
alg.run(10)
t = time.perf_counter()
d = alg.run(gens)
dt = time.perf_counter() - t
print("n %d lam %d P %d dbg %d: %.4f ms/generation" % (n, lam, P, dbg, 1e3 * dt / d))
