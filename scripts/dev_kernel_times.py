"""dev (GPU box): per-kernel HIP-event averages of a CMA workload, optionally under diagnostic bits:
    python scripts/dev_kernel_times.py [n lam P gens [dbg [algo]]]
prints the untimed-region ms per generation first (timers off), then the per-kernel averages."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bboptpy_amd import _ffi   # noqa: E402
if os.environ.get("BBO_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["BBO_LIB"])
import bboptpy_amd as bb   # noqa: E402
import bench   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lam = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
gens = int(sys.argv[4]) if len(sys.argv) > 4 else 60
dbg = int(sys.argv[5]) if len(sys.argv) > 5 else 0
algo = sys.argv[6] if len(sys.argv) > 6 else "ActiveCMAES"
obj = sys.argv[7] if len(sys.argv) > 7 else "rosenbrock"
cls = getattr(bb, algo)
alg = cls(mfev=2 ** 31 - 1, tol=0., np=lam, seed=1, populations=P, poll_every=gens)
g = np.random.default_rng(0).uniform(-10, 10, (P, n))
alg.initialize(getattr(bb.objectives, obj), -10 * np.ones(n), 10 * np.ones(n), g)
if dbg:
    alg.set_state("dbg", [float(dbg)])
for kv in os.environ.get("BBO_SET", "").split(","):      # e.g. BBO_SET=sample128_min=1024,eig_split_maxp=4
    if "=" in kv:
        alg.set_state(kv.split("=")[0], [float(kv.split("=")[1])])
alg.run(10)
t = time.perf_counter()
d = alg.run(gens)
dt = time.perf_counter() - t
print("n %d lam %d P %d dbg %d: %.4f ms/generation, %.3e evals/s" % (n, lam, P, dbg, 1e3 * dt / d, P * lam * d / dt))
alg.set_state("profile", [1.0])
alg.run(gens)
prof = alg.get_state("profile")
tot = 0.
for i, name in enumerate(bench.CMA_KERNELS):
    if prof[2 * i + 1] > 0:
        us = 1e3 * prof[2 * i] / prof[2 * i + 1]
        tot += us
        print("   %-18s %8.1f us" % (name, us))
print("   %-18s %8.1f us" % ("sum", tot))
try:
    print("   eig_mw_fail %s, eig_mw_off %s" % (alg.get_state("eig_mw_fail"), alg.get_state("eig_mw_off")))
except Exception:
    pass
