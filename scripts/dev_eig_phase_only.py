"""dev (GPU box): the decomposition phase alone, P populations, under diagnostic bits (a run that
stops the eigensolver early has no usable basis: whole generations cannot be timed with it)
    python scripts/dev_eig_phase_only.py n P reps dbg"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as bb
from bboptpy_amd import _ffi
n, P, reps, dbg = (int(a) for a in sys.argv[1:5])
g = bb.ActiveCMAES(mfev=10 ** 9, tol=0., np=4 * n, seed=3, populations=P)
g.initialize(bb.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n), np.random.default_rng(1).uniform(-10, 10, (P, n)))
g.run(12)                         # covariances with some structure
if dbg:
    g.set_state("dbg", [float(dbg)])
for rep in range(reps):
    for p in range(P):
        g.set_state("eigenlastev", [0], p)
    g.phase(_ffi.PHASE_EIGEN)
print("done")
