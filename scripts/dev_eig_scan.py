"""dev: identity + near-identity decomposition for every n in a range; prints the n that fail"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bboptpy_amd import _ffi
if sys.argv[1] != "-":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import bboptpy_amd as hip
bad = []
for n in range(int(sys.argv[2]), int(sys.argv[3]) + 1):
    g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=2 * n, seed=1)
    g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
    rng = np.random.default_rng(n)
    X = rng.normal(size=(n, 3 * n))
    errs = []
    for Cm in (np.eye(n), np.eye(n) + 1e-3 * (X @ X.T) / (3 * n), X @ X.T / (3 * n)):
        Cm = 0.5 * (Cm + Cm.T)
        g.set_state("C", Cm); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
        g.phase(_ffi.PHASE_EIGEN)
        B = g.get_state("B").reshape(n, n); D = g.get_state("D")
        errs.append(np.linalg.norm(B @ np.diag(D * D) @ B.T - Cm) / np.linalg.norm(Cm))
    ok = all(e == e and e < 1e-11 for e in errs)
    if not ok:
        bad.append(n)
print(os.path.basename(sys.argv[1]), "failing n:", bad)
