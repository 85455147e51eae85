"""dev: one special matrix of tests/test_cma_gpu.py at one n; prints eigenvalue differences"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bboptpy_amd as hip
from bboptpy_amd import _ffi
from test_cma_gpu import _spd_cases
n = int(sys.argv[1]); which = sys.argv[2]; dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(n)
g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=2 * n, seed=1)
g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
if dbg: g.set_state("dbg", [float(dbg)])
for name, Cm in _spd_cases(n, rng):
    if name != which: continue
    Cm = 0.5 * (Cm + Cm.T)
    g.set_state("C", Cm); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
    g.phase(_ffi.PHASE_EIGEN)
    B = g.get_state("B").reshape(n, n); D = g.get_state("D")
    lam = np.linalg.eigvalsh(Cm)
    print("D^2:", np.round(D * D, 6))
    print("true:", np.round(lam, 6)[:8], "...", np.round(lam, 6)[-8:])
    print("orth", np.linalg.norm(B.T @ B - np.eye(n)) / n, "nan in B", np.isnan(B).sum())
