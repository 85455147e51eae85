"""dev: does the n = 18 identity decomposition depend on what an earlier kernel left in LDS?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bboptpy_amd import _ffi
if sys.argv[1] != "-":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import bboptpy_amd as hip
n = int(sys.argv[2])
poison = sys.argv[3]
g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=2 * n, seed=1)
g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
if poison != "none":
    P = 256
    h = hip.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=4096 if poison == "sampler" else 1024, seed=3, populations=P)
    h.initialize(hip.objectives.rosenbrock, -10 * np.ones(128), 10 * np.ones(128),
                 np.random.default_rng(0).uniform(-10, 10, (P, 128)))
    h.run(2 if poison == "sampler" else 1)
for rep in range(2):
    g.set_state("C", np.eye(n)); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
    g.phase(_ffi.PHASE_EIGEN)
    B = g.get_state("B").reshape(n, n)
    print(poison, "rep", rep, "orth %.3e" % (np.linalg.norm(B.T @ B - np.eye(n)) / n),
          "dup cols:", [c for c in range(n) if np.abs(B[:, c]).sum() != 1.0])
