"""dev: run the identity matrix through a variant library and print the top merge's maps dumped into
the F slab of eig_work (variant F of round 4's n = 18 hunt)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bboptpy_amd import _ffi
if sys.argv[1] != "-":
    _ffi.LIB_PATH = os.path.abspath(sys.argv[1])
import bboptpy_amd as hip
n = int(sys.argv[2])
g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=2 * n, seed=1)
g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
Cm = np.eye(n)
g.set_state("C", Cm); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
g.phase(_ffi.PHASE_EIGEN)
B = g.get_state("B").reshape(n, n)
print("orth", np.linalg.norm(B.T @ B - np.eye(n)) / n)
print("B nonzero cols per row:", [list(np.nonzero(B[r])[0]) for r in range(n)])
w = g.get_state("eig_work")
ld = 32 if n <= 32 else 64
slab = (ld + 32) * (ld + 32) + 72
F = w[2 * slab:3 * slab]
dbg = F[4096:4096 + 400]
for name, off in (("outpos", 0), ("rowmap", 64), ("srcS", 128), ("colroot", 192), ("dp", 256), ("lam", 320)):
    print(name, dbg[off:off + n])
print("k nd nr maxnr direct m a mid sorted_in any_unsorted cnt3:", dbg[384:395])
