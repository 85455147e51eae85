"""dev: stage-by-stage check of the n > 256 eigensolver path from its global scratch (bbo_get
"eig_work": [work | Q_house | F | QF] slabs)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as hip
from bboptpy_amd import _ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 280
ld = (n + 15) // 16 * 16
rng = np.random.default_rng(n)
g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=2 * n, seed=1)
g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
X = rng.normal(size=(n, 3 * n))
Cm = np.eye(n) + 0.3 * (X @ X.T) / (3 * n)
g.set_state("C", Cm); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
g.phase(_ffi.PHASE_EIGEN)
B = g.get_state("B").reshape(n, n); D = g.get_state("D")
lam = np.linalg.eigvalsh(Cm)
print("eigenvalues err", np.abs(D * D - lam).max())
print("B orth", np.linalg.norm(B.T @ B - np.eye(n)), "resid", np.linalg.norm(B @ np.diag(D * D) @ B.T - Cm))
w = g.get_state("eig_work")
slab = (ld + 32) ** 2 + 72
lda = (n + 31) // 32 * 32
work = w[:n * lda].reshape(n, lda)[:, :n]
Qh = w[slab:slab + n * n].reshape(n, n)
F = w[slab + n * n: slab + 2 * n * n].reshape(n, n)
M = w[3 * slab:3 * slab + n * n].reshape(n, n)
print("Qh orth", np.linalg.norm(Qh.T @ Qh - np.eye(n)))
T = Qh.T @ Cm @ Qh
off = T - np.diag(np.diag(T)) - np.diag(np.diag(T, 1), 1) - np.diag(np.diag(T, -1), -1)
print("Qh^T C Qh tridiagonal? off-band norm", np.linalg.norm(off), " (and Qh C Qh^T:", end=" ")
T2 = Qh @ Cm @ Qh.T
off2 = T2 - np.diag(np.diag(T2)) - np.diag(np.diag(T2, 1), 1) - np.diag(np.diag(T2, -1), -1)
print(np.linalg.norm(off2), ")")
print("work (block-diag Q) orth", np.linalg.norm(work.T @ work - np.eye(n)))
print("F orth", np.linalg.norm(F.T @ F - np.eye(n)))
print("M = work F ?", np.linalg.norm(work @ F - M), " M orth", np.linalg.norm(M.T @ M - np.eye(n)))
print("B = Qh M ?", np.linalg.norm(Qh @ M - B))
Tq = M.T @ (T if np.linalg.norm(off) < np.linalg.norm(off2) else T2) @ M
print("M diagonalises T?", np.linalg.norm(Tq - np.diag(np.diag(Tq))))
# halves of the block-diagonal matrix
h = n // 2
print("work off-diagonal blocks", np.linalg.norm(work[:h, h:]), np.linalg.norm(work[h:, :h]))
