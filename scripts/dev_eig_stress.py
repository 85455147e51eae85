"""one-off stress run of the eigensolver over every n in 2..260 (not a test: takes a while)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import bboptpy_amd as hip
from bboptpy_amd import _ffi
from test_cma_gpu import _spd_cases
worst = {}
ns = (list(range(2, 150)) + [150, 160, 176, 191, 192, 193, 200, 224, 240, 255, 256]) if len(sys.argv) < 2 else list(range(int(sys.argv[1]), int(sys.argv[2]) + 1))
for n in ns:
    rng = np.random.default_rng(1000 + n)
    g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=max(4, 2 * n), seed=1)
    g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
    for name, Cm in _spd_cases(n, rng):
        Cm = 0.5 * (Cm + Cm.T)
        g.set_state("C", Cm); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
        g.phase(_ffi.PHASE_EIGEN)
        B = g.get_state("B").reshape(n, n); D = g.get_state("D")
        lam = np.linalg.eigvalsh(Cm); sc = np.abs(lam).max()
        e1 = np.abs(D * D - np.maximum(lam, lam.max() / 1e14)).max() / sc
        e2 = np.linalg.norm(B.T @ B - np.eye(n)) / n
        e3 = np.linalg.norm(B @ np.diag(D * D) @ B.T - Cm) / np.linalg.norm(Cm)
        bad = (not np.all(np.isfinite(B))) or e1 > 1e-11 or e2 > 1e-12 or e3 > 1e-11
        for k, v in (("eig", e1), ("orth", e2), ("resid", e3)):
            if v > worst.get(k, (0,))[0]:
                worst[k] = (v, n, name)
        if bad:
            print("FAIL n=%d %s: eig %.2e orth %.2e resid %.2e" % (n, name, e1, e2, e3), flush=True)
print("worst:", worst)
