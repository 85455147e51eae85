"""dev (GPU box): the multi-workgroup Householder reduction (bbo_eig_mw.hpp) against the
one-workgroup one (diagnostic bit 16777216) on the same matrices: D, B up to sign, residuals."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import bboptpy_amd as hip   # noqa: E402
from bboptpy_amd import _ffi   # noqa: E402
from test_cma_gpu import _spd_cases   # noqa: E402

ns = [int(a) for a in sys.argv[1:]] or [129, 130, 160, 200, 255, 256]
worst = 0.
for n in ns:
    rng = np.random.default_rng(n)
    for name, Cm in _spd_cases(n, rng):
        Cm = 0.5 * (Cm + Cm.T)
        out = []
        for dbg in (0, 16777216):
            g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=max(4, 2 * n), seed=1)
            g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
            if dbg:
                g.set_state("dbg", [float(dbg)])
            g.set_state("C", Cm); g.set_state("fev", [10 ** 6]); g.set_state("eigenlastev", [0])
            g.phase(_ffi.PHASE_EIGEN)
            out.append((g.get_state("B").reshape(n, n).copy(), g.get_state("D").copy()))
        (B0, D0), (B1, D1) = out
        lam = np.linalg.eigvalsh(Cm)
        sc = np.abs(lam).max()
        e_d = np.abs(D0 * D0 - D1 * D1).max() / sc
        e_orth = np.linalg.norm(B0.T @ B0 - np.eye(n)) / n
        e_res = np.linalg.norm(B0 @ np.diag(D0 * D0) @ B0.T - Cm) / np.linalg.norm(Cm)
        worst = max(worst, e_d, e_orth, e_res)
        flag = "" if (e_d < 1e-11 and e_orth < 1e-12 and e_res < 1e-11 and np.all(np.isfinite(B0))) else "  <-- FAIL"
        print("n %3d %-12s D vs one-workgroup %.1e  orth %.1e  resid %.1e%s" % (n, name, e_d, e_orth, e_res, flag), flush=True)
print("worst", worst)
