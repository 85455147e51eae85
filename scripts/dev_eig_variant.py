"""dev: the eigensolver's special-matrix check (tests/test_cma_gpu.py) against a VARIANT build of the
library -- `python scripts/dev_eig_variant.py path/to/lib.so n [n ...]` -- printing the errors
instead of asserting, for A/B diagnosis of a change in bbo_eig*.hpp on the GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from bboptpy_amd import _ffi  # noqa: E402

lib = sys.argv[1]
if lib != "-":
    _ffi.LIB_PATH = os.path.abspath(lib)
import bboptpy_amd as hip  # noqa: E402
from test_cma_gpu import _spd_cases  # noqa: E402

for n in [int(v) for v in sys.argv[2:]]:
    rng = np.random.default_rng(n)
    g = hip.ActiveCMAES(mfev=10 ** 6, tol=1e-12, np=2 * n, seed=1)
    g.initialize(hip.objectives.sphere, -np.ones(n), np.ones(n), np.zeros(n))
    for name, Cm in _spd_cases(n, rng):
        Cm = 0.5 * (Cm + Cm.T)
        g.set_state("C", Cm)
        g.set_state("fev", [10 ** 6])
        g.set_state("eigenlastev", [0])
        if os.environ.get("EIG_DBG"):
            g.set_state("eig_stamps", [1.0])
        if os.environ.get("EIG_BITS"):
            g.set_state("dbg", [float(int(os.environ["EIG_BITS"]))])
        g.phase(_ffi.PHASE_EIGEN)
        if os.environ.get("EIG_DBG"):
            st = g.get_state("eig_stamps")
            import struct
            f = lambda v: struct.unpack("d", struct.pack("q", int(v)))[0]
            print("   unsorted %d  site2 flags %d  first: col %d a %d mid %d b %d dprev %r di %r" % (
                st[38], st[39], st[42], st[43], st[44], st[47], f(st[45]), f(st[46])))
        B = g.get_state("B").reshape(n, n)
        D = g.get_state("D")
        lam = np.linalg.eigvalsh(Cm)
        sc = np.abs(lam).max()
        print("%s n=%d %-14s val %.2e orth %.2e res %.2e" % (
            os.path.basename(lib), n, name,
            np.abs(D * D - np.maximum(lam, lam.max() / 1e14)).max() / sc,
            np.linalg.norm(B.T @ B - np.eye(n)) / n,
            np.linalg.norm(B @ np.diag(D * D) @ B.T - Cm) / np.linalg.norm(Cm)), flush=True)
