"""dev (GPU box, BBO_MW_CLOCKS build through BBO_LIB): the six per-step clocks of the spread
reduction (bbo_eig_mw.hpp, MW_CK): 0 top of the step, 1 product + publish, 2 wait for the pieces,
3 load them, 4 w / next row / record / next reflector / rank-2 update, 5 hand-over to the next step.
    BBO_LIB=scripts/_variants/libmwclk.so python scripts/dev_mw_clocks.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bboptpy_amd import _ffi   # noqa: E402
if os.environ.get("BBO_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["BBO_LIB"])
import bboptpy_amd as bb   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = bb.ActiveCMAES(mfev=10 ** 9, tol=0., np=20, seed=3)
g.initialize(bb.objectives.rastrigin, -5.12 * np.ones(n), 5.12 * np.ones(n),
             np.random.default_rng(1).uniform(-5, 5, n))
g.run(20)
for rep in range(3):
    g.set_state("eig_stamps", [1.0])
    for ph in range(5):
        g.phase(ph)
    st = g.get_state("eig_stamps")
    ck = [float(v) for v in st[40:46]]
    tot = sum(ck)
    print("rep %d: per step (clock64 ticks, averaged over n - 1 = %d): " % (rep, n - 1)
          + "  ".join("%d: %.0f (%.0f%%)" % (k, v, 100 * v / max(tot, 1)) for k, v in enumerate(ck))
          + "   total %.0f" % tot)
