"""dev: how long does a bounded C5 leg (ConcurrentBiPop n = 256 Rastrigin) take, and how many
rounds / regimes does it show, for a given inner tol and budget?  (sizes bench.py's bipop_scaling)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bboptpy_amd as bb
from bboptpy_amd.distributed import ConcurrentBiPop

n = 256
lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
guess = np.random.default_rng(7).uniform(-5.12, 5.12, n)
for tol, budget, slots in [(0.5, 40000, 1), (0.2, 40000, 1), (0.05, 60000, 1), (0.2, 40000 * 8, 8)]:
    drv = ConcurrentBiPop(mfev=budget, tol=tol, sigma0=2., seed=2024, device=0, variant="active",
                          slots_per_rank=slots, world_size=1, rank=0)
    t0 = time.perf_counter()
    drv.optimize(bb.objectives.rastrigin, lo, up, guess)
    dt = time.perf_counter() - t0
    st = drv.state
    print("tol %g budget %d slots %d: %.2f s, rounds %d, large %d small %d, fev %d, best %.3f"
          % (tol, budget, slots, dt, st.round, st.largerestarts, st.smallrestarts, st.fev, st.fxbest))
    for h in st.history:
        print("   ", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items()})
    sys.stdout.flush()
