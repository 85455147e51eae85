"""dev (GPU box): where the wall time of bench.py's bounded C5 leg goes (cProfile of the driver:
device runs against host-side bookkeeping)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as bb
from bboptpy_amd.distributed import ConcurrentBiPop
n = 256
lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
guess = np.random.default_rng(7).uniform(-5.12, 5.12, n)
for rep in range(2):
    drv = ConcurrentBiPop(mfev=40000, tol=0.5, sigma0=2., seed=2024, device=0, variant="active", slots_per_rank=1)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    drv.optimize(bb.objectives.rastrigin, lo, up, guess)
    pr.disable()
    print("wall %.3f s, history:" % (time.perf_counter() - t0))
    for h in list(getattr(drv.state, 'history', []))[:12]:
        print('   ', h)
    if rep == 1:
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
