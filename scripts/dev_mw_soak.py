"""dev (GPU box): many engines sharing one GPU, each with the spread reduction (n = 256): S slots of
concurrent BIPOP restarts; reports throughput and how many engines had a wavefront give up"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as bb
from bboptpy_amd.distributed import ConcurrentBiPop
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 256
lo, up = -5.12 * np.ones(n), 5.12 * np.ones(n)
guess = np.random.default_rng(7).uniform(-5.12, 5.12, n)
drv = ConcurrentBiPop(mfev=20000 * S, tol=0.5, sigma0=2., seed=2024, device=0, variant="active", slots_per_rank=S)
t0 = time.perf_counter()
drv.optimize(bb.objectives.rastrigin, lo, up, guess)
dt = time.perf_counter() - t0
off = sum(int(a.get_state("eig_mw_off")[0]) for a in drv._algs.values())
fail = sum(int(a.get_state("eig_mw_fail")[0]) for a in drv._algs.values())
print("slots %d: %.2f s, %.3e evaluations/s, rounds %d, engines %d, gave up %d, switched off %d"
      % (S, dt, drv.state.fev / dt, drv.state.round, len(drv._algs), fail, off))
