"""dev: one handle of 256 populations against two handles of 128 driven from two host threads
(their kernels interleave on the GPU: the small latency-bound kernels of one in the shadow of the
other's sampler?)"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bboptpy_amd as bb

n, lam, steps = 128, 4096, 100
lo, up = -10 * np.ones(n), 10 * np.ones(n)

def make(P, seed):
    g = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=lam, seed=seed, populations=P, poll_every=steps)
    g.initialize(bb.objectives.rosenbrock, lo, up, np.random.default_rng(seed).uniform(-10, 10, (P, n)))
    g.run(10)
    return g

def timed(handles):
    ts = [threading.Thread(target=h.run, args=(steps,)) for h in handles]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return time.perf_counter() - t0

for split in ([256], [128, 128], [64, 64, 64, 64]):
    hs = [make(P, 11 + k) for k, P in enumerate(split)]
    dt = min(timed(hs) for _ in range(3))
    print(split, "%.3f ms/step  %.3g evals/s" % (dt / steps * 1e3, sum(split) * lam * steps / dt), flush=True)
    del hs
