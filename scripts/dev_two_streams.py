"""dev (GPU box): the metric shape as ONE handle of P populations against TWO handles of P / 2 on
two streams driven from two host threads (does the latency-bound eigensolver of one half hide
behind the other half's sampler / Gram?):  python scripts/dev_two_streams.py [n lam P gens]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bboptpy_amd as bb   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lam = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P = int(sys.argv[3]) if len(sys.argv) > 3 else 256
gens = int(sys.argv[4]) if len(sys.argv) > 4 else 60


def make(p, seed):
    a = bb.ActiveCMAES(mfev=2 ** 31 - 1, tol=0., np=lam, seed=seed, populations=p, poll_every=gens)
    a.initialize(bb.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n),
                 np.random.default_rng(seed).uniform(-10, 10, (p, n)))
    a.run(10)
    return a


one = make(P, 1)
t = time.perf_counter()
one.run(gens)
t1 = time.perf_counter() - t
print("one handle, P = %d: %.4f ms/generation, %.3e evals/s" % (P, 1e3 * t1 / gens, P * lam * gens / t1))
del one
for parts in (2, 4):
    hs = [make(P // parts, 10 + k) for k in range(parts)]
    th = [threading.Thread(target=h.run, args=(gens,)) for h in hs]
    t = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    t2 = time.perf_counter() - t
    print("%d handles, P = %d each: %.4f ms/generation of all, %.3e evals/s" % (parts, P // parts, 1e3 * t2 / gens,
                                                                          P * lam * gens / t2))
    del hs
