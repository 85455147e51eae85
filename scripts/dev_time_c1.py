import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bboptpy_amd as b
n, lam = 10, 20
for P in (1, 256, 4096):
    alg = b.ActiveCMAES(mfev=2**31-1, tol=0., np=lam, seed=1, populations=P)
    alg.initialize(b.objectives.rosenbrock, -10*np.ones(n), 10*np.ones(n), np.random.default_rng(0).uniform(-10,10,(P,n)))
    alg.run(50)
    t=time.time(); alg.run(500); dt=time.time()-t
    print("C1 n=10 lambda=20 P=%d: %.1f us/generation, %.3g evals/s" % (P, dt/500*1e6, P*lam*500/dt))
for P in (1, 4096):
    alg = b.ActiveCMAES(mfev=2**31-1, tol=0., np=lam, seed=1, populations=P)
    alg.initialize(b.objectives.rosenbrock, -10*np.ones(n), 10*np.ones(n), np.random.default_rng(0).uniform(-10,10,(P,n)))
    alg.run(20)
    alg.set_state("profile", [1.0]); alg.run(50)
    prof = alg.get_state("profile")
    names = ["sample", "rank", "whiten", "gram", "paths", "cov", "eigen", "post", "stop"]
    print("P=%d per-kernel us:" % P, {k: round(v, 1) for k, v in zip(names, prof[:len(names)])})
