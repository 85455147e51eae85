import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
names = ["sample", "rank", "whiten", "gram", "paths", "cov", "eigen", "post", "stop"]
for n, lam, P in ((32, 64, 1024), (64, 128, 1024), (20, 40, 2048)):
    alg = b.ActiveCMAES(mfev=2**31-1, tol=0., np=lam, seed=1, populations=P)
    alg.initialize(b.objectives.rosenbrock, -10*np.ones(n), 10*np.ones(n), np.random.default_rng(0).uniform(-10,10,(P,n)))
    alg.run(20)
    t=time.time(); alg.run(100); dt=time.time()-t
    alg.set_state("profile", [1.0]); alg.run(50)
    prof = alg.get_state("profile")
    print("n=%d lambda=%d P=%d: %.1f us/generation, %.3g evals/s |" % (n, lam, P, dt/100*1e6, P*lam*100/dt), {k: round(prof[2*i]*1000/50) for i, k in enumerate(names)})
