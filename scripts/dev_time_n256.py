import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
n = 256
for lam in (20, 640):
    alg=b.ActiveCMAES(mfev=2**31-1,tol=0.,np=lam,seed=1)
    alg.initialize(b.objectives.rastrigin,-5.12*np.ones(n),5.12*np.ones(n),np.random.default_rng(0).uniform(-5,5,n))
    alg.run(10)
    t=time.time(); alg.run(100); dt=time.time()-t
    print("n=256 lambda=%d: %.2f ms/generation, %.3g evals/s" % (lam, dt*10, lam*100/dt))
