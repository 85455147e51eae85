import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
for algo, n, npop, P in (("SHADE", 10, 50, 4096), ("JADE", 10, 50, 4096), ("APSO", 10, 40, 2048), ("CSO", 10, 60, 2048)):
    lo, up = -5*np.ones(n), 5*np.ones(n)
    if algo == "SHADE": alg = b.SHADE(mfev=2**31-1, npinit=npop, tol=0., npmin=npop, seed=1, populations=P)
    elif algo == "JADE": alg = b.JADE(mfev=2**31-1, np=npop, tol=0., seed=1, populations=P)
    elif algo == "APSO": alg = b.APSO(mfev=2**31-1, tol=0., np=npop, seed=1, populations=P)
    else: alg = b.CSO(mfev=2**31-1, stol=0., np=npop, seed=1, populations=P)
    alg.initialize(b.objectives.rastrigin, lo, up, np.zeros((P, n)))
    alg.run(20)
    t=time.time(); alg.run(100); dt=time.time()-t
    alg.set_state("profile", [1.0]); alg.run(50)
    prof = alg.get_state("profile")
    print("%s n=%d np=%d P=%d: %.1f us/generation, %.3g evals/s | kernel us:" % (algo, n, npop, P, dt/100*1e6, P*npop*100/dt), [round(prof[2*i]*1000/50) for i in range(len(prof)//2)])
