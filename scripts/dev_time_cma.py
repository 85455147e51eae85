import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
n=int(sys.argv[1]) if len(sys.argv)>1 else 128
lam=int(sys.argv[2]) if len(sys.argv)>2 else 4096
P=int(sys.argv[3]) if len(sys.argv)>3 else 1
gens=int(sys.argv[4]) if len(sys.argv)>4 else 200
alg=b.ActiveCMAES(mfev=2**31-1,tol=1e-30,np=lam,seed=1,populations=P)
g=np.random.default_rng(0).uniform(-10,10,(P,n))
alg.initialize(b.objectives.rosenbrock,-10*np.ones(n),10*np.ones(n),g)
alg.run(20)
t=time.time(); d=alg.run(gens); dt=time.time()-t
print("n",n,"lam",lam,"P",P,"gens",d,"ms/gen",1e3*dt/d,"evals/s %.3e"%(P*lam*d/dt), "sigma",alg.get_state("sigma")[0], "fbest", alg.get_state("fit_val")[0])
alg.set_state("eig_stamps",[1.0])
alg.run(3)
t=alg.get_state("eig_stamps")
print("eigen phases (us at 100MHz): load %.1f tred %.1f accum %.1f ql %.1f sort/out %.1f" % tuple((t[i+1]-t[i])/100. for i in range(5)))
alg.set_state("dbg",[1.0]); alg.run(3)
t=alg.get_state("eig_stamps")
print("dbg=1 (no apply): tred+accum %.1f ql %.1f" % ((t[3]-t[0])/100., (t[4]-t[3])/100.))
print("sweeps %d pairs %d epochs %d" % (t[8], t[9], t[10]))
