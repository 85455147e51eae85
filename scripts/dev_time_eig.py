import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
n=int(sys.argv[1]) if len(sys.argv)>1 else 128
lam=int(sys.argv[2]) if len(sys.argv)>2 else 4096
alg=b.ActiveCMAES(mfev=2**31-1,tol=0.,np=lam,seed=1)
alg.initialize(b.objectives.rosenbrock,-10*np.ones(n),10*np.ones(n),np.random.default_rng(0).uniform(-10,10,n))
alg.run(30)
alg.set_state("eig_stamps",[1.0]); alg.run(3)
t=alg.get_state("eig_stamps")
us=lambda a,b_: (t[b_]-t[a])/100.
print("total eigen %.1f | tred %.1f accum %.1f | dc total %.1f" % (us(0,5), us(0,1), us(1,3), us(3,4)))
print("dc: copy/scale %.1f leaves %.1f merges L1 %.1f L2 %.1f L3 %.1f finalGEMM %.1f" % (us(16,17),us(17,18),us(18,19),us(19,20),us(20,21),us(22,23)))
print("top merge: sort %.1f deflate %.1f secular %.1f loewner %.1f order+F %.1f gemm %.1f" % (us(24,25),us(25,26),us(26,27),us(27,28),us(28,29),us(29,30)))
print("top secular iterations", t[31], "k/nd/nr?")
print("leaf batches (blk 0..3)", t[12:16])
print("raw stamps 0..5 (us from 0):", [round((t[i]-t[0])/100.,1) for i in range(6)])
