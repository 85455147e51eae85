import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
n, lam = 128, 4096
alg=b.ActiveCMAES(mfev=2**31-1,tol=0.,np=lam,seed=1)
alg.initialize(b.objectives.rosenbrock,-10*np.ones(n),10*np.ones(n),np.random.default_rng(0).uniform(-10,10,n))
for gens in [28]+[1]*12:
    alg.run(gens)
    alg.set_state("eig_stamps",[1.0]); alg.run(1)
    w = alg.get_state("eig_work")
    slab = (n+32)**2+72
    base = slab + n*n + 3*128*128      # G + n^2 (Fg at a=0) + 3*128*128
    d = w[base:base+8+5*128]
    k = int(d[0]); rho = d[1]
    its = d[8:8+k]; dl = d[8+128:8+128+k]; w2 = d[8+256:8+256+k]; mu = d[8+384:8+384+k]; org=d[8+512:8+512+k]
    print("gen", gens, "k", k, "rho %.3g"%rho, "iters: max", its.max(), "mean %.2f"%its.mean(), "hist", np.bincount(its.astype(int)))
    slow = np.argsort(-its)[:6]
    gaps = np.diff(np.append(dl, dl[-1] + rho*w2.sum()))
    for j in slow:
        print("   j=%d its=%d gap=%.3g mu/gap=%.3g w2[j]=%.3g w2[j+1]=%.3g org-j=%d" % (j, its[j], gaps[j], mu[j]/gaps[j], w2[j], w2[min(j+1,k-1)], org[j]-j))
