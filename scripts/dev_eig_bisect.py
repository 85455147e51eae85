import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
from bboptpy_amd import _ffi
n=int(sys.argv[1]); dbg=int(sys.argv[2])
g=b.ActiveCMAES(mfev=10**6,tol=1e-12,np=2*n,seed=1)
g.initialize(b.objectives.sphere,-np.ones(n),np.ones(n),np.zeros(n))
rng=np.random.default_rng(0); X=rng.normal(size=(n,3*n)); C=X@X.T/(3*n)
g.set_state("C",C); g.set_state("fev",[10**6]); g.set_state("eigenlastev",[0]); g.set_state("dbg",[float(dbg)]); g.set_state("eig_stamps",[1.0])
t=time.time(); g.phase(_ffi.PHASE_EIGEN); print("n",n,"dbg",dbg,"ok %.3fs"%(time.time()-t), "leaf guards", g.get_state("eig_stamps")[12:16], flush=True)
if dbg==0:
    B=g.get_state("B").reshape(n,n); D=g.get_state("D")
    print("resid",np.linalg.norm(B@np.diag(D*D)@B.T-C)/np.linalg.norm(C),"orth",np.linalg.norm(B.T@B-np.eye(n)))
if n <= 16:
    np.set_printoptions(precision=4, linewidth=200)
    print("D^2      ", D*D)
    print("eigvalsh ", np.linalg.eigvalsh(C))
    import scipy.linalg as sl
    H=sl.hessenberg(C); print("diag(T)  ", np.sort(np.diag(H)))
    print("BtB diag ", np.diag(B.T@B))
