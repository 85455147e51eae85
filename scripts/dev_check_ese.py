import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
for (n, npart) in ((8, 12), (40, 300), (130, 700)):
    alg = b.APSO(mfev=10**9, tol=0., np=npart, seed=5)
    alg.initialize(b.objectives.sphere, -5*np.ones(n), 5*np.ones(n), np.zeros(n))
    X0 = alg.get_state("x").reshape(npart, n).copy()
    alg.iterate()
    ws = alg.get_state("ws")
    D = np.sqrt(((X0[:, None, :] - X0[None, :, :])**2).sum(-1))
    want = D.sum(1) / (npart - 1)
    err = (ws - want) / want.max()
    bad = np.nonzero(np.abs(err) > 1e-12)[0]
    print(n, npart, "max rel err", np.abs(err).max(), "bad count", bad.size, "first bad", bad[:20], "err", err[bad[:8]])
