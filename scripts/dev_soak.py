"""one-off robustness sweep: random engine / size / batch combinations, short runs, everything
finite, evaluation counters consistent (not a test: run by hand on the GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bboptpy_amd as b
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
objs = [b.objectives.sphere, b.objectives.rosenbrock, b.objectives.rastrigin, b.objectives.ellipsoid, b.objectives.ackley, b.objectives.griewank, b.objectives.schwefel12]
t0 = time.time(); runs = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 60:
    algo = rng.choice(["CMAES", "ActiveCMAES", "SepCMAES", "SHADE", "JADE", "SANSDE", "APSO", "CSO"])
    n = int(rng.choice([1, 2, 3, 7, 10, 16, 17, 31, 32, 33, 50, 64, 65, 100, 127, 128, 129, 150, 200, 255, 256, 257, 300, 384, 512]))
    P = int(rng.choice([1, 2, 3, 4, 5, 9, 33, 64]))
    if n > 200: P = min(P, 9)
    f = objs[rng.integers(len(objs))]
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    kw = dict(seed=int(rng.integers(1 << 30)), populations=P)
    try:
        if algo in ("CMAES", "ActiveCMAES", "SepCMAES"):
            lam = int(rng.choice([4, 7, 16, 20, 50, 64, 65, 130, 300]))
            if algo == "SepCMAES" and n < 2: continue
            # the reference's uncapped (n+2)/3 learning-rate adjustment: c_cov > 1 there
            if algo == "SepCMAES" and lam > 4 * n: continue
            a = getattr(b, algo)(mfev=10**7, tol=1e-12, np=lam, **kw)
            key, per = "xmean", lam
        elif algo == "SHADE":
            npp = int(rng.choice([8, 20, 50, 64, 65, 200, 300])); a = b.SHADE(mfev=10**7, npinit=npp, tol=1e-12, **kw); key, per = "x", None
        elif algo == "JADE":
            npp = int(rng.choice([8, 20, 50, 64, 65, 200, 300])); a = b.JADE(mfev=10**7, np=npp, tol=1e-12, **kw); key, per = "x", None
        elif algo == "SANSDE":
            npp = int(rng.choice([8, 20, 50, 64, 65, 200])); a = b.SANSDE(mfev=10**7, np=npp, tol=1e-12, **kw); key, per = "x", None
        elif algo == "APSO":
            npp = int(rng.choice([8, 20, 50, 64, 65, 200])); a = b.APSO(mfev=10**7, tol=1e-12, np=npp, **kw); key, per = "x", None
        else:
            npp = int(rng.choice([9, 21, 60, 66, 201])); a = b.CSO(mfev=10**7, stol=1e-12, np=npp, **kw); key, per = "x", None
        a.initialize(f, lo, up, rng.uniform(-4, 4, (P, n)))
        gens = int(rng.integers(3, 40))
        done = a.run(gens)
        for p in range(P):
            x = a.get_state(key, p)
            assert np.all(np.isfinite(x)), (algo, n, P, f.name, "non-finite state")
            s = a.solution(p)
            assert np.all(np.isfinite(s.x)) and s.n_evals > 0
        runs += 1
    except AssertionError as e:
        print("FAIL", e, flush=True)
    except Exception as e:
        print("EXC", algo, n, P, f.name, type(e).__name__, str(e)[:120], flush=True)
print("soak: %d runs in %.0f s" % (runs, time.time() - t0))
