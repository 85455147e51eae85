#!/usr/bin/env python3
"""oracle/gen_golden.py -- writes tests/golden/*.json from the REAL reference.

Run in the development container only (needs oracle/_ref/libbbo_ref.so, i.e. the reference
compiled from /root/reference by `make -C oracle ref`).  Each fixture records inputs (seed,
configuration, guess) and the reference's outputs (state after given generations, final
result); tests/test_oracle_golden.py replays the same inputs through the CPU oracle and
demands bit-identical numbers, tests/test_*_gpu.py compare the HIP path against the same
vectors where no randomness is involved.

Floats are stored as hex strings (float.hex) so JSON round-trips them exactly.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyoracle as po   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def hx(a):
    a = np.atleast_1d(np.asarray(a, dtype=np.float64)).ravel()
    return [float(v).hex() for v in a]


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    with open(path, "w") as fh:
        json.dump(obj, fh, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


def gen_rng(R):
    R.seed(123)
    raw = [int(R.f("draw_raw")()) for _ in range(16)]
    R.seed(123)
    uni = [R.f("draw_uniform")(-3., 7.) for _ in range(32)]
    R.seed(123)
    ints = [int(R.f("draw_int")(0, (k % 97) + 1)) for k in range(64)]
    R.seed(123)
    R.f("reset_test_normal")()
    nor = [R.f("draw_normal")() for _ in range(33)]
    dump("rng_mt19937.json", {"seed": 123, "raw": raw, "uniform_m3_7": hx(uni),
                              "int_0_kmod97p1": ints, "normal": hx(nor)})


CMA_SCALARS = ("mu", "mueff", "cc", "cs", "c1", "cmu", "damps", "chi", "eigenfreq", "hlen",
               "ik", "mit")


def gen_cma_constants(R):
    rows = []
    for variant in ("cmaes", "active"):
        for n, lam in ((10, 20), (128, 1024), (128, 4096), (256, 20), (256, 10240), (37, 50)):
            h = po.cma(R, variant, 10 ** 6, 1e-4, lam)
            h.init("sphere", -np.ones(n), np.ones(n), np.zeros(n))
            rec = {"variant": variant, "n": n, "lambda": lam}
            for k in CMA_SCALARS + (("cneg",) if variant == "active" else ()):
                rec[k] = hx(h.scalar(k))[0]
            w = h.get("weights")
            rec["w_head"] = hx(w[:4])
            rec["w_tail"] = hx(w[-1:])
            rows.append(rec)
            h.destroy()
    dump("cma_constants.json", rows)


CMA_STATE = ("xmean", "sigma", "pc", "ps", "C", "B", "D", "invsqrtC", "arx", "fit_val",
             "fit_idx", "it", "fev", "fbest", "fworst")


def gen_cma_runs(R):
    runs = []
    cases = [("active", 10, 20, "rosenbrock", 1, 10000, 1e-4),
             ("active", 10, 20, "rosenbrock", 2, 10000, 1e-4),
             ("cmaes", 10, 20, "rosenbrock", 3, 10000, 1e-4),
             ("active", 16, 12, "rastrigin", 4, 20000, 1e-6),
             ("active", 7, 9, "ellipsoid", 5, 5000, 1e-8)]
    for variant, n, lam, obj, seed, mfev, tol in cases:
        R.seed(seed)
        box = 5.12 if obj == "rastrigin" else 10.
        lo, up = -box * np.ones(n), box * np.ones(n)
        guess = np.random.default_rng(seed).uniform(-box, box, n)
        h = po.cma(R, variant, mfev, tol, lam)
        h.init(obj, lo, up, guess)
        rec = {"variant": variant, "n": n, "lambda": lam, "objective": obj, "seed": seed,
               "mfev": mfev, "tol": tol, "box": box, "guess": hx(guess), "states": [],
               "trace": []}
        z_first = h.peek_normals(3 * lam * n)   # the normals of generations 0..2
        rec["normals_first3"] = hx(z_first)
        gen = 0
        flag = 0
        while h.scalar("fev") < mfev:
            h.iterate()
            gen += 1
            if gen <= 3 or gen in (10, 50):
                rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in CMA_STATE}})
            D = h.get("D")
            rec["trace"].append(hx([h.get("fit_val")[0], h.scalar("sigma"),
                                    np.linalg.norm(h.get("xmean")), D[-1] / D[0]]))
            flag = h.converged()
            if flag:
                break
        x, fev, conv = h.solution()
        rec["result"] = {"generations": gen, "flag": flag, "fev": fev, "converged": conv,
                         "x": hx(x)}
        if len(rec["trace"]) > 120:   # keep the file small: head and tail of the trace
            rec["trace_head"] = rec["trace"][:60]
            rec["trace_tail"] = rec["trace"][-60:]
            del rec["trace"]
        runs.append(rec)
        h.destroy()
    dump("cma_runs.json", runs)


def gen_pop_runs(R):
    runs = []
    n = 8
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    cases = [("shade", dict(mfev=4000, npinit=16, tol=1e-8), ("x", "f", "arch", "MCR", "MF", "k",
                                                               "np", "fev")),
             ("jade", dict(mfev=4000, np_=16, tol=1e-8), ("x", "f", "arch", "mucr", "muf", "np",
                                                          "fev")),
             ("apso", dict(mfev=4000, tol=1e-8, np_=12), ("x", "v", "xb", "f", "fb", "xbest",
                                                          "fbest", "w", "c1", "c2", "state", "it",
                                                          "fev"))]
    for name, kw, keys in cases:
        for obj, seed in (("rastrigin", 11), ("rosenbrock", 12)):
            R.seed(seed)
            h = getattr(po, name)(R, **kw)
            h.init(obj, lo, up, np.zeros(n))
            rec = {"algo": name, "params": kw, "n": n, "objective": obj, "seed": seed,
                   "box": 5., "states": [{"gen": 0, **{k: hx(h.get(k)) for k in keys}}]}
            for gen in range(1, 31):
                h.iterate()
                if name == "apso" and not (0 <= h.scalar("state") <= 4):
                    # the reference read past its rule table (apso.cpp:384): undefined from
                    # here on; the fixture stops at the last well-defined generation
                    rec["ub_at_generation"] = gen
                    break
                if gen in (1, 2, 5, 30) or name == "apso":
                    rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in keys}})
            runs.append(rec)
            h.destroy()
    dump("pop_runs.json", runs)


def gen_restart_runs(R):
    runs = []
    for drv in ("bipop", "ipop"):
        for seed, n, obj in ((1, 6, "rastrigin"), (2, 6, "rosenbrock"), (3, 10, "rastrigin")):
            R.seed(seed)
            lo, up = -5. * np.ones(n), 5. * np.ones(n)
            guess = np.random.default_rng(seed).uniform(-5, 5, n)
            base = po.cma(R, "active", 1, 1e-6, 4)
            h = getattr(po, drv)(R, base, 60000)
            h.init(obj, lo, up, guess)
            # largelambda / largesigma / smalllambda / smallsigma are uninitialised members of
            # the reference until the first run of their regime: recorded only from then on
            keys = ("fev", "it", "fx", "xbest") + (
                ("largebudget", "smallbudget", "largerestarts", "smallrestarts", "bestregime",
                 "fxbest") if drv == "bipop" else ("lambda", "sigma", "fbest"))
            rec = {"driver": drv, "seed": seed, "n": n, "objective": obj, "mfev": 60000,
                   "guess": hx(guess), "schedule": [{k: hx(h.get(k)) for k in keys}]}
            for _ in range(30):
                if h.scalar("fev") >= 60000:
                    break
                if drv == "bipop" and h.scalar("largerestarts") >= 9:
                    break
                h.iterate()
                row = {k: hx(h.get(k)) for k in keys}
                if drv == "bipop" and h.scalar("largerestarts") > 0:
                    row["largelambda"] = hx(h.get("largelambda"))
                    row["largesigma"] = hx(h.get("largesigma"))
                if drv == "bipop" and h.scalar("smallrestarts") > 0:
                    row["smalllambda"] = hx(h.get("smalllambda"))
                    row["smallsigma"] = hx(h.get("smallsigma"))
                rec["schedule"].append(row)
            runs.append(rec)
            h.destroy()
    dump("restart_runs.json", runs)


def _capture_stdout(fn):
    """run fn() with the process-level stdout (fd 1: the reference prints through std::cout)
    redirected into a temporary file; returns the text"""
    import ctypes
    import tempfile
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        os.dup2(tmp.fileno(), 1)
        try:
            fn()
            libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode()


def gen_restart_print(R):
    """the `print=True` Tabular rows (tabular.hpp:65-77; bipop_cmaes.cpp:100-106,153-162;
    ipop_cmaes.cpp:104-109,158-160): the text the reference writes, next to the values it
    formatted"""
    import ctypes as C
    runs = []
    for drv, seed, n, obj in (("bipop", 7, 5, "rastrigin"), ("ipop", 8, 5, "rosenbrock")):
        R.f(drv + "_set_print").argtypes = [C.c_void_p, C.c_int]
        R.f(drv + "_set_print").restype = None
        R.seed(seed)
        lo, up = -5. * np.ones(n), 5. * np.ones(n)
        guess = np.random.default_rng(seed).uniform(-5, 5, n)
        base = po.cma(R, "active", 1, 1e-6, 4)
        h = getattr(po, drv)(R, base, 30000)
        R.f(drv + "_set_print")(h.ptr, 1)
        keys = ("it", "largerestarts", "smallrestarts", "largebudget", "smallbudget", "fev",
                "fx", "fxbest") if drv == "bipop" else ("it", "fev", "lambda", "sigma", "fx",
                                                        "fbest")
        rows = []

        def body():
            h.init(obj, lo, up, guess)
            rows.append({k: hx(h.get(k)) for k in keys})
            for _ in range(8):
                if h.scalar("fev") >= 30000:
                    break
                h.iterate()
                rows.append({k: hx(h.get(k)) for k in keys})

        text = _capture_stdout(body)
        runs.append({"driver": drv, "seed": seed, "n": n, "objective": obj, "mfev": 30000,
                     "lines": text.split("\n"), "values": rows})
        h.destroy()
    dump("restart_print.json", runs)


SEP_STATE = ("xmean", "sigma", "pc", "ps", "csep", "D", "arx", "fit_val", "fit_idx", "it",
             "fev", "fbest", "fworst")


def gen_sep_runs(R):
    """SepCmaes (sep_cmaes.cpp): strategy constants and full trajectories"""
    runs = []
    cases = [(10, 20, "ellipsoid", 21, 20000, 1e-8, False, False),
             (10, 20, "rosenbrock", 22, 20000, 1e-6, False, True),
             (16, 12, "rastrigin", 23, 20000, 1e-6, True, False),
             (40, 30, "discus", 24, 30000, 1e-8, False, True)]
    for n, lam, obj, seed, mfev, tol, bound, adjustlr in cases:
        R.seed(seed)
        box = 5.12 if obj == "rastrigin" else 10.
        lo, up = -box * np.ones(n), box * np.ones(n)
        guess = np.random.default_rng(seed).uniform(-box, box, n)
        h = po.cma(R, "sep", mfev, tol, lam, bound=bound, adjustlr=adjustlr)
        h.init(obj, lo, up, guess)
        rec = {"n": n, "lambda": lam, "objective": obj, "seed": seed, "mfev": mfev, "tol": tol,
               "bound": bound, "adjustlr": adjustlr, "box": box, "guess": hx(guess),
               "constants": {k: hx(h.scalar(k))[0] for k in ("mu", "mueff", "cc", "cs", "ccov",
                                                             "damps", "chi", "hlen", "ik", "mit")},
               "states": [], "trace": []}
        rec["normals_first3"] = hx(h.peek_normals(3 * lam * n))
        gen, flag = 0, 0
        while h.scalar("fev") < mfev:
            h.iterate()
            gen += 1
            if gen <= 3 or gen in (10, 50):
                rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in SEP_STATE}})
            D = h.get("D")
            rec["trace"].append(hx([h.get("fit_val")[0], h.scalar("sigma"),
                                    np.linalg.norm(h.get("xmean")), D.max() / D.min()]))
            flag = h.converged()
            if flag:
                break
        x, fev, conv = h.solution()
        rec["result"] = {"generations": gen, "flag": flag, "fev": fev, "converged": conv,
                         "x": hx(x)}
        if len(rec["trace"]) > 120:
            rec["trace_head"] = rec["trace"][:60]
            rec["trace_tail"] = rec["trace"][-60:]
            del rec["trace"]
        runs.append(rec)
        h.destroy()
    dump("sep_runs.json", runs)


SANSDE_KEYS = ("x", "f", "cr", "p", "fp", "crm", "crrec", "crdeltaf", "pns", "pnf", "fpns", "fpnf",
               "fev", "it")


def gen_sansde_runs(R):
    """SaNSDESearch (sansde.cpp): generations and the adaptation counters"""
    runs = []
    n = 8
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    cases = [("rastrigin", 41, dict(mfev=8000, np_=16, tol=1e-9)),
             ("rosenbrock", 42, dict(mfev=8000, np_=12, tol=1e-9, repaircr=False, crref=3,
                                     pupdate=7, crupdate=5)),
             ("sphere", 43, dict(mfev=8000, np_=10, tol=1e-9, pupdate=10, crupdate=4))]
    for obj, seed, kw in cases:
        R.seed(seed)
        h = po.sansde(R, **kw)
        h.init(obj, lo, up, np.zeros(n))
        rec = {"params": kw, "n": n, "objective": obj, "seed": seed, "box": 5.,
               "states": [{"gen": 0, **{k: hx(h.get(k)) for k in SANSDE_KEYS}}]}
        for gen in range(1, 61):
            h.iterate()
            if gen in (1, 2, 5, 10, 25, 50, 60):
                rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in SANSDE_KEYS}})
        x, fev, conv = h.solution()
        rec["result"] = {"x": hx(x), "fev": fev, "converged": conv}
        runs.append(rec)
        h.destroy()
    dump("sansde_runs.json", runs)


def gen_cso_runs(R):
    """CSOSearch (cso.cpp): swarm, velocities, means after generations 1, 2, 5, 20, 40"""
    runs = []
    n = 6
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    cases = [("rastrigin", 51, dict(mfev=10 ** 7, stol=1e-9, np_=12)),
             ("rosenbrock", 52, dict(mfev=10 ** 7, stol=1e-9, np_=13, pcompete=2)),
             ("sphere", 53, dict(mfev=10 ** 7, stol=1e-9, np_=20, pcompete=4, ring=True)),
             ("ackley", 54, dict(mfev=10 ** 7, stol=1e-9, np_=150, pcompete=2, ring=True,
                                 correct=False, vmax=0.1))]
    for obj, seed, kw in cases:
        R.seed(seed)
        h = po.cso(R, **kw)
        h.init(obj, lo, up, np.zeros(n))
        keys = ["x", "v", "f", "mean", "meanw", "xbest", "fbest", "fev"] + \
            (["pmean", "home"] if kw.get("ring") else [])
        rec = {"params": kw, "n": n, "objective": obj, "seed": seed, "box": 5., "keys": keys,
               "phil": hx(h.get("phil")), "phih": hx(h.get("phih")), "states": []}
        for gen in range(1, 41):
            h.iterate()
            if gen in (1, 2, 5, 20, 40):
                rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in keys}})
        runs.append(rec)
        h.destroy()
    dump("cso_runs.json", runs)


CCPSO_KEYS = ("x", "y", "yhat", "fx", "fy", "k", "ibest", "strat", "fyhat", "phat", "fev", "is",
              "nswarm", "cpswarm", "improved")


def gen_ccpso_runs(R):
    """CCPSOSearch (ccpso.cpp), no local optimizer: state after generations 1, 2, 5, 20, 40.
    (the reference prints _fyhat to stdout every generation: expect noise while this runs)"""
    runs = []
    cases = [(12, "rastrigin", 61, dict(mfev=10 ** 8, stol=1e-9, np_=8, pps=[2, 3, 6])),
             (20, "rosenbrock", 62, dict(mfev=10 ** 8, stol=1e-9, np_=10, pps=[5, 10],
                                         correct=False)),
             (16, "sphere", 63, dict(mfev=10 ** 8, stol=1e-9, np_=6, pps=[1, 2, 4, 8, 16]))]
    for n, obj, seed, kw in cases:
        R.seed(seed)
        lo, up = -5. * np.ones(n), 5. * np.ones(n)
        h = po.ccpso(R, **kw)
        h.init(obj, lo, up, np.zeros(n))
        rec = {"params": kw, "n": n, "objective": obj, "seed": seed, "box": 5., "states": []}
        for gen in range(1, 41):
            h.iterate()
            if gen in (1, 2, 5, 20, 40):
                rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in CCPSO_KEYS}})
        runs.append(rec)
        h.destroy()
    dump("ccpso_runs.json", runs)


def gen_ccpso_local_runs(R):
    """CCPSOSearch WITH its local optimizer (ccpso.cpp:116-118, 371-435; CMAES / ActiveCMAES on
    the swarm weights every `localfreq` generations): state after generations 1, 2, 4, 10, 20.
    One `pps` entry per case -- the reference only has defined behaviour while the number of
    swarms does not grow (Cmaes::init resize()s _b, _c without clearing, cmaes.cpp:53-54)."""
    runs = []
    cases = [(24, "rosenbrock", 71, "cmaes", 3, dict(mfev=10 ** 8, stol=1e-9, np_=8, pps=[4])),
             (20, "rastrigin", 72, "active", 2, dict(mfev=10 ** 8, stol=1e-9, np_=10, pps=[5])),
             (30, "ellipsoid", 73, "cmaes", 1, dict(mfev=10 ** 8, stol=1e-9, np_=12, pps=[6],
                                                    correct=False))]
    for n, obj, seed, variant, lf, kw in cases:
        R.seed(seed)
        lo, up = -5. * np.ones(n), 5. * np.ones(n)
        h = po.ccpso(R, local=po.cma(R, variant, 400, 1e-8, 8), localfreq=lf, **kw)
        h.init(obj, lo, up, np.zeros(n))
        rec = {"params": kw, "n": n, "objective": obj, "seed": seed, "box": 5.,
               "local": {"variant": variant, "mfev": 400, "tol": 1e-8, "np": 8, "localfreq": lf},
               "states": []}
        for gen in range(1, 21):
            h.iterate()
            if gen in (1, 2, 4, 10, 20):
                rec["states"].append({"gen": gen, **{k: hx(h.get(k)) for k in CCPSO_KEYS}})
        runs.append(rec)
        h.destroy()
    dump("ccpso_local_runs.json", runs)


def main():
    po.build_ref()
    R = po.reference()
    if R is None:
        sys.exit("the reference is not available here: fixtures can only be generated in the "
                 "development container")
    only = sys.argv[1] if len(sys.argv) > 1 else None   # e.g. "sep": regenerate one file
    gens = {"rng": gen_rng, "cma_constants": gen_cma_constants, "cma": gen_cma_runs,
            "pop": gen_pop_runs, "restart": gen_restart_runs,
            "restart_print": gen_restart_print, "sep": gen_sep_runs,
            "sansde": gen_sansde_runs, "cso": gen_cso_runs, "ccpso": gen_ccpso_runs,
            "ccpso_local": gen_ccpso_local_runs}
    for name, fn in gens.items():
        if only is None or only == name:
            fn(R)


if __name__ == "__main__":
    main()
