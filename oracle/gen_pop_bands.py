#!/usr/bin/env python3
"""oracle/gen_pop_bands.py -- TEST INFRASTRUCTURE: writes tests/golden/pop_bands.json from the
REAL reference (oracle/_ref/libbbo_ref.so, compiled from /root/reference by `make -C oracle ref`).

Why: the device runs DE / PSO generation-SYNCHRONOUSLY, the reference replaces individuals in
place inside its loop (shade.cpp:181-183, jade.cpp:175-176) and refreshes gbest inside the
particle loop (apso.cpp:194-197).  tests/test_bands_gpu.py ties the two by outcome at n = 10,
np = 30-60 only; at the sizes of BASELINE.json's configs (np = 4096, n = 128 / 512) the
in-generation propagation is another regime.  This script records, for K independently seeded
runs of the compiled reference at CONFIG SCALE, the best objective value reached at fixed
evaluation checkpoints (a convergence curve per run), and stores per checkpoint the quartiles of
log10(best f) over the K runs.  tests/test_pop_bands_gpu.py runs the device on the same problems
(`populations=K`) and holds its medians against these bands.

APSO and the reference's out-of-bounds read.  `APSOSearch::nextState` indexes its rule table with
the state 1..4 (apso.cpp:384: `_rulebase[r][_state]`, rows of four ints), so state 4 reads one int
past the row: undefined behaviour.  In a fresh heap that word is 0 (SURVEY appendix A.10 -- the
value the oracle and the device pin), in a recycled one it is whatever an earlier allocation left,
and the run continues with a garbage state or dies (seen here: the second APSO run of one process
segfaults at iteration 6; AddressSanitizer names apso.cpp:384).  Every APSO seed therefore runs in
a process of its own; a seed whose process dies is recorded under `reference_crashed_seeds` and
the next seed takes its place; and every curve that goes into the fixture is compared with the
oracle's reference mode (async, mt19937, out-of-bounds word = 0): `equals_oracle` says whether
they agree to the last bit, i.e. whether that reference run read a 0 there.

A fixture is data: seeds, configuration, checkpoints and the reference's outputs.  Nothing of the
reference's text goes into it.  Run in the development container only:

    python oracle/gen_pop_bands.py            # ~ 8 minutes of one core
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pyoracle as po   # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "pop_bands.json")
K = 8

# name -> configuration.  `budget_gens`: mfev = budget_gens * np (APSO: (1 + np) per iteration,
# apso.cpp:68); `every`: a checkpoint each `every` * np evaluations.
CASES = {
    # C2 as BASELINE.json words it, with the class's default population-size reduction (npmin=4)
    "shade_lpsr": dict(algo="shade", n=128, np=4096, objective="rastrigin", box=(-5.12, 5.12),
                       budget_gens=600, every=10, npmin=4),
    # C2 as bench.py times it: reduction off (npmin = npinit), steady population
    "shade_fixed": dict(algo="shade", n=128, np=4096, objective="rastrigin", box=(-5.12, 5.12),
                        budget_gens=300, every=10, npmin=4096),
    "jade": dict(algo="jade", n=128, np=4096, objective="rosenbrock", box=(-10., 10.),
                 budget_gens=600, every=10),
    # C4's shape at an np the reference's O(np^2 n) generation allows on one core
    "apso": dict(algo="apso", n=512, np=1024, objective="sphere", box=(-10., 10.),
                 budget_gens=40, every=1),
}


def mfev_of(c):
    per = c["np"] + 1 if c["algo"] == "apso" else c["np"]
    return c["budget_gens"] * per


def checkpoints_of(c):
    per = c["np"] + 1 if c["algo"] == "apso" else c["np"]
    return [k * per for k in range(c["every"], c["budget_gens"] + 1, c["every"])]


def make(lib, c):
    mfev = mfev_of(c)
    if c["algo"] == "shade":
        return po.shade(lib, mfev, c["np"], 0., npmin=c["npmin"])
    if c["algo"] == "jade":
        return po.jade(lib, mfev, c["np"], 0.)
    return po.apso(lib, mfev, 0., c["np"])


def best_f(h, c):
    if c["algo"] == "apso":
        return h.scalar("fbest")
    return float(np.min(h.get("f")))


def curve(lib, c, seed):
    """best f at the first generation whose evaluation count reaches each checkpoint"""
    n = c["n"]
    lo, up = c["box"][0] * np.ones(n), c["box"][1] * np.ones(n)
    lib.seed(seed)
    h = make(lib, c)
    h.init(c["objective"], lo, up, np.zeros(n))      # DE / PSO ignore the guess
    cps = checkpoints_of(c)
    out, fevs = [], []
    k = 0
    mfev = mfev_of(c)
    while k < len(cps):
        h.iterate()
        fev = int(h.scalar("fev"))
        while k < len(cps) and fev >= cps[k]:
            out.append(best_f(h, c))
            fevs.append(fev)
            k += 1
        if fev >= mfev and k < len(cps):      # budget spent one generation early (LPSR)
            while k < len(cps):
                out.append(best_f(h, c))
                fevs.append(fev)
                k += 1
    h.destroy()
    return out, fevs


def curve_in_child(name, seed):
    """the same in a process of its own (APSO: see the header); None when the child died"""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, str(seed)],
                       capture_output=True, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("[")]
    if r.returncode != 0 or not lines:
        return None
    return [float.fromhex(v) for v in json.loads(lines[-1])]


def main():
    lib = po.reference()
    if lib is None:
        sys.exit("oracle/_ref/libbbo_ref.so is not built: run `make -C oracle ref` where "
                 "/root/reference exists")
    if len(sys.argv) == 4 and sys.argv[1] == "--one":
        f, _ = curve(lib, CASES[sys.argv[2]], int(sys.argv[3]))
        print(json.dumps([float(v).hex() for v in f]))
        return
    only = sys.argv[1:]
    doc = {"generator": "oracle/gen_pop_bands.py", "library": "oracle/_ref (compiled reference)",
           "runs_per_case": K, "seed_base": 9000, "cases": {}}
    if only and os.path.exists(OUT):
        with open(OUT) as fh:
            doc = json.load(fh)
    for name, c in CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        curves, seeds, crashed, equal = [], [], [], []
        seed = 9000 + 100 * list(CASES).index(name)
        while len(curves) < K and len(crashed) < 3 * K:
            if c["algo"] == "apso":
                f = curve_in_child(name, seed)
            else:
                f, _ = curve(lib, c, seed)
            if f is None:
                crashed.append(seed)
                print(name, "seed", seed, "the reference's process died (apso.cpp:384)", flush=True)
                seed += 1
                continue
            # the oracle's reference mode on the same seed (all APSO curves; one per DE case)
            if c["algo"] == "apso" or not curves:
                fo, _ = curve(po.oracle(), c, seed)
                equal.append(bool(np.array_equal(np.array(f), np.array(fo))))
            curves.append(f)
            seeds.append(seed)
            print(name, "seed", seed, "%.1f s" % (time.time() - t0), "final f %.6g" % f[-1],
                  "equals oracle:", equal[-1] if equal else None, flush=True)
            seed += 1
        lg = np.log10(np.maximum(np.array(curves), 1e-300))
        q = np.percentile(lg, [0, 25, 50, 75, 100], axis=0)
        doc["cases"][name] = {
            "config": dict(c, box=list(c["box"]), mfev=mfev_of(c), tol=0.),
            "checkpoints_fev": checkpoints_of(c),
            "log10_best_f": {"min": q[0].tolist(), "q1": q[1].tolist(), "median": q[2].tolist(),
                             "q3": q[3].tolist(), "max": q[4].tolist()},
            "runs_log10_best_f": lg.tolist(),
            "seeds": seeds, "reference_crashed_seeds": crashed, "equals_oracle": equal,
            "reference_seconds_one_core": time.time() - t0,
        }
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
