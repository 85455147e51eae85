#!/usr/bin/env python3
"""dev tool (GPU box): per-launch time of the eigensolver kernels (HIP-event timers of the
engine) for a few shapes; `dbg` bits select older code paths for A/B runs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bboptpy_amd as bb   # noqa: E402
import bench               # noqa: E402

shapes = [(128, 4096, 256, 0), (128, 1024, 1, 0), (256, 20, 1, 0), (256, 20, 1, 1024),
          (256, 20, 32, 0), (200, 20, 1, 0), (200, 20, 1, 1024), (160, 64, 8, 0)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for n, lam, P, dbg in shapes:
    wl = dict(algo="ActiveCMAES", n=n, np=lam, objective="rosenbrock", box=(-10., 10.))
    lo, up = -10. * np.ones(n), 10. * np.ones(n)
    guess = np.random.default_rng(1).uniform(-10, 10, (P, n))
    alg = bench.make_optimizer(bb, wl, P, 5, 0, poll=1000)
    alg.initialize(bb.objectives.rosenbrock, lo, up, guess)
    if dbg:
        alg.set_state("dbg", [float(dbg)])
    alg.run(10)
    alg.set_state("profile", [1.0])
    steps = 30
    alg.run(steps)
    prof = alg.get_state("profile")
    out = []
    for i, name in enumerate(bench.CMA_KERNELS):
        ms, calls = prof[2 * i], prof[2 * i + 1]
        if calls > 0:
            out.append("%s %.1f" % (name.replace("cma_", ""), 1e3 * ms / calls))
    print("n=%d lam=%d P=%d dbg=%d | us per launch: %s" % (n, lam, P, dbg, ", ".join(out)))
    sys.stdout.flush()
