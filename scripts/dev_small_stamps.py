#!/usr/bin/env python3
"""dev tool (GPU box): phase clocks inside cma_small_generations (one population, C1 shape)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bboptpy_amd as bb   # noqa: E402

n, lam = 10, 20
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = bb.ActiveCMAES(mfev=10 ** 9, tol=0., np=lam, seed=3, populations=P)
g.initialize(bb.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n),
             np.random.default_rng(1).uniform(-10, 10, (P, n)))
g.run(40)
g.set_state("eig_stamps", [1.0])
for rep in range(3):
    g.run(8)
    st = g.get_state("eig_stamps")
    names = ["sample", "rank", "whiten", "gram", "paths", "cov", "eigen+post", "history/stop"]
    print("rep %d: " % rep + ", ".join("%s %.1f" % (names[i], (st[17 + i] - st[16 + i]) / 100.)
                                        for i in range(8)) + " | generation %.1f us" % ((st[24] - st[16]) / 100.))

print("eigen inside: tred %.1f, QL %.1f, back-transform + sort + post %.1f us" % (
    (st[26] - st[25]) / 100., (st[27] - st[26]) / 100., (st[28] - st[27]) / 100.))
g.set_state("dbg", [64.])          # the nine-kernel sequence
g.run(8)
st = g.get_state("eig_stamps")
print("kernel sequence, eigen inside: tred %.1f, QL %.1f, back-transform + sort + post %.1f us" % (
    (st[26] - st[25]) / 100., (st[27] - st[26]) / 100., (st[28] - st[27]) / 100.))
