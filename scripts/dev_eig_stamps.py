#!/usr/bin/env python3
"""dev tool (GPU box): the eigensolver's phase clocks (bbo_set "eig_stamps") for one n = 128
decomposition of a CMA-like covariance, in microseconds (wall_clock64 runs at 100 MHz)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bboptpy_amd import _ffi   # noqa: E402
if os.environ.get("BBO_LIB"):
    _ffi.LIB_PATH = os.path.abspath(os.environ["BBO_LIB"])
import bboptpy_amd as bb   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = bb.ActiveCMAES(mfev=10 ** 9, tol=0., np=4 * n, seed=3, populations=P)
g.initialize(bb.objectives.rosenbrock, -10 * np.ones(n), 10 * np.ones(n),
             np.random.default_rng(1).uniform(-10, 10, (P, n)) if P > 1 else np.random.default_rng(1).uniform(-10, 10, n))
g.run(30)                       # a covariance with some structure
g.set_state("eig_stamps", [1.0])
if len(sys.argv) > 3:                      # diagnostic bits (8192 / 16384: stop after merge level 1 / 2)
    g.set_state("dbg", [float(sys.argv[3])])
for rep in range(3):
    g.set_state("eig_stamps", [1.0])       # clears the slots
    for ph in range(5):
        g.phase(ph)
    st = g.get_state("eig_stamps")
    names = {0: "start", 1: "tred done", 2: "tred(global) done", 16: "dc start", 17: "dc setup",
             18: "leaves done", 19: "merge level(s) <top-1", 20: "merge level top-1",
             21: "merge top", 22: "merges done", 23: "reflectors/B done", 24: "mg start",
             25: "mg deflate", 26: "mg 26", 27: "mg secular", 28: "mg 28", 29: "mg 29",
             30: "mg end", 32: "mg sec prologue", 33: "mg sec loop (wave 0)", 34: "mg F fragments", 35: "mg product (wave 0)", 36: "refl T factors", 37: "refl panels"}
    t0 = min(v for v in st[:32] if v > 0)
    rows = sorted((v, i) for i, v in enumerate(st) if v > 0 and i < 38 and i not in (8, 9, 10, 12, 13, 14, 15, 31))
    print("rep %d" % rep)
    prev = t0
    for v, i in rows:
        print("   %-26s +%7.1f us   (at %7.1f)" % (names.get(i, "slot %d" % i), (v - prev) / 100.,
                                                    (v - t0) / 100.))
        prev = v
    print("   secular iterations (top merge): %d; leaf sweeps: %s" % (st[31], st[12:16]))
    if os.environ.get("SECULAR_HIST"):
        print("   secular iterations per root (1, 2, ... 12, more): %s; worst: %d iterations at root %d of %d" % (
            [int(v) for v in st[34:47]], st[47] // 1000000, (st[47] // 1000) % 1000, st[47] % 1000))
    elif len(st) >= 48 and any(st[40:48]):
        print("   step clocks (cycles, BBO_EIG_STEP_CLOCKS build): barrier1 %d, chain+u %d, L11 product %d, "
              "L22/L21 product %d, barrier2 %d, w %d, rank-2 %d, loop end %d" % tuple(st[40:48]))
