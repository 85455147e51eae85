"""dev: how far do device and oracle drift when the oracle keeps its own eigendecomposition
(signs aligned only)?  prints per-generation relative errors"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
import bboptpy_amd as hip
from bboptpy_amd import _ffi
L = po.oracle()
def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
for n, lam, obj in [(128, 512, "rosenbrock"), (128, 4096, "ellipsoid"), (64, 200, "rosenbrock")]:
    rng = np.random.default_rng(n + lam)
    lo, up = -10. * np.ones(n), 10. * np.ones(n)
    guess = rng.uniform(-5, 5, n)
    g = hip.ActiveCMAES(mfev=10 ** 9, tol=1e-14, np=lam, seed=99)
    g.initialize(getattr(hip.objectives, obj), lo, up, guess)
    g.set_state("record_normals", [1.0])
    o = po.cma(L, "active", 10 ** 9, 1e-14, lam)
    o.set_rng(po.RNG_INJECT)
    o.init(obj, lo, up, guess)
    for gen in range(8):
        Bd = g.get_state("B").reshape(n, n); Bo = o.get("B").reshape(n, n)
        sg = np.sign(np.sum(Bd * Bo, axis=0))
        o.set("B", (Bo * sg[None, :]).ravel())
        eB = rel(Bd, Bo * sg[None, :])
        g.phase(_ffi.PHASE_SAMPLE_EVALUATE)
        o.inject_z(g.get_state("zlast")); o.step("sample"); o.step("evaluate_sort")
        eX = rel(g.get_state("arx"), o.get("arx"))
        g.phase(_ffi.PHASE_RANK)
        same = np.array_equal(g.get_state("fit_idx").astype(int), o.get("fit_idx").astype(int))
        for ph in (_ffi.PHASE_UPDATE, _ffi.PHASE_EIGEN, _ffi.PHASE_HISTORY_STOP): g.phase(ph)
        o.step("update_distribution"); o.step("update_history")
        D = o.get("D") ** 2
        print(n, lam, obj, "gen", gen, "B %.1e arx %.1e xmean %.1e C %.1e D %.1e same-rank %s mingap %.1e" % (
            eB, eX, rel(g.get_state("xmean"), o.get("xmean")),
            rel(np.tril(g.get_state("C").reshape(n, n)), np.tril(o.get("C").reshape(n, n))),
            rel(g.get_state("D"), o.get("D")), same, np.diff(D).min() / D.max()))
