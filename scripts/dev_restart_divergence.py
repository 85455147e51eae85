#!/usr/bin/env python3
"""dev tool (GPU box): where do the device's and the oracle's restart schedules part?
Follows the oracle BIPOP/IPOP driver run by run, replays every inner run on one device engine
(set_params + set_seed + optimize, as bbo_restart.hip does) and, at the first run whose
evaluation count differs, replays that run generation by generation on both sides."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import bboptpy_amd as hip   # noqa: E402
import pyoracle as po       # noqa: E402

GOLDEN = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def main(driver="bipop", variant="active", n=6, obj="rastrigin", seed=21, mfev=40000, tol=1e-6):
    O = po.oracle()
    lo, up = -5. * np.ones(n), 5. * np.ones(n)
    guess = np.random.default_rng(seed).uniform(-5, 5, n)
    nflag = os.environ.get("NFLAG", "1") != "0"
    kw = {"nbipop": nflag} if driver == "bipop" else {"nipop": nflag}
    o = getattr(po, driver)(O, po.cma(O, variant, 1, tol, 4), mfev, **kw)
    o.set_mode(False, po.RNG_PHILOX, seed)
    o.init(obj, lo, up, guess)
    runs = [(int(o.scalar("last_lambda")), o.scalar("last_sigma"), int(o.scalar("last_inner_fev")),
             guess.copy(), o.scalar("fx"))]
    fevs = [0, int(o.scalar("fev"))]
    caps = [None]
    for _ in range(12):
        if o.scalar("fev") >= mfev:
            break
        lb = int(o.scalar("largebudget")) if driver == "bipop" else 0
        o.iterate()
        caps.append(lb >> 1 if driver == "bipop" and int(o.scalar("last_regime")) == 2 else None)
        runs.append((int(o.scalar("last_lambda")), o.scalar("last_sigma"),
                     int(o.scalar("last_inner_fev")), o.get("x0").copy(), o.scalar("fx")))
        fevs.append(int(o.scalar("fev")))
    cls = hip.ActiveCMAES if variant == "active" else hip.CMAES
    g = cls(mfev=1, tol=tol, np=4)
    oc = po.cma(O, variant, 1, tol, 4)       # a second oracle engine, replayed in step
    for r, (lam, sig, used, x0, fx) in enumerate(runs):
        maxit = int(100. + 50. * (n + 3) * (n + 3) / np.sqrt(1. * lam))
        maxfev = min(maxit * lam, mfev - fevs[r])
        if caps[r] is not None:
            maxfev = min(maxfev, caps[r])      # small regime: at most half the large budget
        sd = (seed + GOLDEN * r) & M64
        g.set_params(lam, sig, maxfev)
        g.set_seed(sd)
        sol = g.optimize(getattr(hip.objectives, obj), lo, up, x0)
        flag_d = int(g.get_state("flag")[0])
        oc.step("set_params", lam, float(sig), maxfev)
        oc.set_rng(po.RNG_PHILOX, sd)
        xo, fevo, _ = oc.optimize(obj, lo, up, x0)
        print("run %d lam=%d sigma=%.4g cap=%d: driver-oracle used %d | engine-oracle %d flag %d | "
              "device %d flag %d" % (r, lam, sig, maxfev, used, fevo, int(oc.scalar("flag")),
                                     sol.n_evals, flag_d), flush=True)
        force = int(os.environ.get("REPLAY_RUN", "-1"))
        print("     f*: driver-oracle %.6e engine-oracle %.6e device %.6e" % (
            fx, O.objective(obj, xo), hip.objectives.__dict__[obj](sol.x)), flush=True)
        if sol.n_evals != fevo or r == force:
            print("  -> generation-by-generation replay of run %d" % r, flush=True)
            # rebuild both engines' pre-run state by replaying the earlier runs
            g2 = cls(mfev=1, tol=tol, np=4)
            o2 = po.cma(O, variant, 1, tol, 4)
            for q in range(r):
                lq, sq, _, xq, _ = runs[q]
                mq = min(int(100. + 50. * (n + 3) * (n + 3) / np.sqrt(1. * lq)) * lq,
                         mfev - fevs[q])
                if caps[q] is not None:
                    mq = min(mq, caps[q])
                g2.set_params(lq, sq, mq)
                g2.set_seed((seed + GOLDEN * q) & M64)
                g2.optimize(getattr(hip.objectives, obj), lo, up, xq)
                o2.step("set_params", lq, float(sq), mq)
                o2.set_rng(po.RNG_PHILOX, (seed + GOLDEN * q) & M64)
                o2.optimize(obj, lo, up, xq)
            g2.set_params(lam, sig, maxfev)
            g2.set_seed(sd)
            g2.initialize(getattr(hip.objectives, obj), lo, up, x0)
            o2.step("set_params", lam, float(sig), maxfev)
            o2.set_rng(po.RNG_PHILOX, sd)
            o2.init(obj, lo, up, x0)
            Bd, Bo = g2.get_state("B").reshape(n, n), o2.get("B").reshape(n, n)
            print("   after init: d B %.2e  d|B| %.2e  dC(lower) %.2e  dD %.2e  d isc %.2e" % (
                np.abs(Bd - Bo).max(), np.abs(np.abs(Bd) - np.abs(Bo)).max(),
                np.abs(np.tril(g2.get_state("C").reshape(n, n)) - np.tril(o2.get("C").reshape(n, n))).max(),
                np.abs(g2.get_state("D") - o2.get("D")).max(),
                np.abs(g2.get_state("invsqrtC") - o2.get("invsqrtC")).max()), flush=True)
            print("   column sign agreement:", np.sign(np.sum(Bd * Bo, axis=0)), flush=True)
            np.set_printoptions(precision=4, linewidth=150)
            print("   B device:\n", Bd, "\n   B oracle:\n", Bo, flush=True)
            for gen in range(1, 2000):
                g2.iterate()
                o2.iterate()
                dx = np.abs(g2.get_state("xmean") - o2.get("xmean")).max()
                ds = abs(g2.get_state("sigma")[0] - o2.scalar("sigma")) / o2.scalar("sigma")
                Cg = np.tril(g2.get_state("C").reshape(n, n))
                Co = np.tril(o2.get("C").reshape(n, n))
                dC = np.abs(Cg - Co).max() / np.abs(Co).max()
                rk = int(np.sum(g2.get_state("fit_idx") != o2.get("fit_idx")))
                Dd, Do = g2.get_state("D"), o2.get("D")
                dB = np.abs(np.abs(g2.get_state("B")) - np.abs(o2.get("B"))).max()
                fd, fo = int(g2.get_state("flag")[0]), o2.converged()
                if gen <= 5 or gen % 4 == 0 or rk or fd or fo:
                    print("   gen %4d rankdiff %3d dx %.2e dsigma %.2e dC %.2e d|B| %.2e cond %.2e flags %d %d "
                          "f %.6e %.6e" % (gen, rk, dx, ds, dC, dB, (Do[-1] / Do[0]) ** 2, fd, fo,
                                           g2.get_state("fit_val")[0], o2.get("fit_val")[0]),
                          flush=True)
                if fd or fo or ds > 1e-3:
                    break
            return


if __name__ == "__main__":
    a = sys.argv[1:]
    if a:
        main(a[0], a[1], int(a[2]), a[3], int(a[4]))
    else:
        main()
