"""dev (CPU, numpy): the secular-equation iteration of bbo_eig_dc.hpp restated in numpy, to count
iterations per root on the merges of realistic matrices and to try other first guesses / steps
before building them into the kernel.   python scripts/dev_secular_model.py [variant ...]"""
import sys

import numpy as np
import scipy.linalg as sl

EPS = 2.0 ** -52


def merges_of(C, leaf=8):
    """(d, z, rho) of every merge of the divide and conquer of tridiag(C), by level (0 = lowest)."""
    n = C.shape[0]
    H = sl.hessenberg(C)
    d = np.diag(H).copy()
    e = np.diag(H, -1).copy()
    am = max(np.abs(d).max(), np.abs(e).max())
    sc = 2.0 ** (1 - np.frexp(am)[1])
    d *= sc
    e *= sc
    nb = 1
    while (n + nb - 1) // nb > leaf:
        nb *= 2
    bounds = [(i * n) // nb for i in range(nb + 1)]
    for t in range(1, nb):
        r = abs(e[bounds[t] - 1])
        d[bounds[t] - 1] -= r
        d[bounds[t]] -= r
    blocks = []
    for t in range(nb):
        a, b = bounds[t], bounds[t + 1]
        T = np.diag(d[a:b]) + np.diag(e[a:b - 1], 1) + np.diag(e[a:b - 1], -1)
        w, Q = np.linalg.eigh(T)
        blocks.append((a, b, w, Q))
    out = []
    level = 0
    while len(blocks) > 1:
        nxt = []
        lv = []
        for i in range(0, len(blocks) - 1, 2):
            a, mid, w1, Q1 = blocks[i]
            _, b, w2, Q2 = blocks[i + 1]
            rho_in = e[mid - 1]
            sgn = 1. if rho_in >= 0 else -1.
            z = np.concatenate([Q1[-1, :], sgn * Q2[0, :]])
            dd = np.concatenate([w1, w2])
            zn2 = (z * z).sum()
            z = z / np.sqrt(zn2)
            rho = abs(rho_in) * zn2
            lv.append((dd.copy(), z.copy(), rho))
            m = b - a
            Qb = np.zeros((m, m))
            Qb[:mid - a, :mid - a] = Q1
            Qb[mid - a:, mid - a:] = Q2
            M = np.diag(dd) + rho * np.outer(z, z)
            w, V = np.linalg.eigh(M)
            nxt.append((a, b, w, Qb @ V))
        if len(blocks) % 2:
            nxt.append(blocks[-1])
        out.append(lv)
        blocks = nxt
        level += 1
    return out


def deflate(dd, z, rho):
    o = np.argsort(dd, kind="stable")
    dS, zS = dd[o].copy(), z[o].copy()
    tol = 8 * EPS * max(np.abs(dS).max(), np.abs(zS).max())
    kept = []
    pj = -1
    for j in range(len(dS)):
        if rho * abs(zS[j]) <= tol:
            continue
        if pj < 0:
            pj = j
            continue
        t = dS[j] - dS[pj]
        if abs(t * zS[j] * zS[pj]) <= tol * (zS[j] ** 2 + zS[pj] ** 2):
            tau = np.hypot(zS[j], zS[pj])
            c, s = zS[j] / tau, -zS[pj] / tau
            zS[j] = tau
            zS[pj] = 0.
            tt = dS[pj] * c * c + dS[j] * s * s
            dS[j] = dS[pj] * s * s + dS[j] * c * c
            dS[pj] = tt
            pj = j
        else:
            kept.append(pj)
            pj = j
    if pj >= 0:
        kept.append(pj)
    kept = np.array(kept, dtype=int)
    dl = dS[kept]
    oo = np.argsort(dl, kind="stable")
    return dl[oo], (zS[kept] ** 2)[oo]


def solve(dl, w2, rho, variant="device"):
    """iterations per root (the count the kernel's loop would make), and the roots' offsets"""
    k = len(dl)
    its = np.zeros(k, dtype=int)
    mus = np.zeros(k)
    dorgs = np.zeros(k)
    if k < 2:
        return its, mus, dorgs
    for j in range(k):
        last = j == k - 1
        dj = dl[j]
        dn = dj + rho * w2.sum() if last else dl[j + 1]
        midp = 0.5 * (dj + dn)
        fm = 1. + rho * (w2 / (dl - midp)).sum()
        left = fm > 0 or last
        o = min(j if left else j + 1, k - 1)
        dorg = dl[o]
        gap = dn - dj
        lo = 0. if left else -0.5 * gap
        hi = (gap if last else 0.5 * gap) if left else 0.
        mu = 0.5 * (lo + hi)
        delta = dl - dorg
        if variant.startswith("dev"):
            wo = w2[o]
            rest1 = fm + (2. if left else -2.) * rho * wo / gap
            g0 = rho * wo / rest1
            if g0 == g0 and lo < g0 < hi:
                mu = g0
        elif variant.startswith("two"):
            # dlaed4's first guess: the two neighbouring poles with their true weights, the rest
            # frozen at the midpoint
            if not last:
                dlp0 = dj - dorg
                drp0 = dn - dorg
                # rest at the midpoint, without poles j, j+1
                c = fm - rho * w2[j] / (dj - midp) - rho * w2[j + 1] / (dn - midp)
                # c + rho w_j/(dlp0 - x) + rho w_j1/(drp0 - x) = 0
                aa, bb = rho * w2[j], rho * w2[j + 1]
                # c (dlp0 - x)(drp0 - x) + aa (drp0 - x) + bb (dlp0 - x) = 0
                A2 = c
                A1 = -(c * (dlp0 + drp0) + aa + bb)
                A0 = c * dlp0 * drp0 + aa * drp0 + bb * dlp0
                disc = max(A1 * A1 - 4 * A2 * A0, 0.)
                qq = -0.5 * (A1 + (1. if A1 >= 0 else -1.) * np.sqrt(disc))
                cands = []
                if A2 != 0:
                    cands.append(qq / A2)
                if qq != 0:
                    cands.append(A0 / qq)
                for g0 in cands:
                    if g0 == g0 and lo < g0 < hi:
                        mu = g0
                        break
            else:
                wo = w2[o]
                rest1 = fm + 2. * rho * wo / gap
                g0 = rho * wo / rest1
                if g0 == g0 and lo < g0 < hi:
                    mu = g0
        swtch = False
        fprev = None
        for it in range(64):
            r = 1. / (delta - mu)
            t = w2 * r
            low = np.arange(k) <= j
            psi = rho * t[low].sum()
            dpsi = rho * (t[low] * r[low]).sum()
            phi = rho * t[~low].sum()
            dphi = rho * (t[~low] * r[~low]).sum()
            fabs_ = np.abs(t).sum()
            f = 1. + psi + phi
            err = 8. * EPS * (1. + rho * fabs_ * (1. + k))
            if abs(f) <= err:
                its[j] = it + 1
                break
            early = False
            if fprev is not None and abs(f) > abs(fprev) / 10.:
                swtch = not swtch
            fprev = f
            if f < 0:
                lo = mu
            else:
                hi = mu
            dlp = (dj - dorg) - mu
            nmu = 0.5 * (lo + hi)
            if last:
                aa = dpsi * dlp * dlp
                ss = psi - dpsi * dlp
                c0 = 1. + ss + phi
                eta = dlp + aa / c0
                cand = mu + eta
                if eta == eta and lo < cand < hi:
                    nmu = cand
            else:
                drp = (dn - dorg) - mu
                use_fw = variant.endswith("fw") or (variant.endswith("sw") and swtch)
                if use_fw:
                    # fixed weight: the origin pole keeps its true weight, the other pole's weight
                    # and the constant match f and f'
                    fv, dfv = 1. + psi + phi, dpsi + dphi
                    if left:
                        aa = rho * w2[j]
                        bb = drp * drp * (dfv - aa / (dlp * dlp))
                    else:
                        bb = rho * w2[j + 1]
                        aa = dlp * dlp * (dfv - bb / (drp * drp))
                    ss = fv - 1. - aa / dlp - bb / drp
                    rr = 0.
                else:
                    aa = dpsi * dlp * dlp
                    ss = psi - dpsi * dlp
                    bb = dphi * drp * drp
                    rr = phi - dphi * drp
                c0 = 1. + ss + rr
                A1 = -(c0 * (dlp + drp) + aa + bb)
                A0 = c0 * dlp * drp + aa * drp + bb * dlp
                disc = max(A1 * A1 - 4. * c0 * A0, 0.)
                qq = -0.5 * (A1 + (1. if A1 >= 0 else -1.) * np.sqrt(disc))
                e1 = qq / c0 if c0 != 0 else np.nan
                e2 = A0 / qq if qq != 0 else np.nan
                c1, c2 = mu + e1, mu + e2
                if e1 == e1 and lo < c1 < hi:
                    nmu = c1
                elif e2 == e2 and lo < c2 < hi:
                    nmu = c2
            if not (hi - lo > 4. * EPS * max(abs(lo), abs(hi))):
                its[j] = it + 1 + 100
                break
            if "ea" in variant.split("_") and not last:
                dist = min(abs((dj - dorg) - nmu), abs((dn - dorg) - nmu))
                if abs(nmu - mu) <= 2.0 ** -27 * dist and nmu != 0.5 * (lo + hi):
                    mu = nmu
                    its[j] = it + 1
                    break
            mu = nmu
        else:
            its[j] = 64
        mus[j] = mu
        dorgs[j] = dorg
    return its, mus, dorgs


def matrices(n, rng):
    out = []
    Qr = np.linalg.qr(rng.standard_normal((n, n)))[0]
    for name, spec in (("log-uniform 1e-6..1", 10 ** rng.uniform(-6, 0, n)),
                       ("cma-like 1..100", np.linspace(1, 100, n) * (1 + 0.01 * rng.standard_normal(n))),
                       ("near identity", 1 + 1e-3 * rng.standard_normal(n))):
        out.append((name, (Qr * spec) @ Qr.T))
    X = rng.standard_normal((4 * n, n))
    out.append(("wishart", X.T @ X / (4 * n)))
    # a covariance after a few rank-mu updates from the identity
    C = np.eye(n)
    for g in range(30):
        Y = np.linalg.cholesky(C) @ rng.standard_normal((n, 64))
        C = 0.8 * C + 0.2 * (Y @ Y.T) / 64
    out.append(("30 rank-mu updates", C))
    return out


def main():
    variants = sys.argv[1:] or ["device", "dev_ea"]
    rng = np.random.default_rng(5)
    n = 128
    for name, C in matrices(n, rng):
        C = 0.5 * (C + C.T)
        lv = merges_of(C)
        for v in variants:
            line = "%-22s %-8s" % (name, v)
            worst = 0.
            for L, ms in enumerate(lv):
                allit, wavemax = [], []
                for dd, z, rho in ms:
                    dl, w2 = deflate(dd, z, rho)
                    its, mus, dorgs = solve(dl, w2, rho, v)
                    its = its % 100
                    if v != 'device':
                        i0, m0, d0 = solve(dl, w2, rho, 'device')
                        gaps = np.minimum(np.abs(m0), np.abs(np.diff(np.append(dl, dl[-1] + rho * w2.sum())) - np.abs(m0)))
                        worst = max(worst, np.max(np.abs((mus + dorgs) - (m0 + d0)) / np.maximum(gaps, 1e-300)))
                    allit += list(its)
                    lpr = 4
                    per_wave = 64 // lpr
                    wavemax += [its[i:i + per_wave].max() for i in range(0, len(its), per_wave)]
                line += "  L%d: mean %.2f, wave-max mean %.2f max %d" % (L, np.mean(allit), np.mean(wavemax), max(wavemax))
            print(line + ('  worst rel dev %.1e' % worst if v != 'device' else ''))


if __name__ == "__main__":
    main()
