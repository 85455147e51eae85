"""numpy prototype of the divide-and-conquer tridiagonal eigensolver that bbo_eig_dc.hpp
implements on the device (development aid; not imported by the product or the tests)."""
import numpy as np

EPS = 2.0 ** -53


def secular_roots(d, w2, rho):
    """roots of 1 + rho * sum(w2_i / (d_i - lam)) = 0, d ascending, w2 > 0, rho > 0.
    Returns (origin index o_j, mu_j) with lam_j = d[o_j] + mu_j, all j in parallel."""
    k = d.size
    j = np.arange(k)
    last = j == k - 1
    dn = np.where(last, d[-1] + rho * w2.sum(), d[np.minimum(j + 1, k - 1)])
    # choose the origin: evaluate f at the midpoint
    mid = 0.5 * (d + dn)
    fm = 1. + rho * (w2[None, :] / (d[None, :] - mid[:, None])).sum(axis=1)
    left = (fm > 0) | last
    o = np.where(left, j, np.minimum(j + 1, k - 1))
    delta = d[None, :] - d[o][:, None]            # delta[j, i] = d_i - origin_j
    gap = dn - d
    lo = np.where(left, 0., -0.5 * gap)
    hi = np.where(left, np.where(last, gap, 0.5 * gap), 0.)
    mu = 0.5 * (lo + hi)
    done = np.zeros(k, bool)
    for it in range(80):
        den = delta - mu[:, None]
        t = w2[None, :] / den
        f = 1. + rho * t.sum(axis=1)
        fp = rho * (t / den).sum(axis=1)
        err = 8 * EPS * (1. + rho * np.abs(t).sum(axis=1) * (1 + k))   # generous
        conv = np.abs(f) <= err
        done |= conv
        # bracket update
        lo = np.where(f < 0, mu, lo)
        hi = np.where(f >= 0, mu, hi)
        # rational (two-pole) step: split at the origin pole
        # psi: poles <= j  (left), phi: poles > j (right)
        idx = np.arange(k)[None, :]
        lmask = idx <= j[:, None]
        psi = (t * lmask).sum(axis=1) * rho
        dpsi = (t / den * lmask).sum(axis=1) * rho
        phi = (t * ~lmask).sum(axis=1) * rho
        dphi = (t / den * ~lmask).sum(axis=1) * rho
        # poles: dl = delta[j, j] - mu (left pole), dr = delta[j, j+1] - mu (right pole)
        dl = delta[j, j] - mu
        dr = np.where(last, np.inf, delta[j, np.minimum(j + 1, k - 1)] - mu)
        # match psi ~ s + a/(dl - eta), phi ~ r + b/(dr - eta) at eta = 0 (middle way)
        a = dpsi * dl * dl
        s = psi - dpsi * dl
        with np.errstate(invalid="ignore", over="ignore"):
            b = np.where(last, 0., dphi * dr * dr)
            r = np.where(last, phi, phi - dphi * dr)
        c0 = 1. + s + r
        # solve c0 + a/(dl - eta) + b/(dr - eta) = 0
        # -> c0 (dl-eta)(dr-eta) + a (dr-eta) + b (dl-eta) = 0
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            A2 = c0
            A1 = -(c0 * (dl + dr) + a + b)
            A0 = c0 * dl * dr + a * dr + b * dl
            disc = np.maximum(A1 * A1 - 4 * A2 * A0, 0.)
            sq = np.sqrt(disc)
            q = -0.5 * (A1 + np.sign(A1 + (A1 == 0)) * sq)
            e1 = q / A2
            e2 = A0 / q
            # last root: single pole model c0 + a/(dl - eta) = 0 -> eta = dl + a/c0
            el = dl + a / c0
        newmu = np.full(k, np.nan)
        for cand in (e1, e2):
            ok = np.isfinite(cand) & (mu + cand > lo) & (mu + cand < hi)
            newmu = np.where(np.isnan(newmu) & ok, mu + cand, newmu)
        okl = last & np.isfinite(el) & (mu + el > lo) & (mu + el < hi)
        newmu = np.where(okl, mu + el, newmu)
        newmu = np.where(np.isnan(newmu), 0.5 * (lo + hi), newmu)
        mu = np.where(done, mu, newmu)
        if done.all():
            break
        if np.all((hi - lo) <= 4 * EPS * np.maximum(np.abs(lo), np.abs(hi))):
            break
    return o, mu, it + 1


def merge(d, z, rho, Q):
    """eigen-decomposition of diag(d) + rho z z^T given Q (columns ~ d); returns lam, Qnew"""
    m = d.size
    nz = np.linalg.norm(z)
    z = z / nz
    rho = rho * nz * nz
    order = np.argsort(d, kind="stable")
    d, z, Q = d[order].copy(), z[order].copy(), Q[:, order].copy()
    tol = 8 * EPS * max(np.abs(d).max(), np.abs(z).max())
    if rho * np.abs(z).max() <= tol:
        return d, Q
    keep = []
    defl = []
    pj = -1
    for jj in range(m):
        if rho * abs(z[jj]) <= tol:
            defl.append(jj)
            continue
        if pj < 0:
            pj = jj
            continue
        s, c = z[pj], z[jj]
        tau = np.hypot(c, s)
        t = d[jj] - d[pj]
        c /= tau
        s = -s / tau
        if abs(t * c * s) <= tol:
            z[jj] = tau
            z[pj] = 0.
            qp, qj = Q[:, pj].copy(), Q[:, jj].copy()
            Q[:, pj] = c * qp + s * qj
            Q[:, jj] = -s * qp + c * qj
            tt = d[pj] * c * c + d[jj] * s * s
            d[jj] = d[pj] * s * s + d[jj] * c * c
            d[pj] = tt
            defl.append(pj)
            pj = jj
        else:
            keep.append(pj)
            pj = jj
    if pj >= 0:
        keep.append(pj)
    keep = np.array(keep, int)
    defl = np.array(defl, int)
    k = keep.size
    dl, w = d[keep], z[keep]
    # dl must be ascending (deflation by rotation can perturb order slightly)
    oo = np.argsort(dl, kind="stable")
    dl, w, keep = dl[oo], w[oo], keep[oo]
    if k == 1:
        lam = np.array([dl[0] + rho * w[0] * w[0]])
        S = np.ones((1, 1))
    else:
        o, mu, its = secular_roots(dl, w * w, rho)
        delta = (dl[:, None] - dl[o][None, :]) - mu[None, :]     # delta[i, j] = d_i - lam_j
        lam = dl[o] + mu
        # Loewner: what = sqrt( prod_j (lam_j - d_i) / prod_{j != i} (d_j - d_i) ) / sqrt(rho)
        what = np.empty(k)
        for i in range(k):
            num = -delta[i, :]                   # lam_j - d_i
            den = np.delete(dl - dl[i], i)
            # interleave to avoid overflow: pair num_j with den_j
            ratio = np.empty(k)
            ratio[:i] = num[:i] / den[:i]
            ratio[i] = num[i]
            ratio[i + 1:] = num[i + 1:] / den[i:]
            what[i] = np.sqrt(np.abs(np.prod(ratio))) * np.sign(w[i])
        S = what[:, None] / delta
        S /= np.linalg.norm(S, axis=0)[None, :]
    Qk = Q[:, keep] @ S
    lam_all = np.concatenate([lam, d[defl]])
    Q_all = np.concatenate([Qk, Q[:, defl]], axis=1)
    oo = np.argsort(lam_all, kind="stable")
    return lam_all[oo], Q_all[:, oo]


def dc_eigh_tridiag(d, e, leaf=16):
    n = d.size
    d = d.astype(float).copy()
    e = np.asarray(e, float).copy()
    # scale to unit max-norm by a power of two (exact), like dstedc's dlascl
    amax = max(np.abs(d).max(), np.abs(e).max() if e.size else 0.)
    scale = 1.
    if amax > 0:
        scale = 2.0 ** (-np.floor(np.log2(amax)))
        d *= scale
        e *= scale
    bounds = [0, n]
    while max(b - a for a, b in zip(bounds[:-1], bounds[1:])) > leaf:
        nb = [bounds[0]]
        for a, b in zip(bounds[:-1], bounds[1:]):
            if b - a > leaf:
                nb.append((a + b) // 2)
            nb.append(b)
        bounds = nb
    for b in bounds[1:-1]:
        r = abs(e[b - 1])
        d[b - 1] -= r
        d[b] -= r
    blocks = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        T = np.diag(d[a:b]) + np.diag(e[a:b - 1], 1) + np.diag(e[a:b - 1], -1)
        lam, Q = np.linalg.eigh(T)
        blocks.append((a, b, lam, Q))
    while len(blocks) > 1:
        nxt = []
        for (a, mid, l1, Q1), (_, b, l2, Q2) in zip(blocks[0::2], blocks[1::2]):
            rho = e[mid - 1]
            z = np.concatenate([Q1[-1, :], np.sign(rho) * Q2[0, :]])
            m1, m2 = mid - a, b - mid
            Q = np.zeros((m1 + m2, m1 + m2))
            Q[:m1, :m1] = Q1
            Q[m1:, m1:] = Q2
            lam, Qn = merge(np.concatenate([l1, l2]), z, abs(rho), Q)
            nxt.append((a, b, lam, Qn))
        if len(blocks) % 2:
            nxt.append(blocks[-1])
        blocks = nxt
    return blocks[0][2] / scale, blocks[0][3]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    worst = 0
    for trial in range(60):
        n = int(rng.choice([16, 37, 64, 100, 128]))
        kind = trial % 6
        if kind == 0:
            A = np.eye(n)
        elif kind == 1:
            A = np.eye(n) + 1e-3 * np.cov(rng.normal(size=(n, 4 * n)))
        elif kind == 2:
            X = rng.normal(size=(n, 3 * n)) * np.logspace(0, -6, n)[:, None]
            A = X @ X.T
        elif kind == 3:
            A = np.diag(np.repeat(rng.normal(size=n // 4 + 1), 4)[:n]) + 1e-13 * np.cov(rng.normal(size=(n, n)))
        elif kind == 4:
            A = np.cov(rng.normal(size=(n, 2 * n))) * 1e-24
        else:
            A = np.diag(np.arange(1., n + 1)) + 1e-9 * np.ones((n, n))
        from scipy.linalg import hessenberg
        H, Qh = hessenberg(A, calc_q=True)
        d, e = np.diag(H).copy(), np.diag(H, 1).copy()
        lam, Q = dc_eigh_tridiag(d, e)
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        sc = max(np.abs(T).max(), 1e-300)
        res = np.abs(T @ Q - Q * lam[None, :]).max() / sc
        orth = np.abs(Q.T @ Q - np.eye(n)).max()
        ref = np.linalg.eigvalsh(T)
        ev = np.abs(lam - ref).max() / sc
        worst = max(worst, res, orth, ev)
        print("trial %2d kind %d n %3d  resid %.2e  orth %.2e  eig %.2e" % (trial, kind, n, res, orth, ev))
    print("worst", worst)
