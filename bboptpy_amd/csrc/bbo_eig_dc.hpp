// bbo_eig_dc.hpp -- divide-and-conquer eigensolver for the symmetric TRIDIAGONAL matrix
// produced by the Householder phase (n <= 128), run by the same workgroup as cma_eigen.
//
// Why: the reference's tql2 (cmaes.cpp:383-456) is a strictly serial recurrence -- ~1.4 n^2
// dependent Givens steps of ~1e2 cycles each on a GPU lane, 2-3 ms at n = 128 -- and it sat on
// the critical path of every generation.  Cuppen's divide and conquer does O(n^2) scalar work
// in parallel lanes plus matrix products, and has no long dependent chain:
//   1. scale T by a power of two (exact) to unit max-norm; tear it into leaves of <= 16 rows
//      (rank-one tears, d[b-1] -= |e|, d[b] -= |e|);
//   2. leaves: cyclic Jacobi, one wavefront per leaf;
//   3. merge pairs bottom-up: sort the poles, deflate (tiny z, or two close poles rotated
//      together), solve the secular equation 1 + rho sum z_i^2/(d_i - lam) = 0 for all roots in
//      parallel (4 lanes per root; two-pole rational steps inside a bisection bracket, the
//      root carried as (origin pole, offset) so d_i - lam is exact), recompute z by Loewner's
//      formula (Gu-Eisenstat: orthogonality to rounding error without extended precision),
//      form the eigenvectors of the rank-one update and multiply them into Q on the matrix
//      cores (v_mfma_f64_16x16x4_f64), in place, 16 rows at a time;
//   4. B = Q_house * Q_T on the matrix cores.
// The eigenVALUES and the invariants (C = B D^2 B^T, B^T B = I, C^-1/2) agree with the
// reference's to rounding; the eigenvector SIGNS are D&C's, not tql2's.  scripts/
// dc_prototype.py is the numpy model this file was developed against.
#pragma once

#include "bbo_cma.hpp"

namespace bbo {

typedef double dc_d4 __attribute__((ext_vector_type(4)));

constexpr int DC_LEAF = 16;
constexpr int DC_MAXB = 16;          // max leaves (n <= 128 -> 8)
constexpr double DC_EPS = 0x1.0p-53;

struct DcMat {
    double *a;
    int ld;
    __device__ double& operator()(int i, int j) const { return a[(size_t) i * ld + j]; }
};

__device__ inline void dc_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- leaf: cyclic Jacobi on the s x s tridiagonal block, one wavefront -------------------
// A (s x s, leading dimension 16) lives in LDS scratch, the eigenvectors at Q(a + r, a + c)
__device__ inline void dc_leaf_jacobi(const DcMat &Q, int a, int s, double *A, double *dv,
        const double *ev, int lane)
{
    for (int q = lane; q < s * s; q += 64) {
        const int r = q / s, c = q - r * s;
        double v = 0.;
        if (r == c) v = dv[a + r];
        else if (c == r + 1) v = ev[a + r];
        else if (r == c + 1) v = ev[a + c];
        A[r * 16 + c] = v;
        Q(a + r, a + c) = r == c ? 1. : 0.;
    }
    dc_wave_sync();
    const int mp = (s + 1) & ~1;          // players, padded to even
    const int pairidx = lane >> 3, sub = lane & 7;
    for (int sweep = 0; sweep < 12; sweep++) {
        double off = 0., dia = 0.;
        for (int q = lane; q < s * s; q += 64) {
            const int r = q / s, c = q - r * s;
            const double v = A[r * 16 + c];
            if (r == c) dia += v * v;
            else off += v * v;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_xor(off, o, 64);
            dia += __shfl_xor(dia, o, 64);
        }
        if (off <= 1e-34 * dia || off == 0.) break;
        for (int round = 0; round < mp - 1; round++) {
            int p, q;
            if (pairidx == 0) {
                p = mp - 1;
                q = round;
            } else {
                p = (round + pairidx) % (mp - 1);
                q = (round - pairidx + (mp - 1)) % (mp - 1);
            }
            if (p > q) {
                const int t = p;
                p = q;
                q = t;
            }
            const bool valid = pairidx < (mp >> 1) && q < s;
            double cth = 1., sth = 0.;
            if (valid) {
                const double app = A[p * 16 + p], aqq = A[q * 16 + q], apq = A[p * 16 + q];
                if (apq != 0.) {
                    const double tau = (aqq - app) / (2. * apq);
                    const double t = (tau >= 0. ? 1. : -1.) / (fabs(tau) + sqrt(1. + tau * tau));
                    cth = 1. / sqrt(1. + t * t);
                    sth = t * cth;
                }
            }
            dc_wave_sync();
            if (valid) {   // columns p, q of A and of the eigenvector block
                for (int r = sub; r < s; r += 8) {
                    const double x = A[r * 16 + p], y = A[r * 16 + q];
                    A[r * 16 + p] = cth * x - sth * y;
                    A[r * 16 + q] = sth * x + cth * y;
                    const double vx = Q(a + r, a + p), vy = Q(a + r, a + q);
                    Q(a + r, a + p) = cth * vx - sth * vy;
                    Q(a + r, a + q) = sth * vx + cth * vy;
                }
            }
            dc_wave_sync();
            if (valid) {   // rows p, q of A
                for (int c = sub; c < s; c += 8) {
                    const double x = A[p * 16 + c], y = A[q * 16 + c];
                    A[p * 16 + c] = cth * x - sth * y;
                    A[q * 16 + c] = sth * x + cth * y;
                }
            }
            dc_wave_sync();
        }
    }
    for (int r = lane; r < s; r += 64) dv[a + r] = A[r * 16 + r];
    dc_wave_sync();
}

// LDS work area of one merge
struct DcWork {
    double *dS, *zS;      // [m] poles / z in ascending pole order (modified by deflation)
    double *dl, *w;       // [k] non-deflated poles (ascending) and their z
    double *mu, *what;    // [k] root offsets, Loewner z
    double *lam;          // [m] all eigenvalues of the merged block (unsorted)
    double *ninv;         // [k] 1 / column norm
    double *rotc, *rots;  // [m] deflation rotations
    int *srcS;            // [m] sorted position -> original column
    int *kp, *dp;         // [k] kept / [m-k] deflated sorted positions
    int *org;             // [k] origin pole of root j
    int *outpos;          // [m] output column of eigenvalue q
    int *rotp, *rotj;     // [m] rotation row pairs
    int *cnt;             // [4]: k, ndefl, nrot, flag
    double *red;          // [16] reduction scratch
};

__device__ inline double dc_block_sum(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) s += red[w];
    return s;
}

__device__ inline double dc_block_max(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < (int) (blockDim.x >> 6); w++) s = fmax(s, red[w]);
    return s;
}

// merge the blocks [a, mid) and [mid, b): Q (LDS) holds their eigenvectors on the diagonal
// blocks (zeros elsewhere inside [a,b)^2), dv their eigenvalues; rho = coupling e[mid-1].
// F (global, m x m) is scratch for the eigenvector factor.
__device__ inline void dc_merge(const DcMat &Q, int a, int mid, int b, double rho_in, double *dv,
        double *F, const DcWork &W, long long *stamps)
{
#define MG_STAMP(slot) do { if (stamps && threadIdx.x == 0 && b - a == 128) stamps[slot] = wall_clock64(); } while (0)
    MG_STAMP(24);
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int m = b - a;
    const double sgn = rho_in >= 0. ? 1. : -1.;

    // ---- poles and z, ascending -------------------------------------------------------
    double zi = 0., di = 0.;
    if (tid < m) {
        const int col = a + tid;
        di = dv[col];
        zi = col < mid ? Q(mid - 1, col) : sgn * Q(mid, col);
    }
    const double zn2 = dc_block_sum(tid < m ? zi * zi : 0., W.red);
    const double zn = sqrt(zn2);
    const double rho = fabs(rho_in) * zn2;
    if (tid < m) {
        zi /= zn;
        W.lam[tid] = di;          // unsorted copies for the ranking below
        W.what[tid] = zi;
    }
    __syncthreads();
    if (tid < m) {
        int r = 0;
        for (int j = 0; j < m; j++) {
            const double dj = W.lam[j];
            r += (dj < di) || (dj == di && j < tid);
        }
        W.dS[r] = di;
        W.zS[r] = zi;
        W.srcS[r] = a + tid;
    }
    const double dmax = dc_block_max(tid < m ? fabs(di) : 0., W.red);
    const double zmax = dc_block_max(tid < m ? fabs(zi) : 0., W.red);
    const double tol = 8. * DC_EPS * fmax(dmax, zmax);
    __syncthreads();

    MG_STAMP(25);
    // ---- deflation (sequential scan, LAPACK dlaed2's rules) ---------------------------
    if (tid == 0) {
        int k = 0, nd = 0, nr = 0;
        if (rho * zmax <= tol) {
            for (int j = 0; j < m; j++) W.dp[nd++] = j;
        } else {
            int pj = -1;
            for (int j = 0; j < m; j++) {
                if (rho * fabs(W.zS[j]) <= tol) {
                    W.dp[nd++] = j;
                    continue;
                }
                if (pj < 0) {
                    pj = j;
                    continue;
                }
                double s = W.zS[pj], cth = W.zS[j];
                const double tau = hypot(cth, s);
                const double t = W.dS[j] - W.dS[pj];
                cth /= tau;
                s = -s / tau;
                if (fabs(t * cth * s) <= tol) {
                    W.zS[j] = tau;
                    W.zS[pj] = 0.;
                    W.rotp[nr] = pj;
                    W.rotj[nr] = j;
                    W.rotc[nr] = cth;
                    W.rots[nr] = s;
                    nr++;
                    const double tt = W.dS[pj] * cth * cth + W.dS[j] * s * s;
                    W.dS[j] = W.dS[pj] * s * s + W.dS[j] * cth * cth;
                    W.dS[pj] = tt;
                    W.dp[nd++] = pj;
                    pj = j;
                } else {
                    W.kp[k++] = pj;
                    pj = j;
                }
            }
            if (pj >= 0) W.kp[k++] = pj;
        }
        W.cnt[0] = k;
        W.cnt[1] = nd;
        W.cnt[2] = nr;
    }
    __syncthreads();
    const int k = W.cnt[0], nd = W.cnt[1], nr = W.cnt[2];

    // kept poles ascending (a rotation may perturb the order by rounding)
    if (tid < k) {
        const int pos = W.kp[tid];
        const double dk = W.dS[pos];
        int r = 0;
        for (int j = 0; j < k; j++) {
            const double dj = W.dS[W.kp[j]];
            r += (dj < dk) || (dj == dk && j < tid);
        }
        W.dl[r] = dk;
        W.w[r] = W.zS[pos];
        W.org[r] = pos;            // temporarily: sorted position of kept pole r
    }
    __syncthreads();
    if (tid < k) W.kp[tid] = W.org[tid];
    __syncthreads();

    MG_STAMP(26);
    // ---- secular equation: 4 lanes per root ---------------------------------------------
    if (k == 1) {
        if (tid == 0) {
            W.mu[0] = rho * W.w[0] * W.w[0];
            W.org[0] = 0;
        }
    } else if (k > 1) {
        const int j = tid >> 2, sub = tid & 3;
        const bool act = j < k;
        const bool last = j == k - 1;
        double wsum = 0.;
        if (act && last)
            for (int i = sub; i < k; i += 4) wsum += W.w[i] * W.w[i];
        wsum += __shfl_xor(wsum, 1, 4);
        wsum += __shfl_xor(wsum, 2, 4);
        const double dj = act ? W.dl[j] : 0.;
        const double dn = act ? (last ? dj + rho * wsum : W.dl[j + 1]) : 1.;
        // origin: the sign of f at the midpoint
        double fm = 0.;
        {
            const double midp = 0.5 * (dj + dn);
            if (act)
                for (int i = sub; i < k; i += 4) fm += W.w[i] * W.w[i] / (W.dl[i] - midp);
            fm += __shfl_xor(fm, 1, 4);
            fm += __shfl_xor(fm, 2, 4);
            fm = 1. + rho * fm;
        }
        const bool left = fm > 0. || last;
        const int o = left ? j : j + 1;
        const double dorg = act ? W.dl[min(o, k - 1)] : 0.;
        const double gap = dn - dj;
        double lo = left ? 0. : -0.5 * gap;
        double hi = left ? (last ? gap : 0.5 * gap) : 0.;
        double mu = 0.5 * (lo + hi);
        bool done = !act;
        for (int it = 0; it < 64; it++) {
            double f = 0., fp = 0., psi = 0., dpsi = 0., phi = 0., dphi = 0., fabs_ = 0.;
            if (!done) {
                for (int i = sub; i < k; i += 4) {
                    const double den = (W.dl[i] - dorg) - mu;
                    const double wi = W.w[i];
                    const double t = wi * wi / den;
                    const double tp = t / den;
                    fabs_ += fabs(t);
                    if (i <= j) {
                        psi += t;
                        dpsi += tp;
                    } else {
                        phi += t;
                        dphi += tp;
                    }
                }
            }
#pragma unroll
            for (int s = 1; s < 4; s <<= 1) {
                psi += __shfl_xor(psi, s, 4);
                dpsi += __shfl_xor(dpsi, s, 4);
                phi += __shfl_xor(phi, s, 4);
                dphi += __shfl_xor(dphi, s, 4);
                fabs_ += __shfl_xor(fabs_, s, 4);
            }
            if (!done) {
                psi *= rho; dpsi *= rho; phi *= rho; dphi *= rho;
                f = 1. + psi + phi;
                fp = dpsi + dphi;
                const double err = 8. * DC_EPS * (1. + rho * fabs_ * (1. + k));
                if (fabs(f) <= err) {
                    done = true;
                } else {
                    if (f < 0.) lo = mu;
                    else hi = mu;
                    // two-pole rational model around the bracketing poles (middle way)
                    const double dlp = (dj - dorg) - mu;
                    double nmu = 0.5 * (lo + hi);
                    if (last) {
                        const double aa = dpsi * dlp * dlp, ss = psi - dpsi * dlp;
                        const double c0 = 1. + ss + phi;
                        const double eta = dlp + aa / c0;
                        const double cand = mu + eta;
                        if (eta == eta && cand > lo && cand < hi) nmu = cand;
                    } else {
                        const double drp = (dn - dorg) - mu;
                        const double aa = dpsi * dlp * dlp, ss = psi - dpsi * dlp;
                        const double bb = dphi * drp * drp, rr = phi - dphi * drp;
                        const double c0 = 1. + ss + rr;
                        const double A1 = -(c0 * (dlp + drp) + aa + bb);
                        const double A0 = c0 * dlp * drp + aa * drp + bb * dlp;
                        const double disc = fmax(A1 * A1 - 4. * c0 * A0, 0.);
                        const double qq = -0.5 * (A1 + (A1 >= 0. ? 1. : -1.) * sqrt(disc));
                        const double e1 = qq / c0, e2 = A0 / qq;
                        const double c1 = mu + e1, c2 = mu + e2;
                        if (e1 == e1 && c1 > lo && c1 < hi) nmu = c1;
                        else if (e2 == e2 && c2 > lo && c2 < hi) nmu = c2;
                    }
                    if (!(hi - lo > 4. * DC_EPS * fmax(fabs(lo), fabs(hi)))) done = true;
                    else mu = nmu;
                }
                (void) fp;
            }
            if (__syncthreads_count(done ? 0 : 1) == 0) break;
        }
        if (act && sub == 0) {
            W.mu[j] = mu;
            W.org[j] = min(o, k - 1);
        }
    }
    __syncthreads();

    MG_STAMP(27);
    // ---- Loewner z and column norms -------------------------------------------------------
    if (k > 1) {
        const int i = tid >> 2, sub = tid & 3;
        double prod = 1.;
        if (i < k) {
            const double di2 = W.dl[i];
            for (int jj = sub; jj < k; jj += 4) {
                // lam_jj - d_i = (d_org - d_i) + mu
                const double num = (W.dl[W.org[jj]] - di2) + W.mu[jj];
                // pair it with a denominator d_j' - d_i, j' != i
                if (jj < i) prod *= num / (W.dl[jj] - di2);
                else if (jj == i) prod *= num;
                else prod *= num / (W.dl[jj] - di2);
            }
        }
        prod *= __shfl_xor(prod, 1, 4);
        prod *= __shfl_xor(prod, 2, 4);
        if (i < k && sub == 0) {
            const double v = sqrt(fabs(prod));
            W.what[i] = W.w[i] >= 0. ? v : -v;
        }
        __syncthreads();
        const int j = tid >> 2;
        double ss = 0.;
        if (j < k)
            for (int ii = sub; ii < k; ii += 4) {
                const double del = (W.dl[ii] - W.dl[W.org[j]]) - W.mu[j];
                const double s = W.what[ii] / del;
                ss += s * s;
            }
        ss += __shfl_xor(ss, 1, 4);
        ss += __shfl_xor(ss, 2, 4);
        if (j < k && sub == 0) W.ninv[j] = 1. / sqrt(ss);
    } else if (k == 1) {
        if (tid == 0) {
            W.what[0] = 1.;
            W.ninv[0] = 1.;
        }
    }
    __syncthreads();

    MG_STAMP(28);
    // ---- all eigenvalues, their output order ------------------------------------------------
    if (tid < k) W.lam[tid] = W.dl[W.org[tid]] + W.mu[tid];
    if (tid < nd) W.lam[k + tid] = W.dS[W.dp[tid]];
    __syncthreads();
    if (tid < m) {
        const double v = W.lam[tid];
        int r = 0;
        for (int j = 0; j < m; j++) {
            const double u = W.lam[j];
            r += (u < v) || (u == v && j < tid);
        }
        W.outpos[tid] = r;
    }
    __syncthreads();

    // ---- F = (deflation rotations) x (eigenvector factor), rows in ORIGINAL column order ---
    for (int q = tid; q < m * m; q += T) F[q] = 0.;
    __syncthreads();
    if (k == 1) {
        if (tid == 0) F[(size_t) (W.srcS[W.kp[0]] - a) * m + W.outpos[0]] = 1.;
    } else {
        for (int q = tid; q < k * k; q += T) {
            const int i = q / k, j = q - i * k;
            const double del = (W.dl[i] - W.dl[W.org[j]]) - W.mu[j];
            F[(size_t) (W.srcS[W.kp[i]] - a) * m + W.outpos[j]] = W.what[i] / del * W.ninv[j];
        }
    }
    if (tid < nd) F[(size_t) (W.srcS[W.dp[tid]] - a) * m + W.outpos[k + tid]] = 1.;
    __syncthreads();
    for (int r = nr - 1; r >= 0; r--) {
        // Q G with G = [[c, -s], [s, c]] on sorted columns (p, j): rows p, j of F mix
        const int rp = W.srcS[W.rotp[r]] - a, rj = W.srcS[W.rotj[r]] - a;
        const double cth = W.rotc[r], s = W.rots[r];
        for (int cidx = tid; cidx < m; cidx += T) {
            const double x = F[(size_t) rp * m + cidx], y = F[(size_t) rj * m + cidx];
            F[(size_t) rp * m + cidx] = cth * x - s * y;
            F[(size_t) rj * m + cidx] = s * x + cth * y;
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();

    MG_STAMP(29);
    // ---- Q[a:b, a:b] <- Q[a:b, a:b] F on the matrix cores, 16 rows at a time, in place:
    // every wavefront finishes reading the 16 old rows (all k) before any of them is stored
    const int fr = lane & 15, fk = lane >> 4;
    const int ntile = (m + 15) >> 4;
    const int kpad = (m + 3) & ~3;
    for (int rt = 0; rt < ntile; rt++) {
        const int ct = wave;              // m <= 128: at most 8 column tiles, 8 wavefronts
        dc_d4 acc = { 0., 0., 0., 0. };
        const int col = ct * 16 + fr;
        const int arow = rt * 16 + fr;
        if (ct < ntile) {
            for (int ks = 0; ks < (kpad >> 2); ks++) {
                const int kk = 4 * ks + fk;
                const double av = (arow < m && kk < m) ? Q(a + arow, a + kk) : 0.;
                const double bv = (kk < m && col < m) ? F[(size_t) kk * m + col] : 0.;
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
        }
        __syncthreads();
        if (ct < ntile) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rt * 16 + (lane >> 4) + 4 * r;
                if (row < m && col < m) Q(a + row, a + col) = acc[r];
            }
        }
    }
    __syncthreads();
    if (tid < m) dv[a + W.outpos[tid]] = W.lam[tid];
    __syncthreads();
    MG_STAMP(30);
#undef MG_STAMP
}

// D&C driver.  On entry: dv = diagonal, ev[i] = coupling (i, i+1) (ev[n-1] = 0), Q (LDS) =
// Householder matrix Q_house.  On exit: dv = eigenvalues ascending, Bout (global, ld) =
// Q_house * Q_T, i.e. the eigenvectors of the original matrix in columns.
// G (global): 2 * n * n doubles of scratch.  LDS: scratch >= 10 * 130 + 16 doubles, iscratch >=
// 7 * 130 + 4 ints, leafA >= 16 * 256 doubles (may overlap scratch/iscratch: used before them... no:
// the block reduction that sets the scale uses W.red, so leafA must not overlap W.red).
__device__ inline void eig_dc_phase(const DcMat &Q, int n, double *dv, double *ev, double *G,
        double *Bout, int ldb, double *scratch, int *iscratch, double *leafA,
        long long *stamps)
{
#define DC_STAMP(slot) do { if (stamps && threadIdx.x == 0) stamps[slot] = wall_clock64(); } while (0)
    DC_STAMP(16);
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    double *Qh = G;                       // Q_house, n x n row-major
    double *F = G + (size_t) n * n;       // merge factor
    __shared__ int bounds[DC_MAXB + 1];
    __shared__ int nblk_s;
    __shared__ double scale_s;

    for (int q = tid; q < n * n; q += T) {
        const int r = q / n, c = q - r * n;
        Qh[q] = Q(r, c);
    }
    // scale to unit max-norm by a power of two
    double am = 0.;
    for (int i = tid; i < n; i += T) am = fmax(am, fmax(fabs(dv[i]), fabs(ev[i])));
    DcWork W;
    {
        double *p = scratch;
        const int M = 130;
        W.dS = p; p += M; W.zS = p; p += M; W.dl = p; p += M; W.w = p; p += M;
        W.mu = p; p += M; W.what = p; p += M; W.lam = p; p += M; W.ninv = p; p += M;
        W.rotc = p; p += M; W.rots = p; p += M; W.red = p; p += 16;
        int *ip = iscratch;
        W.srcS = ip; ip += M; W.kp = ip; ip += M; W.dp = ip; ip += M; W.org = ip; ip += M;
        W.outpos = ip; ip += M; W.rotp = ip; ip += M; W.rotj = ip; ip += M; W.cnt = ip;
    }
    am = dc_block_max(am, W.red);
    __syncthreads();
    if (tid == 0) {
        int ex = 0;
        if (am > 0.) frexp(am, &ex);
        scale_s = am > 0. ? ldexp(1., 1 - ex) : 1.;
        // bounds: halve until every block has <= DC_LEAF rows
        int nb = 1;
        bounds[0] = 0;
        bounds[1] = n;
        while (true) {
            int widest = 0;
            for (int i = 0; i < nb; i++) widest = max(widest, bounds[i + 1] - bounds[i]);
            if (widest <= DC_LEAF) break;
            int tmp[DC_MAXB + 1];
            int c2 = 0;
            tmp[c2++] = bounds[0];
            for (int i = 0; i < nb; i++) {
                if (bounds[i + 1] - bounds[i] > DC_LEAF) tmp[c2++] = (bounds[i] + bounds[i + 1]) / 2;
                tmp[c2++] = bounds[i + 1];
            }
            nb = c2 - 1;
            for (int i = 0; i <= nb; i++) bounds[i] = tmp[i];
        }
        nblk_s = nb;
    }
    __syncthreads();
    const double scale = scale_s;
    const int nblk = nblk_s;
    for (int i = tid; i < n; i += T) {
        dv[i] *= scale;
        ev[i] *= scale;
    }
    // Q becomes the eigenvector matrix of T: clear it
    for (int q = tid; q < n * Q.ld; q += T) Q.a[q] = 0.;
    __syncthreads();
    // rank-one tears at the block boundaries
    if (tid >= 1 && tid < nblk) {
        const int bd = bounds[tid];
        const double r = fabs(ev[bd - 1]);
        dv[bd - 1] -= r;
        dv[bd] -= r;
    }
    __syncthreads();

    DC_STAMP(17);
    // ---- leaves: one wavefront each, its 16 x 16 work matrix in LDS scratch (the merge work
    // area is not in use yet).  W.red sits behind the first 8 leaf slots' worth? No: the leaf
    // slots start at `leafA`, past the reduction scratch used above.
    for (int blk = wave; blk < nblk; blk += (T >> 6)) {
        const int a = bounds[blk], s = bounds[blk + 1] - a;
        dc_leaf_jacobi(Q, a, s, leafA + (size_t) blk * 256, dv, ev, lane);
    }
    __syncthreads();

    DC_STAMP(18);
    // ---- merges, bottom-up ---------------------------------------------------------------------
    // level structure = the halving above run backwards: adjacent blocks pair up while the
    // block list is walked left to right
    int cur[DC_MAXB + 1];
    int nc = nblk;
    for (int i = 0; i <= nblk; i++) cur[i] = bounds[i];
    while (nc > 1) {
        int nxt[DC_MAXB + 1];
        int nn = 0;
        nxt[nn++] = cur[0];
        for (int i = 0; i + 1 < nc; i += 2) {
            const int a = cur[i], mid = cur[i + 1], b = cur[i + 2];
            dc_merge(Q, a, mid, b, ev[mid - 1], dv, F, W, stamps);
            nxt[nn++] = b;
        }
        if (nc & 1) nxt[nn++] = cur[nc];
        nc = nn - 1;
        for (int i = 0; i <= nc; i++) cur[i] = nxt[i];
        DC_STAMP(19 + (nc == 1 ? 2 : nc == 2 ? 1 : 0));
    }
    __syncthreads();
    // a single leaf (n <= 16) never went through a merge: sort its eigenpairs here
    const bool single = nblk == 1;
    if (single) {
        if (tid < n) {
            const double v = dv[tid];
            int r = 0;
            for (int j = 0; j < n; j++) {
                const double u = dv[j];
                r += (u < v) || (u == v && j < tid);
            }
            W.outpos[tid] = r;
            W.lam[tid] = v;
        }
        __syncthreads();
        if (tid < n) dv[W.outpos[tid]] = W.lam[tid];
        __syncthreads();
    }
    const double inv = 1. / scale;
    for (int i = tid; i < n; i += T) dv[i] *= inv;
    __syncthreads();

    DC_STAMP(22);
    // ---- B = Q_house * Q_T on the matrix cores ------------------------------------------------
    const int ntile = (n + 15) >> 4;
    const int kpad = (n + 3) & ~3;
    const int fr = lane & 15, fk = lane >> 4;
    for (int t = wave; t < ntile * ntile; t += (T >> 6)) {
        const int rt = t / ntile, ct = t - rt * ntile;
        dc_d4 acc = { 0., 0., 0., 0. };
        const int arow = rt * 16 + fr, col = ct * 16 + fr;
        for (int ks = 0; ks < (kpad >> 2); ks++) {
            const int kk = 4 * ks + fk;
            const double av = (arow < n && kk < n) ? Qh[(size_t) arow * n + kk] : 0.;
            const double bv = (kk < n && col < n) ? Q(kk, col) : 0.;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rt * 16 + (lane >> 4) + 4 * r;
            if (row < n && col < n)
                Bout[(size_t) row * ldb + (single ? W.outpos[col] : col)] = acc[r];
        }
    }
    __syncthreads();
    DC_STAMP(23);
#undef DC_STAMP
}

} // namespace bbo
