// bbo_eig_dc.hpp -- divide-and-conquer eigensolver for the symmetric TRIDIAGONAL matrix
// produced by the Householder phase (n <= 128), run by the same workgroup as cma_eigen.
//
// Why: the reference's tql2 (cmaes.cpp:383-456) is a strictly serial recurrence -- ~1.4 n^2
// dependent Givens steps of ~1e2 cycles each on a GPU lane, 2-3 ms at n = 128 -- and it sat on
// the critical path of every generation.  Cuppen's divide and conquer does O(n^2) scalar work
// in parallel lanes plus matrix products, and has no long dependent chain:
//   1. scale T by a power of two (exact) to unit max-norm; tear it into leaves of <= 16 rows
//      (rank-one tears, d[b-1] -= |e|, d[b] -= |e|);
//   2. leaves: the reference's own QL recurrence (bbo_eig_ql.hpp), one wavefront per leaf,
//      all leaves at once (a 16 x 16 leaf is ~200 Givens steps, not 18 000);
//   3. merge pairs bottom-up, ALL merges of a level at once (each by its own team of
//      wavefronts): sort the poles, deflate (tiny z, or two close poles rotated together),
//      solve the secular equation 1 + rho sum z_i^2/(d_i - lam) = 0 for all roots in parallel
//      (up to 4 lanes per root; two-pole rational steps inside a bisection bracket, the root
//      carried as (origin pole, offset) so d_i - lam is exact), recompute z by Loewner's
//      formula (Gu-Eisenstat: orthogonality to rounding error without extended precision),
//      form the eigenvectors of the rank-one update and multiply them into Q on the matrix
//      cores (v_mfma_f64_16x16x4_f64), in place, 16 rows at a time;
//   4. B = Q_house * Q_T on the matrix cores.
// The eigenVALUES and the invariants (C = B D^2 B^T, B^T B = I, C^-1/2) agree with the
// reference's to rounding; the eigenvector SIGNS are D&C's, not tql2's.  scripts/
// dc_prototype.py is the numpy model this file was developed against.
#pragma once

#include "bbo_cma.hpp"
#include "bbo_eig_ql.hpp"

namespace bbo {

typedef double dc_d4 __attribute__((ext_vector_type(4)));

constexpr int DC_LEAF = 16;
constexpr int DC_MAXB = 16;          // max leaves, n <= 256
constexpr int DC_MAXB_BIG = 32;      // ... n <= 512 (the BIG instantiations)
constexpr double DC_EPS = 0x1.0p-53;
constexpr int DC_KSTEPS = 32;        // k-steps of a 128-deep MFMA contraction

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

// the synchronisation of a team inside a merge: a team of ONE wavefront (the lowest merge level:
// 16-pole merges of 8 x 8 leaves, a wavefront each) only has to order its own LDS traffic -- its
// ~25 phase boundaries cost a wavefront fence each instead of a workgroup barrier that waits for
// the slowest of eight unrelated merges (round 4).  `wv` is the same for every thread of the
// workgroup (all teams of a level have the same number of wavefronts).
__device__ inline void dc_sync(bool wv)
{
    if (wv) dc_wave_sync();
    else __syncthreads();
}

// 1/x to (nearly) full precision: hardware estimate + two Newton steps
__device__ inline double dc_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
}

// sqrt(x) for x >= 0 from the reciprocal-root estimate (one third-order correction + one step on
// the root: full fp64 to a rounding error); 0 for x = 0
__device__ inline double dc_sqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double err = fma(-x * y, y, 1.);
    y = fma(y * err, fma(err, 0.375, 0.5), y);
    double r = x * y;
    r = fma(fma(-r, r, x), 0.5 * y, r);
    return x > 0. && x < __builtin_huge_val() ? r : x;
}

// Key of a ranking by counting.  The maps built from such rankings (sorted position -> source
// column, root -> output column) are used as INDICES: a ranking must be a permutation whatever
// the data.  With a NaN every comparison is false, all NaN entries would get the same rank and
// some positions none (the index read back from them is then arbitrary -- a covariance with a
// NaN entry made the eigensolver fault).  NaN ranks as +infinity; equal keys rank by position.
__device__ inline double dc_key(double v) { return v == v ? v : __builtin_huge_val(); }

// Rank by counting: the number of entries q < cnt of v (LDS) whose key is below `key`, or equal
// to it with q < self.  Eight entries are requested before the first is looked at, and the trip
// count is the wavefront's (cnt is the team's): written as a plain loop over v[q] this compiled to
// one LDS round trip per two entries -- 64 serial waits for a 128-entry ranking, 3 us, three
// rankings per merge.
__device__ inline int dc_rank_of(const double *v, int cnt, double key, int self)
{
    const int cu = __builtin_amdgcn_readfirstlane(cnt);
    int r = 0;
    for (int q0 = 0; q0 < cu; q0 += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = v[min(q0 + u, cu - 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            // (bitwise on purpose: && / || compile to a branch per entry here)
            const double d = dc_key(x[u]);
            const int qq = q0 + u;
            const int lt = d < key ? 1 : 0, eq = d == key ? 1 : 0, before = qq < self ? 1 : 0;
            r += (qq < cu ? 1 : 0) & (lt | (eq & before));
        }
    }
    return r;
}

// The same rank for an entry of one of two lists that are EACH in ascending key order: its place in
// its own list plus the number of entries of the other list that go before it -- entries with a
// smaller key, and (for an entry of the second list, whose positions all come later) those with an
// equal one.  A binary search: log2 dependent LDS reads instead of 128 comparisons.
// ONE loop with a trip count that is the same for every lane (`trips` >= log2(cnt) + 1 for the
// longest list of the team), the list, its length and the tie rule per-lane VALUES: written as
// `first ? a + search(listB, strict) : b + search(listA, or_equal)` -- two data-dependent while
// loops under a select -- the 128- / 256-thread instantiations of the eigensolver came out of the
// compiler with the tie rule of the two lists exchanged and the last entry of the second list
// ranked 0 (round 4: B of the identity matrix at n = 18 had column 0 twice and column 8 missing;
// every n in 17..64 was wrong, and whether a build showed it depended on unrelated code nearby).
__device__ inline int dc_count_before(const double *v, int cnt, double key, bool or_equal, int trips)
{
    int lo = 0, hi = cnt;          // v[0 .. lo) go before, v[hi .. cnt) do not
    for (int t = 0; t < trips; t++) {
        const bool open = lo < hi;
        const int mid = (lo + hi) >> 1;
        const double d = dc_key(v[open ? mid : 0]);       // (v[0] is a legal address for cnt = 0)
        const bool before = or_equal ? d <= key : d < key;
        lo = (open && before) ? mid + 1 : lo;
        hi = (open && !before) ? mid : hi;
    }
    return lo;
}

// rank of entry `self` (key) of the concatenation [list A: n_a entries | list B: n_b entries] at v
#ifdef DC_OLD_TWO_BRANCH
// (the form of rounds 2-3, kept for scripts/hiptests/rank_tie.hip and variant builds: two
// data-dependent while loops under a divergent select -- see DESIGN.md section 4)
__device__ inline int dc_count_before_while(const double *v, int cnt, double key, bool or_equal)
{
    int lo = 0, hi = cnt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const double d = dc_key(v[mid]);
        const bool before = or_equal ? d <= key : d < key;
        lo = before ? mid + 1 : lo;
        hi = before ? hi : mid;
    }
    return lo;
}
__device__ inline int dc_rank_sorted2(const double *v, int n_a, int n_b, double key, int self, int trips)
{
    return self < n_a ? self + dc_count_before_while(v + n_a, n_b, key, false)
                      : (self - n_a) + dc_count_before_while(v, n_a, key, true);
}
#else
__device__ inline int dc_rank_sorted2(const double *v, int n_a, int n_b, double key, int self, int trips)
{
    const bool first = self < n_a;
    const double *other = first ? v + n_a : v;
    const int ocnt = first ? n_b : n_a;
    const int own = first ? self : self - n_a;
    return own + dc_count_before(other, ocnt, key, !first, trips);
}
#endif

// the team of wavefronts that works on one merge
struct DcTeam {
    int active;       // has a merge at this level
    int ttid;         // thread index inside the team
    int tthreads;     // threads of the team
    int wave0, nwaves, twave;
    int id;
    int sorted_in;    // both blocks come out of merges: their eigenvalues are in ascending order
    int wave_scope;   // the team is one wavefront and synchronises as one (dc_sync)
};

// LDS work area shared by all merges of a level: every array is indexed by the block's
// global column range [a, b), so concurrent merges use disjoint slices
struct DcWork {
    double *dS, *zS;      // poles / z in ascending pole order (modified by deflation)
    double *dl, *w2;      // non-deflated poles (ascending), squared z
    double *ws;           // sign carrier: z of the non-deflated poles
    double *mu, *what;    // root offsets, Loewner z
    double *lam;          // all eigenvalues of the merged block (unsorted)
    double *ninv;         // 1 / column norm
    double *rotc, *rots;  // deflation rotations
    int *srcS;            // sorted position -> original column
    int *kp, *dp;         // kept / deflated sorted positions
    int *org;             // origin pole of root j
    int *outpos;          // output column of eigenvalue q
    int *rotp, *rotj;     // rotation row pairs
    int *rowmap;          // original column -> kept index i, or -(1 + output column)
    int *colroot;         // output column -> root j, or -1
    int *cnt;             // [team][4]: k, ndefl, nrot, 'needs the sequential scan'
    double *red;          // [16] reduction scratch (one slot per wavefront)
    unsigned long long *balk, *bald;   // [8] per-wavefront ballots: kept / deflated poles
    int *maxnr;           // [1]
};

// (wavefront stage on the cross-lane data path, eig_wave_sum / dc_wave_max: six ds_bpermute round
// trips through the LDS crossbar before)
__device__ inline double dc_wave_max(double v)
{
    v = fmax(v, eig_dpp<0x128>(v));   // row_ror:8
    v = fmax(v, eig_dpp<0x124>(v));   // row_ror:4
    v = fmax(v, eig_dpp<0x122>(v));   // row_ror:2
    v = fmax(v, eig_dpp<0x121>(v));   // row_ror:1
    return fmax(fmax(eig_readlane(v, 0), eig_readlane(v, 16)), fmax(eig_readlane(v, 32), eig_readlane(v, 48)));
}

__device__ inline double dc_team_sum(double v, double *red, const DcTeam &tm)
{
    v = eig_wave_sum(v);
    if (tm.wave_scope) return 0. + v;        // (what the loop below adds up for one wavefront)
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.;
    for (int w = 0; w < tm.nwaves; w++) s += red[tm.wave0 + w];
    return s;
}

// the same sum (same order, same bits) + "some thread of the team raised its flag", on the same two
// barriers: the flags travel in the upper half of red (a workgroup has at most 8 wavefronts)
__device__ inline double dc_team_sum_flag(double v, bool flag, bool &any, double *red, const DcTeam &tm)
{
    v = eig_wave_sum(v);
    const bool wf = __ballot(flag) != 0ull;
    if (tm.wave_scope) {
        any = wf;
        return 0. + v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = v;
        red[8 + (threadIdx.x >> 6)] = wf ? 1. : 0.;
    }
    __syncthreads();
    double s = 0., f = 0.;
    for (int w = 0; w < tm.nwaves; w++) {
        s += red[tm.wave0 + w];
        f += red[8 + tm.wave0 + w];
    }
    any = f != 0.;
    return s;
}

__device__ inline double dc_team_max(double v, double *red, const DcTeam &tm)
{
    v = dc_wave_max(v);
    if (tm.wave_scope) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[tm.wave0];
    for (int w = 1; w < tm.nwaves; w++) s = fmax(s, red[tm.wave0 + w]);
    return s;
}

template<int LPR>
__device__ inline double dc_quad_sum(double v)
{
    if (LPR >= 2) v += eig_quad_xor1(v);
    if (LPR >= 4) v += eig_quad_xor2(v);
    if (LPR >= 8) v += eig_dpp<0x141>(v);     // row_half_mirror: the other quad of the half row
    if (LPR >= 16) v += eig_dpp<0x140>(v);    // row_mirror: the other half of the row
    return v;
}

// the work area of the merges at p (LDS inside a workgroup; a global image of it between the
// kernels of a split top merge): arrays of M entries, see DcWork
__device__ inline DcWork dc_work_layout(double *p, int M, int *maxnr, size_t *doubles_used = nullptr)
{
    DcWork W;
    double *p0 = p;
    W.dS = p; p += M; W.zS = p; p += M; W.dl = p; p += M; W.w2 = p; p += M; W.ws = p; p += M;
    W.mu = p; p += M; W.what = p; p += M; W.lam = p; p += M; W.ninv = p; p += M;
    W.rotc = p; p += M; W.rots = p; p += M; W.red = p; p += 16;
    W.balk = reinterpret_cast<unsigned long long*>(p); p += 8;
    W.bald = reinterpret_cast<unsigned long long*>(p); p += 8;
    int *ip = reinterpret_cast<int*>(p);
    W.srcS = ip; ip += M; W.kp = ip; ip += M; W.dp = ip; ip += M; W.org = ip; ip += M;
    W.outpos = ip; ip += M; W.rotp = ip; ip += M; W.rotj = ip; ip += M;
    W.rowmap = ip; ip += M; W.colroot = ip; ip += M; W.cnt = ip; ip += 4 * 8;
    W.maxnr = maxnr;
    if (doubles_used) *doubles_used = (size_t) (p - p0) + ((size_t) (ip - reinterpret_cast<int*>(p)) + 1) / 2;
    return W;
}

// ---- the secular equation 1 + rho sum_i w_i / (d_i - lam) = 0: root j by LPR lanes (thread rt of
// the team: j = rt / LPR), every lane the poles i = sub + LPR t.  Reads W.dl, W.w2 (k kept poles,
// ascending); writes W.mu[j] (offset from the origin pole), W.org[j], W.lam[j] (the origin pole) and
// raises W.cnt[3] when a root leaves its interval (non-finite input).  A function of its own since
// round 4: the top merge of a matrix wider than 128 runs it on several workgroups
// (cma_eig_secular), the merges inside a workgroup inline it as before.
template<int LPR>
__device__ __forceinline__ void dc_secular(const DcWork &W, int k, double rho, int rt, bool on,
        long long *stamps, bool stamp_here, int width)
{
#define SEC_STAMP(slot) do { if (stamps && stamp_here && threadIdx.x == 0) stamps[slot] = wall_clock64(); } while (0)
    const int j = rt / LPR, sub = rt % LPR;
    const bool act = on && k > 1 && j < k;
    const bool last = j == k - 1;
    double wsum = 0.;
    if (act && last)
        for (int i = sub; i < k; i += LPR) wsum += W.w2[i];
    wsum = dc_quad_sum<LPR>(wsum);
    const double dj = act ? W.dl[j] : 0.;
    const double dn = act ? (last ? dj + rho * wsum : W.dl[j + 1]) : 1.;
    // origin: the sign of f at the midpoint
    double fm = 0.;
    {
        const double midp = 0.5 * (dj + dn);
        if (act) {
            // (four entries requested together, from clamped addresses, and added in the list's order:
            // entry by entry this loop is one dependent round trip per entry when the work area is
            // the global image of cma_eig_secular -- 8 to 16 of them in front of the solve)
            const int npl0 = (k + LPR - 1) / LPR;
            for (int t0 = 0; t0 < npl0; t0 += 4) {
                double ww[4], dd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ic = min(sub + (t0 + u) * LPR, k - 1);
                    ww[u] = W.w2[ic];
                    dd[u] = W.dl[ic];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (sub + (t0 + u) * LPR < k) fm += ww[u] * dc_rcp(dd[u] - midp);
            }
        }
        fm = 1. + rho * dc_quad_sum<LPR>(fm);
    }
    const bool left = fm > 0. || last;
    const int o = act ? min(left ? j : j + 1, k - 1) : 0;
    const double dorg = act ? W.dl[o] : 0.;
    const double gap = dn - dj;
    double lo = left ? 0. : -0.5 * gap;
    double hi = left ? (last ? gap : 0.5 * gap) : 0.;
    double mu = 0.5 * (lo + hi);
    if (act) {
        // first guess: the origin pole with its TRUE weight, all other poles frozen at their
        // midpoint value: 1 + rho (w_o / (d_o - x) + rest) = 0.  Exact in the limit of a tiny
        // weight (root glued to its pole -- where the fitted two-pole model below starts
        // blind and took 10-17 steps), inside the bracket by construction otherwise.
        const double wo = W.w2[o];
        const double rest1 = fm + (left ? 2. : -2.) * rho * wo / gap;    // 1 + rho rest
        const double g0 = rho * wo / rest1;
        if (g0 == g0 && g0 > lo && g0 < hi) mu = g0;
    }
    SEC_STAMP(32);
    bool done = !act;
#ifdef BBO_EIG_SECULAR_HIST
    int hist_it = 0;
#endif
    // The poles of a lane, i = sub + LPR t, in REGISTERS for the whole solve when there are at
    // most 32 of them (LPR = 4: every merge of n <= 128): delta_t = d_i - d_origin and w_t are
    // loaded once, an evaluation is then a subtraction, a reciprocal and the sums -- no LDS
    // read, no address, no end-of-list test per pole and iteration (the loop is bound by
    // vector issue: 31 slots per pole from LDS, 20 from registers).  A pole beyond the list
    // carries w = 0 and a delta no root comes near.
    constexpr int NREG = 32;
    const int npl = (k + LPR - 1) / LPR;
    const int npl_s = __builtin_amdgcn_readfirstlane(npl);     // (k is the team's: uniform in a wavefront)
    const bool inreg = LPR >= 4 && npl_s <= NREG;
    double dreg[NREG], wreg[NREG];
    if (inreg) {
#pragma unroll
        for (int t = 0; t < NREG; t++) {
            const int i = sub + t * LPR;
            const bool in = act && i < k;
            const int ic = in ? i : 0;
            const double dd = W.dl[ic], wl = W.w2[ic];
            dreg[t] = in ? dd - dorg : 0x1p+1000;
            wreg[t] = in ? wl : 0.;
        }
    }
    for (int it = 0; it < 64; it++) {
        // psi / phi: the terms of the poles up to j / beyond j.  The iterate stays strictly
        // between d_j and d_j+1, so the first are the NEGATIVE terms and the second the positive
        // ones: the split is a v_min / v_max with zero (it was two 64-bit selects on the pole's
        // position), and sum |term| = phi - psi needs no accumulator of its own (round 4).
        double psi = 0., dpsi = 0., phi = 0., dphi = 0.;
        if (!done && inreg) {
            // same terms in the same order as the LDS form below: same bits
            // (four poles per scalar branch: four independent reciprocal chains in flight; the
            // padding poles of the last group add exact zeros)
#pragma unroll
            for (int t0 = 0; t0 < NREG; t0 += 4) {
                if (t0 < npl_s) {
                    double r[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) r[u] = dc_rcp(dreg[t0 + u] - mu);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double tt = wreg[t0 + u] * r[u];
                        const double tp = fmin(tt, 0.), tq = fmax(tt, 0.);
                        psi += tp;
                        dpsi = __builtin_fma(tp, r[u], dpsi);
                        phi += tq;
                        dphi = __builtin_fma(tq, r[u], dphi);
                    }
                }
            }
        } else if (!done) {
            // the poles up to j feed psi, the rest phi.  ONE loop with a wavefront-uniform trip
            // count (k is the team's) over this lane's poles, four at a time: unconditional
            // loads from a clamped index, the psi / phi split and the end of the list by
            // selects.  (Written as two runs with per-lane bounds -- up to j, beyond j -- the
            // loop compiled to an exec-mask loop with every load under its own branch and its
            // own wait.)  Adding the zeros of the other run leaves each sum as it was.
            for (int t0 = 0; t0 < npl; t0 += 4) {
                double rr[4], ww[4], dd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ic = min(sub + (t0 + u) * LPR, k - 1);
                    ww[u] = W.w2[ic];
                    dd[u] = W.dl[ic];
                }
                // (all four requests are out before the first value is waited for: left to
                // itself the scheduler, short of registers in this kernel, reuses one register
                // quad for all four loads and waits after each)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int i = sub + (t0 + u) * LPR;
                    const bool in = i < k;
                    ww[u] = in ? ww[u] : 0.;
                    rr[u] = dc_rcp(in ? (dd[u] - dorg) - mu : 1.);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double t = ww[u] * rr[u];
                    const double tp = fmin(t, 0.), tq = fmax(t, 0.);
                    psi += tp;
                    dpsi = __builtin_fma(tp, rr[u], dpsi);
                    phi += tq;
                    dphi = __builtin_fma(tq, rr[u], dphi);
                }
            }
        }
        psi = dc_quad_sum<LPR>(psi);
        dpsi = dc_quad_sum<LPR>(dpsi);
        phi = dc_quad_sum<LPR>(phi);
        dphi = dc_quad_sum<LPR>(dphi);
        if (!done) {
            const double fabs_ = phi - psi;
            psi *= rho; dpsi *= rho; phi *= rho; dphi *= rho;
            const double f = 1. + psi + phi;
            const double err = 8. * DC_EPS * (1. + rho * fabs_ * (1. + k));
            if (fabs(f) <= err) {
                done = true;
#ifdef BBO_EIG_SECULAR_HIST
                hist_it = it + 1;
#endif
            } else {
                if (f < 0.) lo = mu;
                else hi = mu;
                // two-pole rational model around the bracketing poles (middle way)
                const double dlp = (dj - dorg) - mu;
                double nmu = 0.5 * (lo + hi);
                if (last) {
                    const double aa = dpsi * dlp * dlp, ss = psi - dpsi * dlp;
                    const double c0 = 1. + ss + phi;
                    const double eta = dlp + aa * dc_rcp(c0);
                    const double cand = mu + eta;
                    if (eta == eta && cand > lo && cand < hi) nmu = cand;
                } else {
                    const double drp = (dn - dorg) - mu;
                    const double aa = dpsi * dlp * dlp, ss = psi - dpsi * dlp;
                    const double bb = dphi * drp * drp, rr = phi - dphi * drp;
                    const double c0 = 1. + ss + rr;
                    const double A1 = -(c0 * (dlp + drp) + aa + bb);
                    const double A0 = c0 * dlp * drp + aa * drp + bb * dlp;
                    // (the model only proposes the next iterate, the bracket and the residual
                    // test decide: hardware estimates + Newton, not the IEEE sequences)
                    const double disc = fmax(A1 * A1 - 4. * c0 * A0, 0.);
                    const double qq = -0.5 * (A1 + (A1 >= 0. ? 1. : -1.) * dc_sqrt(disc));
                    const double e1 = qq * dc_rcp(c0), e2 = A0 * dc_rcp(qq);
                    const double c1 = mu + e1, c2 = mu + e2;
                    if (e1 == e1 && c1 > lo && c1 < hi) nmu = c1;
                    else if (e2 == e2 && c2 > lo && c2 < hi) nmu = c2;
                }
                if (!(hi - lo > 4. * DC_EPS * fmax(fabs(lo), fabs(hi)))) {
                    done = true;
#ifdef BBO_EIG_SECULAR_HIST
                    hist_it = it + 1 + 100;      // (ended by the bracket, not the residual)
#endif
                } else {
                    // A model step (not a bisection) shorter than 2^-22 of the distance to the
                    // nearer pole is the last one: the iteration converges quadratically, the
                    // step that would follow is below the rounding of mu, and the evaluation
                    // that would only confirm it is a sixth of the solve (round 4; on the
                    // merges of scripts/dev_secular_model.py the accepted offset is, bit for
                    // bit, the one the confirming evaluation accepts).
                    const double step = fabs(nmu - mu);
                    const double dist = fmin(fabs((dj - dorg) - nmu), last ? fabs(nmu) : fabs((dn - dorg) - nmu));
                    if (nmu != 0.5 * (lo + hi) && step <= 0x1p-22 * dist) {
                        done = true;
#ifdef BBO_EIG_SECULAR_HIST
                        hist_it = it + 1;
#endif
                    }
                    mu = nmu;
                }
            }
        }
        // every wavefront leaves when ITS roots are done (the loop reads the merge's vectors
        // and writes nothing shared): no workgroup barrier per iteration, and a finished
        // wavefront leaves its SIMD's issue slots to the one still iterating
        if (__ballot(!done) == 0ull) {
            if (stamps && stamp_here && threadIdx.x == 0) stamps[31] = it + 1;
            break;
        }
    }
    SEC_STAMP(33);
#ifdef BBO_EIG_SECULAR_HIST
    // (diagnostic build: histogram of the iterations per root of the widest merge in slots
    // 34..46, the root with the most in 47 as j * 1000 + iterations)
    if (stamps && act && sub == 0 && width > 64) {
        atomicAdd((unsigned long long*) &stamps[34 + min(hist_it, 12)], 1ull);
        atomicMax((unsigned long long*) &stamps[47], (unsigned long long) hist_it * 1000000ull + j * 1000ull + k);
    }
#endif
    if (act && sub == 0) {
        W.mu[j] = mu;
        W.org[j] = o;
        W.lam[j] = dorg;       // (the origin pole itself, for the Loewner products below)
        // root j lies in [d_j, d_j+1] -- in floating point too, for finite input -- which is what
        // puts the roots in ascending order for the output ranking below.  Where it does not
        // hold (NaN / Inf in the merge) the flag of the sequential deflation scan, consumed
        // by now, sends that ranking to the form that is a permutation whatever the data.
        const double lj = dorg + mu;
        if (!(lj >= dj && lj <= dn)) W.cnt[3] = 1;
    }
    if (on && k == 1 && rt == 0) {
        W.mu[0] = rho * W.w2[0];
        W.org[0] = 0;
    }
#undef SEC_STAMP
}

// One level of merges: the team `tm` merges the blocks [a, mid) and [mid, b) (inactive teams
// walk through the same barriers).  Q (LDS) holds the blocks' eigenvectors on its diagonal
// blocks (zeros elsewhere inside [a,b)^2), dv their eigenvalues, rho_in = e[mid-1].
// Fg (global) is scratch for the eigenvector factor of this merge (m x m).
// BIG: instantiations for 256 < n <= 512 (two passes over a level's merges, merges the
// register-resident product cannot hold); the n <= 256 ones compile exactly as before
template<int LPR, bool do_gemm, bool BIG = false, bool WIDE = true>
__device__ inline void dc_merge_level(const DcMat &Q, const DcTeam &tm, int a, int mid, int b,
        double rho_in, double *dv, double *Fg, const DcWork &W0, long long *stamps,
        int mlevel = 0, double *Tg = nullptr,     // widest merge of this level; m x m global scratch
        int part = 0)       // 1: stop before the secular equation, 2: start behind it (cma_eig_secular)
{
#define MG_STAMP(slot) do { if (stamps && threadIdx.x == 0 && a == 0) stamps[slot] = wall_clock64(); } while (0)
    MG_STAMP(24);
    const int lane = threadIdx.x & 63;
    const int ttid = tm.ttid, TT = tm.tthreads;
    const bool on = tm.active != 0;
    const bool wv = tm.wave_scope != 0;
    const int m = on ? b - a : 0;
    // trips of the binary-search rankings: log2(longest list) + 1, the same for every team
    const int search_trips = 32 - __builtin_clz(max(mlevel > 0 ? mlevel : (BIG ? 512 : 256), 1));
    const double sgn = rho_in >= 0. ? 1. : -1.;
    // this merge's slices of the shared work arrays
    DcWork W = W0;
    W.dS += a; W.zS += a; W.dl += a; W.w2 += a; W.ws += a; W.mu += a; W.what += a; W.lam += a;
    W.ninv += a; W.rotc += a; W.rots += a; W.srcS += a; W.kp += a; W.dp += a; W.org += a;
    W.outpos += a; W.rotp += a; W.rotj += a; W.rowmap += a; W.colroot += a;
    W.cnt += 4 * tm.id;

    // (the part of a merge in front of the secular equation, as a unit: the top merge of a matrix
    // wider than 128 can stop behind it and hand the roots to several workgroups)
    auto front = [&]() -> double {
    // ---- poles and z, ascending -------------------------------------------------------
    double zi = 0., di = 0.;
    // The binary-search ranking below is a permutation only if EACH input list ascends in dc_key.
    // Finite output of a merge does (rankings by key); a block contaminated by NaN / Inf need not
    // (keys [inf, 1] against [2] would rank 1, 1, 0 and leave a slot of the index maps unwritten:
    // the fault class of DESIGN.md section 4).  Every entry tests its own predecessor; a violation
    // rides on the reduction of |z|^2 (no extra barrier) and selects ranking by counting.
    bool unsorted = false;
    if (ttid < m) {
        const int col = a + ttid;
        di = dv[col];
        zi = col < mid ? Q(mid - 1, col) : sgn * Q(mid, col);
        if (tm.sorted_in) {
            const double dprev = dv[max(col - 1, a)];
            unsorted = ttid > 0 && col != mid && dc_key(dprev) > dc_key(di);
        }
    }
    bool any_unsorted;
    const double zn2 = dc_team_sum_flag(ttid < m ? zi * zi : 0., unsorted, any_unsorted, W.red, tm);
    const double zn = dc_sqrt(zn2);
    const double rho = fabs(rho_in) * zn2;
    if (ttid < m) {
        zi *= dc_rcp(zn);
        W.lam[ttid] = di;          // unsorted copies for the ranking below
        W.what[ttid] = zi;
    }
    dc_sync(wv);
    if (ttid < m) {
        const int m1 = mid - a;
        const bool by_search = tm.sorted_in && !any_unsorted;
        const int r = !by_search ? dc_rank_of(W.lam, m, dc_key(di), ttid)
                : dc_rank_sorted2(W.lam, m1, m - m1, dc_key(di), ttid, search_trips);
        W.dS[r] = di;
        W.zS[r] = zi;
        W.srcS[r] = a + ttid;
    }
    const double dmax = dc_team_max(ttid < m ? fabs(di) : 0., W.red, tm);
    const double zmax = dc_team_max(ttid < m ? fabs(zi) : 0., W.red, tm);
    const double tol = 8. * DC_EPS * fmax(dmax, zmax);
    dc_sync(wv);

    MG_STAMP(25);
    // ---- deflation (LAPACK dlaed2's rules).  Fast path, all lanes: drop the poles with a
    // negligible z, keep the others in order, and test every pair of NEIGHBOURING kept poles
    // for the close-pole rotation; only if some pair asks for one does lane 0 redo the scan
    // sequentially (the rotations chain).  Same outcome as the scan whenever none fires.
    const bool alltiny = rho * zmax <= tol;
    {
        const bool mine = ttid < m;
        const bool tiny = mine && (alltiny || rho * fabs(W.zS[ttid]) <= tol);
        const unsigned long long bk = __ballot(mine && !tiny), bd = __ballot(tiny);
        if (lane == 0) {
            W.balk[tm.wave0 + tm.twave] = bk;
            W.bald[tm.wave0 + tm.twave] = bd;
        }
        if (ttid == 0) W.cnt[3] = 0;
        dc_sync(wv);
        int pk = 0, pd = 0, tk = 0, td = 0;
        for (int w = 0; w < tm.nwaves; w++) {
            const int ck = __popcll(W.balk[tm.wave0 + w]), cd = __popcll(W.bald[tm.wave0 + w]);
            if (w < tm.twave) {
                pk += ck;
                pd += cd;
            }
            tk += ck;
            td += cd;
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        pk += __popcll(bk & below);
        pd += __popcll(bd & below);
        if (mine && !tiny) W.kp[pk] = ttid;
        if (tiny) W.dp[pd] = ttid;
        if (on && ttid == 0) {
            W.cnt[0] = tk;
            W.cnt[1] = td;
            W.cnt[2] = 0;
        }
        dc_sync(wv);
        if (on && ttid >= 1 && ttid < tk) {
            const int pj = W.kp[ttid - 1], j = W.kp[ttid];
            const double zj = W.zS[j], zp = W.zS[pj];
            const double t = W.dS[j] - W.dS[pj];
            if (fabs(t * zj * zp) <= tol * (zj * zj + zp * zp)) W.cnt[3] = 1;
        }
        dc_sync(wv);
    }
    if (on && ttid == 0 && W.cnt[3]) {
        int k = 0, nd = 0, nr = 0;
        {
            int pj = -1;
            for (int j = 0; j < m; j++) {
                const double zj = W.zS[j];
                if (rho * fabs(zj) <= tol) {
                    W.dp[nd++] = j;
                    continue;
                }
                if (pj < 0) {
                    pj = j;
                    continue;
                }
                // |t c s| <= tol with c = z_j / tau, s = -z_pj / tau, tau^2 = z_j^2 + z_pj^2,
                // tested without forming tau
                const double zp = W.zS[pj];
                const double t = W.dS[j] - W.dS[pj];
                if (fabs(t * zj * zp) <= tol * (zj * zj + zp * zp)) {
                    const double tau = hypot(zj, zp);
                    const double cth = zj / tau, s = -zp / tau;
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
        atomicMax(W.maxnr, nr);
    }
    dc_sync(wv);
    const int k = on ? W.cnt[0] : 0, nr = on ? W.cnt[2] : 0;

    // kept poles ascending (a rotation may perturb the order by rounding)
    if (ttid < k) {
        const int pos = W.kp[ttid];
        const double dk = W.dS[pos];
        // (without a rotation the kept poles are a subsequence of the sorted ones: in order)
        int r = ttid;
        if (nr > 0) {
            r = 0;
            const double kk = dc_key(dk);
            for (int j = 0; j < k; j++) {
                const double dj = dc_key(W.dS[W.kp[j]]);
                r += (dj < kk) || (dj == kk && j < ttid);
            }
        }
        const double z = W.zS[pos];
        W.dl[r] = dk;
        W.ws[r] = z;
        W.w2[r] = z * z;
        W.org[r] = pos;            // temporarily: sorted position of kept pole r
    }
    dc_sync(wv);
    if (ttid < k) W.kp[ttid] = W.org[ttid];
    dc_sync(wv);
    return rho;
    };     // front
    double rho = 0.;
    if (part != 2) rho = front();
    const int k = on ? W.cnt[0] : 0, nd = on ? W.cnt[1] : 0, nr = on ? W.cnt[2] : 0;
    // (rotations anywhere on this level; a one-wavefront team shares no barrier with the others
    // and goes by its own count)
    const int maxnr = wv ? nr : *W.maxnr;

    MG_STAMP(26);
    if (part == 1) {
        // (the caller saves the work area; rho travels in a free slot of the reduction scratch)
        if (on && ttid == 0) W.red[14] = rho;
        dc_sync(wv);
        return;
    }
    // ---- secular equation: LPR lanes per root (dc_secular) ---------------------------------
    if (part == 0) {
        dc_secular<LPR>(W, k, rho, ttid, on, stamps, a == 0, b - a);
        dc_sync(wv);
    }

    MG_STAMP(27);
    // ---- Loewner z and column norms -------------------------------------------------------
    {
        const int i = ttid / LPR, sub = ttid % LPR;
        double prod = 1.;
        if (on && k > 1 && i < k) {
            // lam_jj - d_i = (d_org - d_i) + mu, paired with a denominator d_jj - d_i.  Four roots at
            // a time, their three values each requested together (the origin poles were left in
            // W.lam by the secular stage: no index chain), four partial products; the trip count
            // is the wavefront's.  (A plain loop over jj was a chain of four dependent LDS reads
            // and one product per root.)
            const double di2 = W.dl[i];
            const int npl_l = __builtin_amdgcn_readfirstlane((k + LPR - 1) / LPR);
            double p4[4] = { 1., 1., 1., 1. };
            for (int t0 = 0; t0 < npl_l; t0 += 4) {
                double lo_[4], mu_[4], dj_[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int jc = min(sub + (t0 + u) * LPR, k - 1);
                    lo_[u] = W.lam[jc];
                    mu_[u] = W.mu[jc];
                    dj_[u] = W.dl[jc];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int jj = sub + (t0 + u) * LPR;
                    const double num = (lo_[u] - di2) + mu_[u];
                    const double f = num * dc_rcp(jj == i ? 1. : dj_[u] - di2);
                    p4[u] *= jj < k ? f : 1.;
                }
            }
            prod = (p4[0] * p4[1]) * (p4[2] * p4[3]);
        }
        if (LPR >= 2) prod *= eig_quad_xor1(prod);
        if (LPR >= 4) prod *= eig_quad_xor2(prod);
        if (on && k > 1 && i < k && sub == 0) {
            const double v = sqrt(fabs(prod));
            W.what[i] = W.ws[i] >= 0. ? v : -v;
        }
        if (on && k == 1 && ttid == 0) {
            W.what[0] = 1.;
            W.ninv[0] = 1.;
        }
        dc_sync(wv);
        const int j = i;
        double ss = 0.;
        if (on && k > 1 && j < k) {
            const double dorgj = W.lam[j], muj = W.mu[j];
            const int npl_l = __builtin_amdgcn_readfirstlane((k + LPR - 1) / LPR);
            for (int t0 = 0; t0 < npl_l; t0 += 4) {
                double wh[4], dd[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ic = min(sub + (t0 + u) * LPR, k - 1);
                    wh[u] = W.what[ic];
                    dd[u] = W.dl[ic];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int ii = sub + (t0 + u) * LPR;
                    const double s = wh[u] * dc_rcp((dd[u] - dorgj) - muj);
                    ss += ii < k ? s * s : 0.;
                }
            }
        }
        ss = dc_quad_sum<LPR>(ss);
        if (on && k > 1 && j < k && sub == 0) W.ninv[j] = 1. / sqrt(ss);
    }
    dc_sync(wv);

    MG_STAMP(28);
    // ---- all eigenvalues, their output order, the row / column maps of F ---------------------
    if (ttid < k) W.lam[ttid] = W.dl[W.org[ttid]] + W.mu[ttid];
    if (ttid < nd) W.lam[k + ttid] = W.dS[W.dp[ttid]];
    dc_sync(wv);
    // (the roots come in ascending order -- each lies on its side of the pole it shares with its
    // neighbour, in floating point too -- and so do the deflated poles unless a rotation changed them)
    if (ttid < m) {
        const double key = dc_key(W.lam[ttid]);
        W.outpos[ttid] = (nr > 0 || W.cnt[3] != 0) ? dc_rank_of(W.lam, m, key, ttid)
                : dc_rank_sorted2(W.lam, k, nd, key, ttid, search_trips);
    }
    dc_sync(wv);
    if (ttid < k) {
        W.rowmap[W.srcS[W.kp[ttid]] - a] = ttid;
        W.colroot[W.outpos[ttid]] = ttid;
    }
    if (ttid < nd) {
        W.rowmap[W.srcS[W.dp[ttid]] - a] = -(1 + W.outpos[k + ttid]);
        W.colroot[W.outpos[k + ttid]] = -1;
    }
    dc_sync(wv);
    // The in-place product below keeps F as MFMA B fragments in registers: when no deflation
    // rotation fired anywhere on this level (the common case) every lane forms ITS fragment
    // entries straight from the merge's vectors -- same expression, same bits -- and F never
    // travels through global memory (a store, a fence, two barriers and a load per merge).
    constexpr bool FRAG_OK = do_gemm;
    const bool generic_prod = BIG && do_gemm && (mlevel > 4 * DC_KSTEPS || ((mlevel + 15) >> 4) > 2 * tm.nwaves);
    const bool direct = FRAG_OK && maxnr == 0 && !generic_prod;
    if (!direct) {
    // F (rows in ORIGINAL column order), written in one pass.  A thread keeps ONE column (its
    // root's origin pole, offset and norm in registers) and walks down the rows, four at a time
    // (the row map and the poles are broadcast reads, the stores of a wavefront are consecutive).
    // (Element by element -- an integer division and a chain of five dependent LDS reads each --
    // this pass took 44 us of the n = 256 top merge.)
    if (m > 0) {
        const int ncg = TT >= m ? TT / m : 1;              // row groups of the team
        for (int c0 = 0; c0 < m; c0 += TT) {
            const int cidx = c0 + (TT >= m ? ttid % m : ttid);
            const int rg = TT >= m ? ttid / m : 0;
            if (cidx < m && rg < ncg) {
                const int j = W.colroot[cidx];
                const double dorgj = j >= 0 ? W.dl[W.org[j]] : 0.;
                const double muj = j >= 0 ? W.mu[j] : 0.;
                const double nj = j >= 0 ? W.ninv[j] : 0.;
                for (int r0 = rg; r0 < m; r0 += 4 * ncg) {
                    int iv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) iv[u] = W.rowmap[min(r0 + u * ncg, m - 1)];
                    __builtin_amdgcn_sched_barrier(0);
                    double wh[4], dd[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int ic = iv[u] >= 0 ? iv[u] : 0;
                        wh[u] = W.what[ic];
                        dd[u] = W.dl[ic];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int r = r0 + u * ncg;
                        const int i = iv[u];
                        const double root = k == 1 ? 1. : wh[u] * dc_rcp((dd[u] - dorgj) - muj) * nj;
                        const double v = i >= 0 ? (j >= 0 ? root : 0.) : (-(1 + i) == cidx ? 1. : 0.);
                        if (r < m) Fg[(size_t) r * m + cidx] = v;
                    }
                }
            }
        }
    }
    dc_sync(wv);
    // deflation rotations, in reverse: Q G with G = [[c, -s], [s, c]] on sorted columns (p, j)
    for (int r = maxnr - 1; r >= 0; r--) {
        if (r < nr) {
            const int rp = W.srcS[W.rotp[r]] - a, rj = W.srcS[W.rotj[r]] - a;
            const double cth = W.rotc[r], s = W.rots[r];
            for (int cidx = ttid; cidx < m; cidx += TT) {
                const double x = Fg[(size_t) rp * m + cidx], y = Fg[(size_t) rj * m + cidx];
                Fg[(size_t) rp * m + cidx] = cth * x - s * y;
                Fg[(size_t) rj * m + cidx] = s * x + cth * y;
            }
        }
        dc_sync(wv);
    }
    __threadfence_block();
    dc_sync(wv);
    }   // !direct

    MG_STAMP(29);
    // (the top merge of a matrix wider than 128 leaves F in global memory: the two products
    // Q F and Q_house (Q F) are separate whole-GPU kernels, cma_eig_gemm)
    if (generic_prod) {
    // ---- merges the register-resident form below cannot hold (n > 256 only): it keeps the F
    // fragments of at most TWO 16-column tiles per wavefront of the team, 128 deep -- enough for
    // every level of n <= 256 (m <= 32 x the team's wavefronts), not for 32 leaves on 8
    // wavefronts (m = 35 or 64 on one, 128 on two, 256 on four).  Here every 16 x 16 tile of the
    // product is formed by itself, operands straight from global memory (four k-steps requested
    // ahead), into the scratch Tg, then copied back.  The same for every team of the level.
    const int fr = lane & 15, fk = lane >> 4;
    const int ntl = (mlevel + 15) >> 4, ntile = (m + 15) >> 4;
    const int ksteps = (m + 3) >> 2;
    for (int t = tm.twave; t < ntl * ntl; t += tm.nwaves) {
        const int rt = t / ntl, ct = t - rt * ntl;
        if (!on || rt >= ntile || ct >= ntile) continue;
        dc_d4 acc = { 0., 0., 0., 0. };
        const int arow = rt * 16 + fr, col = ct * 16 + fr;
        for (int ks0 = 0; ks0 < ksteps; ks0 += 4) {
            double av[4], bv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int kk = 4 * (ks0 + u) + fk;
                av[u] = (arow < m && kk < m) ? Q(a + arow, a + kk) : 0.;
                bv[u] = (kk < m && col < m) ? Fg[(size_t) kk * m + col] : 0.;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = rt * 16 + (lane >> 4) + 4 * r;
            if (row < m && col < m) Tg[(size_t) row * m + col] = acc[r];
        }
    }
    __threadfence_block();
    dc_sync(wv);
    for (int q = ttid; q < m * m; q += TT) {
        const int r = q / m, c = q - r * m;
        Q(a + r, a + c) = Tg[q];
    }
    } else if (do_gemm) {
    // ---- Q[a:b, a:b] <- Q[a:b, a:b] F on the matrix cores, 16 rows at a time, in place.
    // A wavefront keeps the F fragments of its column tile(s) in registers for the whole
    // merge; every wavefront finishes reading the 16 old rows before any of them is stored.
    const int fr = lane & 15, fk = lane >> 4;
    const int ntile = (m + 15) >> 4;
    const int ksteps = (m + 3) >> 2;
    // (WIDE = false, the matrix in LDS, n <= 128: a team always has a wavefront per column tile)
    constexpr int NT = WIDE ? 2 : 1;
    double bfrag[NT][DC_KSTEPS];
    if (direct) {
#pragma unroll
        for (int u = 0; u < NT; u++) {
            const int ct = tm.twave + u * tm.nwaves;
            const int col = ct * 16 + fr;
            const bool cok = on && ct < ntile && col < m;
            const int j = cok ? W.colroot[col] : -1;
            const double dorgj = j >= 0 ? W.dl[W.org[j]] : 0.;
            const double muj = j >= 0 ? W.mu[j] : 0.;
            const double nj = j >= 0 ? W.ninv[j] : 0.;
#pragma unroll
            for (int ks = 0; ks < DC_KSTEPS; ks++) {
                const int kk = 4 * ks + fk;
                double v = 0.;
                if (cok && kk < m) {
                    const int i = W.rowmap[kk];
                    if (i >= 0) {
                        if (j >= 0) v = k == 1 ? 1. : W.what[i] * dc_rcp((W.dl[i] - dorgj) - muj) * nj;
                    } else if (-(1 + i) == col) {
                        v = 1.;
                    }
                }
                bfrag[u][ks] = v;
            }
        }
    } else {
#pragma unroll
    for (int u = 0; u < NT; u++) {
        const int ct = tm.twave + u * tm.nwaves;
        const int col = ct * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < DC_KSTEPS; ks++) {
            const int kk = 4 * ks + fk;
            bfrag[u][ks] = (on && ct < ntile && kk < m && col < m) ? Fg[(size_t) kk * m + col] : 0.;
        }
    }
    }
    MG_STAMP(34);
    // every team walks the same number of row tiles (the widest merge of the level).  Q is still
    // BLOCK DIAGONAL here (Q_1 on [a, mid), Q_2 on [mid, b), zeros elsewhere): a row tile inside
    // one block only has its block's columns to contract over -- half the k-steps of the merge
    // multiply stored zeros, and skipping them leaves every sum as it was (x + 0 * f)
    const int m1 = mid - a;
    const int rtiles = mlevel > 0 ? min((mlevel + 15) >> 4, (DC_KSTEPS * 4) / 16) : (DC_KSTEPS * 4) / 16;
    for (int rt = 0; rt < rtiles; rt++) {
        dc_d4 acc[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) acc[u] = dc_d4 { 0., 0., 0., 0. };
        const int arow = rt * 16 + fr;
        // (the team's: scalars for the branches below)
        const int klo = __builtin_amdgcn_readfirstlane(16 * rt >= m1 ? m1 >> 2 : 0);
        const int khi = __builtin_amdgcn_readfirstlane(16 * rt + 16 <= m1 ? (m1 + 3) >> 2 : ksteps);
        if (rt < ntile && tm.twave < ntile) {
            // A batch of k-steps at a time: their 16 x 4 operand pieces are REQUESTED first, from
            // in-bounds (clamped) addresses, and zeroed by selects where they lie outside the merge;
            // then the products run back to back.  (With the load of a piece under the test for
            // its position, each k-step was a branch, a load, a wait and one product: 6.6 k cycles
            // per row tile where the products need 1 k.)
            const int rowc = a + min(arow, m - 1);
            const bool two = WIDE && tm.twave + tm.nwaves < ntile;
            // (CH k-steps per batch: 16 where the registers allow it, 8 next to two tiles' F fragments)
            constexpr int CH = WIDE ? 8 : 16;
#pragma unroll
            for (int h = 0; h < DC_KSTEPS / CH; h++) {
                if (CH * h < khi && CH * h + CH > klo) {
                    double av[CH];
#pragma unroll
                    for (int e = 0; e < CH; e++) av[e] = Q(rowc, a + min(4 * (CH * h + e) + fk, m - 1));
                    __builtin_amdgcn_sched_barrier(0);
                    // (a second column tile only when the merge is wider than 16 columns per
                    // wavefront of its team -- uneven splits; for n = 128 every wavefront has one)
                    if (two) {
#pragma unroll
                        for (int e = 0; e < CH; e++) {
                            const int ks = CH * h + e;
                            if (ks >= klo && ks < khi) {
                                const double x = (arow < m && 4 * ks + fk < m) ? av[e] : 0.;
                                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bfrag[0][ks], acc[0], 0, 0, 0);
                                acc[NT - 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bfrag[NT - 1][ks], acc[NT - 1], 0, 0, 0);
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < CH; e++) {
                            const int ks = CH * h + e;
                            if (ks >= klo && ks < khi) {
                                const double x = (arow < m && 4 * ks + fk < m) ? av[e] : 0.;
                                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, bfrag[0][ks], acc[0], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        dc_sync(wv);
        if (rt < ntile) {
#pragma unroll
            for (int u = 0; u < NT; u++) {
                const int col = (tm.twave + u * tm.nwaves) * 16 + fr;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = rt * 16 + (lane >> 4) + 4 * r;
                    if (row < m && col < m) Q(a + row, a + col) = acc[u][r];
                }
            }
        }
    }
    MG_STAMP(35);
    }   // do_gemm
    dc_sync(wv);
    if (ttid < m) dv[a + W.outpos[ttid]] = W.lam[ttid];
    dc_sync(wv);
    MG_STAMP(30);
#undef MG_STAMP
}

// ---- leaf: the QL recurrence on the s x s tridiagonal block, one wavefront --------------
// ws: per-wavefront LDS scratch, >= 2 * 20 doubles + 64 double2 + 3 * 64 ints
__device__ inline void dc_leaf_ql(const DcMat &Q, int a, int s, const double *dv_in,
        const double *ev_in, double *dv_out, double *ws, int lane, long long *dbgout)
{
    // (d, e) of the block in registers, lane i its row i (ql_produce_reg); LDS: the Givens pairs
    // and sweep descriptors of a round, for the consumer
    double2 *rot = reinterpret_cast<double2*>(ws + 42);
    int *desc = reinterpret_cast<int*>(ws + 42 + 128);
    double d = lane < s ? dv_in[a + lane] : 0.;
    double e = lane + 1 < s ? ev_in[a + lane] : 0.;
    for (int q = lane; q < s * s; q += 64) {
        const int r = q / s, c = q - r * s;
        Q(a + r, a + c) = r == c ? 1. : 0.;
    }
    dc_wave_sync();
    EigMat blk { &Q(a, a), Q.ld };
    QlState st { 0, 0, 1, 0, 0., 0., 0 };
    // QL has no iteration limit in the reference; ql_produce_reg<true> stops after 30 sweeps per
    // eigenvalue (a uniform counter), which only bounds a run on non-finite / subnormal input
    // (a lane touches only its own row of the block: the rotations need no fence among lanes)
    const int sweeps = ql_produce_reg<true>(st, s, d, e, rot, desc, 64, lane, blk);
    if (dbgout && lane == 0) dbgout[0] = sweeps;
    if (lane < s) dv_out[a + lane] = d;
    dc_wave_sync();
}

// two leaves by one wavefront, one in each 32-lane half (ql_leaf_pair): blocks [a0, a0 + s0) and
// [a1, a1 + s1) (s1 = 0: none), s <= 32
__device__ inline void dc_leaf_ql_pair(const DcMat &Q, int a0, int s0, int a1, int s1,
        const double *dv_in, const double *ev_in, double *dv_out, int lane)
{
    const int hl = lane & 31;
    const int a = lane < 32 ? a0 : a1, s = lane < 32 ? s0 : s1;
    double d = hl < s ? dv_in[a + hl] : 0.;
    double e = hl + 1 < s ? ev_in[a + hl] : 0.;
    for (int q = hl; q < s * s; q += 32) {
        const int r = q / s, c = q - r * s;
        Q(a + r, a + c) = r == c ? 1. : 0.;
    }
    dc_wave_sync();
    double *zrow = &Q(a + (hl < s ? hl : 0), a);
    ql_leaf_pair(s, d, e, zrow, lane);
    if (hl < s) dv_out[a + hl] = d;
    dc_wave_sync();
}

// T factors of the reflector panels (see dc_apply_reflectors): T_b for panel b = reflectors
// 16b .. 16b+15, one wavefront per panel, written to Tg[b][16][16].  V: global n x n (row i =
// u_i, zero from column i on), tau[i] = 1 / h_i or 0.  scratch (LDS): 16 x 17 doubles per
// wavefront.  All threads of the workgroup call it.
__device__ __forceinline__ void dc_build_T(int n, const double *V, const double *tau, double *Tg,
        double *scratch, int ldv = 0)      // ldv: row stride of V (0: dense, n)
{
    if (ldv == 0) ldv = n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int npanel = (n + 15) >> 4;
    constexpr int LS = 17;
    for (int b = wave; b < npanel; b += NW) {
        double *Sb = scratch + (size_t) wave * 16 * LS;
        const int i0 = 16 * b, reach = min(n, i0 + 16);       // rows < reach can be non-zero
        const int refl = i0 + fr;
        const double *vrow = V + (size_t) min(refl, n - 1) * ldv;
        dc_d4 acc = { 0., 0., 0., 0. };
        for (int r0 = 0; r0 < reach; r0 += 4 * DC_KSTEPS) {
            // (all requests first, from clamped addresses; zeros by selects: a load under its
            // position test is a branch, a load and a wait of an L2 round trip -- each)
            double va[DC_KSTEPS];
#pragma unroll
            for (int q = 0; q < DC_KSTEPS / 2; q++) {
                const int k0 = r0 + 8 * q + 2 * fk;
                va[2 * q] = vrow[min(k0, n - 1)];
                va[2 * q + 1] = vrow[min(k0 + 1, n - 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < DC_KSTEPS / 2; q++) {
                const int k0 = r0 + 8 * q + 2 * fk;
                va[2 * q] = (refl < n && k0 < reach) ? va[2 * q] : 0.;
                va[2 * q + 1] = (refl < n && k0 + 1 < reach) ? va[2 * q + 1] : 0.;
            }
#pragma unroll
            for (int q = 0; q < DC_KSTEPS / 2; q++) {
                if (r0 + 8 * q < reach) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[2 * q], va[2 * q], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[2 * q + 1], va[2 * q + 1], acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) Sb[(fk + 4 * r) * LS + fr] = acc[r];    // S = V_b^T V_b
        dc_wave_sync();
        if (lane < 16) {
            double t[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const double ti = i0 + i < n ? tau[i0 + i] : 0.;
                double sum = 0.;
#pragma unroll
                for (int m = 0; m < i; m++) sum = fma(t[m], Sb[m * LS + i], sum);
                t[i] = lane < i ? -ti * sum : (lane == i ? ti : 0.);
            }
#pragma unroll
            for (int i = 0; i < 16; i++) Tg[(size_t) b * 256 + lane * 16 + i] = t[i];
        }
        dc_wave_sync();
    }
}

// ---- B = H(n-1) ... H(1) Q_T, the reflectors applied in blocked (compact WY) form on the matrix
// cores.  V (global, n x n, row i = u_i, zero from column i on), tau[i] = 1 / h_i (0: no
// reflector), Q (LDS) = Q_T on entry.  Panel b = reflectors 16b .. 16b+15:
//     H(16b+15) ... H(16b) = I - V_b T_b^T V_b^T,   T_b upper triangular (LAPACK dlarft, forward),
//     T(0:i, i) = -tau_i T(0:i, 0:i) (V_b^T V_b)(0:i, i),  T(i, i) = tau_i.
// 1. Wavefront w builds T_w: Gram matrix by MFMA (all of its reflector entries requested up
//    front: V sits in L2), written to its LDS patch; then lane l < 16 owns ROW l of T in
//    registers and runs the 16-step recurrence on its own (it needs only its row and column i of
//    the Gram matrix, a broadcast read): no wavefront hand-overs.  T_w goes to global (Tg).
// 2. Panels in the order the reflectors act.  The 16 x n panel is staged in LDS once for all
//    wavefronts (the next one is prefetched into registers meanwhile); wavefront w owns the
//    16-column tile w of Q: W = V_b^T Q, W <- T_b^T W (the accumulator layout of one product IS
//    the B-operand layout of the next), Q -= V_b W; rows beyond the reflectors' reach skipped.
// k index of a contraction over rows: k(ks, fk) = 8 (ks >> 1) + 2 fk + (ks & 1), so a lane's two
// k-steps read one 16-byte piece of a reflector.
// scratch (LDS): 16 * 132 doubles; Tg (global): 16 * 256 doubles.
template<int TT>
__device__ inline void dc_apply_reflectors(const DcMat &Q, int n, const double *V, const double *tau,
        double *Tg, double *scratch, double *Bout, int ldb, const int *outpos,
        long long *stamps = nullptr, bool t_ready = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int npanel = (n + 15) >> 4;
    constexpr int LDV = 132;
    // (outpos, if any, lives in the work area this stage reuses: n <= 16 then, one entry each)
    const int outpos_col = (outpos && 16 * wave + fr < n) ? outpos[16 * wave + fr] : -1;
    __syncthreads();      // the merge work area is free; tau and V are visible
    if (!t_ready) dc_build_T(n, V, tau, Tg, scratch);
    // (T is written and read by THIS workgroup: workgroup scope.  A device-scope fence here
    // waits for the whole chip's write traffic when 256 workgroups reach it together -- it was
    // 70 us of the 127 this stage took in a 256-population batch, against 55 for one population)
    __threadfence_block();
    __syncthreads();
    if (stamps && threadIdx.x == 0) stamps[36] = wall_clock64();
    // ---- 2. panels ---------------------------------------------------------------------------------
    double *Vp = scratch;
    const int ctile = wave, col = 16 * ctile + fr;
    // Reflector j of a panel is staged in LDS row vrow(j) = 4 (j & 3) + (j >> 2): the update's
    // operand A(row, k = 4 ks + fk) = V[4 ks + fk][row] then comes from LDS row 4 fk + ks, and the
    // four fk groups of a wavefront's read start 32 banks apart instead of 8 -- two passes per
    // read, the minimum for 64 x 8 bytes, instead of four; W = V^T Q reads row vrow(fr), the same
    // set of rows as before.
    auto vrow = [](int j) { return 4 * (j & 3) + (j >> 2); };
    // this thread's pieces of a staged panel (16 reflectors x 32 pieces of 4 columns: one piece
    // per thread of a 512-thread workgroup, two / four of a 256- / 128-thread one)
    constexpr int NPIECE = 512 / TT;
    double pre[NPIECE][4], tpre[4];
    auto prefetch = [&](int b) {
        const int i0 = 16 * b;
#pragma unroll
        for (int h = 0; h < NPIECE; h++) {
            const int e = tid + h * TT;
            const int pj = e >> 5, pc = (e & 31) * 4;
            const int rf = i0 + pj;
            const double *vr = V + (size_t) min(rf, n - 1) * n;
#pragma unroll
            for (int u = 0; u < 4; u++) pre[h][u] = vr[min(pc + u, n - 1)];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ks++) tpre[ks] = Tg[(size_t) b * 256 + (4 * ks + fk) * 16 + fr];
        // (raw values: what lies outside the panel is zeroed where the panel is staged, so that
        // nothing waits for these loads before the products of the panel in hand)
    };
    // The wavefront's 16-column tile of Q stays in REGISTERS for all panels: lane (fr, fk) holds
    // Q(16 rt + 4 r + fk, col) in qreg[rt][r] -- the accumulator layout of the update
    // Q -= V_b W for row tile rt AND the B-operand layout of k-step 4 rt + r of W = V_b^T Q, so the
    // same registers feed both products and Q is read from LDS once and written to B once.
    // (Until round 3 every product re-read its pieces of Q from LDS under position tests: a
    // branch, a load and a wait per k-step, 50 us for the eight panels of n = 128.)
    dc_d4 qreg[DC_KSTEPS / 4];
    {
        const int colc = min(col, n - 1);
#pragma unroll
        for (int rt = 0; rt < DC_KSTEPS / 4; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * rt + fk + 4 * r;
                const double x = Q(min(row, n - 1), colc);
                qreg[rt][r] = (row < n && col < n) ? x : 0.;
            }
    }
    prefetch(0);
    for (int b = 0; b < npanel; b++) {
        const int reach = min(n, 16 * b + 16);
        __syncthreads();                         // the previous panel has been read by everyone
#pragma unroll
        for (int h = 0; h < NPIECE; h++) {
            const int e = tid + h * TT;
            const int pj = e >> 5, pc = (e & 31) * 4;
            const bool rok = 16 * b + pj < n;
#pragma unroll
            for (int u = 0; u < 4; u++) Vp[vrow(pj) * LDV + pc + u] = (rok && pc + u < reach) ? pre[h][u] : 0.;
        }
        double tv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) tv[ks] = 4 * ks + fk <= fr ? tpre[ks] : 0.;
        __syncthreads();
        if (b + 1 < npanel) prefetch(b + 1);
        if (16 * ctile < n) {
            // W = V_b^T Q[:, tile]: A(m = fr, k = 16 rt + 4 r + fk) = reflector fr of the panel at
            // that row (zero beyond the panel's reach, like the rows of Q beyond n).
            // (Requesting the operand pieces of all eight row groups of both products up front --
            // 128 more registers -- was slower: 37 us for the panels against 27.)
            dc_d4 w = { 0., 0., 0., 0. };
#pragma unroll
            for (int rt = 0; rt < DC_KSTEPS / 4; rt++) {
                if (16 * rt < reach) {
                    double a4[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) a4[r] = Vp[vrow(fr) * LDV + 16 * rt + 4 * r + fk];
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        w = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[r], qreg[rt][r], w, 0, 0, 0);
                }
            }
            // W <- T_b^T W: A(m = fr, k = 4 ks + fk) = T(4 ks + fk, fr); B operand of k-step ks =
            // accumulator register ks of W
            dc_d4 w2 = { 0., 0., 0., 0. };
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[ks], w[ks], w2, 0, 0, 0);
            // Q[:, tile] -= V_b W, 16 rows at a time, in the registers
#pragma unroll
            for (int rt = 0; rt < DC_KSTEPS / 4; rt++) {
                if (16 * rt < reach) {
                    double a4[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) a4[ks] = -Vp[(4 * fk + ks) * LDV + 16 * rt + fr];   // = vrow(4 ks + fk)
#pragma unroll
                    for (int ks = 0; ks < 4; ks++)
                        qreg[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[ks], w2[ks], qreg[rt], 0, 0, 0);
                }
            }
        }
    }
    if (stamps && threadIdx.x == 0) stamps[37] = wall_clock64();
    // B straight from the registers (rows of 16 consecutive columns)
    if (col < n) {
        const int oc = outpos ? outpos_col : col;
#pragma unroll
        for (int rt = 0; rt < DC_KSTEPS / 4; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * rt + fk + 4 * r;
                if (row < n) {
                    Bout[(size_t) row * ldb + oc] = qreg[rt][r];
                    Q(row, col) = qreg[rt][r];      // (the caller packs the sampler's operand from the LDS copy)
                }
            }
    }
    __syncthreads();
}

// D&C driver.  On entry: dv = diagonal, ev[i] = coupling (i, i+1) (ev[n-1] = 0), Q (LDS) =
// Householder matrix Q_house.  On exit: dv = eigenvalues ascending, Bout (global, ld) =
// Q_house * Q_T, i.e. the eigenvectors of the original matrix in columns.
// G (global): 2 * n * n doubles of scratch.  scratch (LDS): >= 3400 doubles.
// ext_top != 0 (n > 128, Q in global memory): stop after the scalar part of the top merge; on
// exit Q = blockdiag(Q_1, Q_2), F (= G + n*n, n x n) = the top merge's eigenvector factor with
// columns in ascending eigenvalue order, and the caller forms B = Q_house (Q F).
// (forceinline: called once with Q in LDS and once with Q in global memory from cma_eigen -- as a
// shared out-of-line function it would see generic pointers and address everything with FLAT
// instructions)
template<int TT = 512, bool BIG = false, bool WIDE = true>
__device__ __forceinline__ void eig_dc_phase(const DcMat &Q, int n, double *dv, double *ev, double *G,
        double *Bout, int ldb, double *scratch, long long *stamps, int dbg, int ext_top = 0,
        const double *hv = nullptr, bool qh_ready = false, double *Tscratch = nullptr, int mode = 0,
        bool t_prebuilt = false,     // (mode 2: the reflector panels' T factors are in place already)
        int top_part = 0, double *Wimg = nullptr)   // (mode 2: 1 = up to the secular equation, the work
                                                     // area goes to Wimg; 2 = resume from Wimg behind it)
{
    // mode (round 4, 128 < n <= 256 split over workgroups): 0 = the whole decomposition;
    // 1 = a HALF of a torn matrix as a problem of its own (Q in LDS, no reflectors): leaves and all
    //     merges, then the eigenvectors of the tridiagonal block go to Bout as they are;
    // 2 = the TOP merge only: the two halves [0, n / 2) and [n / 2, n) arrive solved (dv = their
    //     eigenvalues in ascending order each, Q = their eigenvector blocks, global memory)
#define DC_STAMP(slot) do { if (stamps && threadIdx.x == 0) stamps[slot] = wall_clock64(); } while (0)
    DC_STAMP(16);
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int NW = T >> 6;
    double *Qh = G;                       // Q_house, n x n row-major
    double *F = G + (size_t) n * n;       // merge factors (one m x m slab per merge)
    constexpr int MAXB = BIG ? DC_MAXB_BIG : DC_MAXB;
    __shared__ int bounds[MAXB + 1];
    __shared__ int nblk_s;
    __shared__ int maxnr_s;
    __shared__ double scale_s;

    // (qh_ready: the reduction has left V in Qh itself -- at n = 256 this copy, 512 KB global to
    // global by one workgroup, took 54 us)
    if (!qh_ready)
    for (int r = wave; r < n; r += NW)
        for (int c = lane; c < n; c += 64) Qh[(size_t) r * n + c] = Q(r, c);
    // hv != null: Q holds the STASHED REFLECTORS of the Householder stage (row i = u_i, zero from
    // column i on; H(i) = I - u_i u_i^T / hv[i]), not the accumulated Q_house: Qh is V then, and
    // the scalars 1 / h move out of the way of the work area (hv lives inside `scratch`)
    double *taug = G + (size_t) 2 * n * n;
    if (hv)
        for (int i = tid; i < n; i += T) taug[i] = hv[i] != 0. ? 1. / hv[i] : 0.;
    // The T factors of the reflector panels (dc_apply_reflectors) while the reflectors still sit
    // in the LDS matrix: built at the end, from their global copy, the Gram products waited for L2
    // round trips (11 us of the stage at n = 128).  The work area is free here; hv lives in it and
    // has just been read.
    const bool t_early = hv && !ext_top && !qh_ready;
    if (t_early) {
        __threadfence_block();
        __syncthreads();
        dc_build_T(n, Q.a, taug, taug + n, scratch, Q.ld);
        __syncthreads();
    }
    size_t w_doubles = 0;
    DcWork W = dc_work_layout(scratch, ext_top ? n + 2 : 130, &maxnr_s, &w_doubles);
    // scale to unit max-norm by a power of two
    double am = 0.;
    for (int i = tid; i < n; i += T) am = fmax(am, fmax(fabs(dv[i]), fabs(ev[i])));
    {
        DcTeam all { 1, tid, T, 0, NW, wave, 0 };
        am = dc_team_max(am, W.red, all);
    }
    __syncthreads();
    if (tid == 0) {
        int ex = 0;
        if (am > 0.) frexp(am, &ex);
        scale_s = am > 0. ? ldexp(1., 1 - ex) : 1.;
        // blocks of <= leaf_rows rows.  With the matrix in LDS
        // (16 < n <= 128) the leaves are 8 x 8: a QL leaf is a serial chain of ~2 (s^2 / 2)
        // rotations, the extra level of merges -- 16-pole merges, one wavefront each -- costs a
        // third of what the smaller leaves save (round 3, once the merges had become cheap)
        const int leaf_rows = (!ext_top && n > 16 && n <= 128 && !(dbg & 32768)) ? 8 : DC_LEAF;
        // (a power-of-two number of blocks whose sizes differ by at most one, edges floor(i n / nb):
        // every level pairs all of its blocks.  Until round 3 thread 0 halved the blocks one level
        // at a time through a per-thread array -- scratch memory: most of the 15 us this set-up took)
        int nb = 1;
        while ((n + nb - 1) / nb > leaf_rows && nb < MAXB) nb <<= 1;
        nblk_s = mode == 2 ? 2 : nb;
    }
    __syncthreads();
    if (tid <= nblk_s) bounds[tid] = (int) (((long long) tid * n) / nblk_s);
    __syncthreads();
    const double scale = scale_s;
    const int nblk = nblk_s;
    for (int i = tid; i < n; i += T) {
        dv[i] *= scale;
        ev[i] *= scale;
    }
    // Q becomes the eigenvector matrix of T: clear it (16-byte stores where the rows allow: the
    // global work matrix of n > 128 is 512 KB for this one workgroup)
    if (mode == 2) {
        // (the halves' blocks are in place)
    } else if ((Q.ld & 1) == 0 && (reinterpret_cast<size_t>(Q.a) & 15) == 0) {
        double2 *q2 = reinterpret_cast<double2*>(Q.a);
        for (int q = tid; q < (n * Q.ld) >> 1; q += T) q2[q] = make_double2(0., 0.);
    } else {
        for (int q = tid; q < n * Q.ld; q += T) Q.a[q] = 0.;
    }
    __syncthreads();
    // rank-one tears at the block boundaries (mode 2: the halves were torn before they were solved)
    if (tid >= 1 && tid < nblk && mode != 2) {
        const int bd = bounds[tid];
        const double r = fabs(ev[bd - 1]);
        dv[bd - 1] -= r;
        dv[bd] -= r;
    }
    __syncthreads();

    DC_STAMP(17);
    // ---- leaves: one wavefront each; the merge work area is not in use yet ------------------
    // (two per wavefront, one in each 32-lane half: dc_leaf_ql_pair; diagnostic bit 524288 keeps
    // the one-leaf-at-a-time form)
    if (mode == 2) {
        // (no leaves)
    } else if (!(dbg & 8) && !(dbg & 524288)) {
        for (int pr = wave; 2 * pr < nblk; pr += NW) {
            const int b0 = 2 * pr, b1 = 2 * pr + 1;
            const int a0 = bounds[b0], s0 = bounds[b0 + 1] - a0;
            const int a1 = b1 < nblk ? bounds[b1] : 0, s1 = b1 < nblk ? bounds[b1 + 1] - a1 : 0;
            dc_leaf_ql_pair(Q, a0, s0, a1, s1, dv, ev, dv, lane);
        }
    } else
    if (!(dbg & 8))
    for (int blk = wave; blk < nblk; blk += NW) {
        const int a = bounds[blk], s = bounds[blk + 1] - a;
        dc_leaf_ql(Q, a, s, dv, ev, dv, scratch + (size_t) wave * 272, lane,
                stamps ? stamps + 12 + (blk & 3) : nullptr);
    }
    __syncthreads();

    DC_STAMP(18);
    // ---- merges, bottom-up; all merges of a level run at once, each by its own team --------
    // Block i of level L is the run of leaves [i 2^L, min((i + 1) 2^L, nblk)): merges pair
    // neighbours and an odd block at the end is carried, so the edges of every level are entries
    // of `bounds` -- read where they are needed.  (Kept in per-thread arrays indexed at run time,
    // the edges lived in scratch memory: 5 us of bookkeeping per level.)
    auto edge = [&](int i, int L) { return bounds[min(i << L, nblk)]; };
    int nc = nblk;
    // (diagnostic bits 8192 / 16384: stop after the first / second level, so that the phase clocks
    // of dc_merge_level -- the first team's, every level overwrites them -- show THAT level)
    int levels_done = 0;
    while (nc > 1 && !(dbg & 4) && !((dbg & 8192) && levels_done >= 1) && !((dbg & 16384) && levels_done >= 2)) {
        const int L = levels_done;
        levels_done++;
        const int nm = nc >> 1;                       // merges at this level
        int teams = 1;
        while (teams < nm) teams <<= 1;               // 1, 2, 4, 8
        int wpt = NW / teams > 0 ? NW / teams : 1;   // wavefronts per team
        // lanes per secular root: as many as every team of this level can give its poles
        // (chosen from the WIDEST merge so that all teams run the same code path)
        int m = 0;
        for (int i = 0; i < nm; i++) m = max(m, edge(2 * i + 2, L) - edge(2 * i, L));
        // (round 5) a merge of at most 16 poles never needs more than ONE wavefront (four lanes per
        // root, one column tile): where the workgroup has wavefronts to spare -- the halves of a
        // split decomposition: four merges, eight wavefronts -- a second one per team only turns the
        // team's ~25 phase boundaries from wavefront fences into workgroup barriers
        // (diagnostic bit 65536*64 = 4194304 is taken; bit 8388608 forces barriers anyway)
        if (m <= 16 && !BIG) wpt = 1;
        // a level has at most NW teams at a time: 16 merges (32 leaves, n > 256) take two passes
        const int tcount = min(teams, NW);
        for (int pass = 0; pass * tcount < nm && (BIG || pass == 0); pass++) {
        DcTeam tm;
        tm.id = wave / wpt;
        const int q0 = pass * tcount + tm.id;
        tm.active = q0 < nm;
        tm.wave0 = tm.id * wpt;
        tm.nwaves = wpt;
        tm.twave = wave - tm.wave0;
        tm.ttid = tid - 64 * tm.wave0;
        tm.tthreads = 64 * wpt;
        tm.wave_scope = wpt == 1 && !(dbg & 8388608);      // (diagnostic bit: workgroup barriers)
        if (tid == 0) maxnr_s = 0;
        __syncthreads();
        const int q = tm.active ? q0 : 0;
        // (a block that came out of a merge -- two leaves or more -- has its eigenvalues in order)
        tm.sorted_in = mode == 2 || (L >= 1 && min((2 * q + 2) << L, nblk) - ((2 * q + 1) << L) >= 2);
        const int a = edge(2 * q, L), mid = edge(2 * q + 1, L), b = edge(2 * q + 2, L);
        double *Fg = F + (size_t) a * n;
        double *Tbuf = Tscratch ? Tscratch + (size_t) a * n : nullptr;
        const double rho = ev[mid - 1];
        if (ext_top && nc == 2) {
            // the top merge of a matrix wider than 128 (512 threads): scalar part only
            if (top_part == 2) {
                // (eight loads in flight per thread: copied element by element every LDS store waited
                // for its own round trip to L2 -- 6 us of this kernel at n = 128)
                for (size_t q0 = tid; q0 < w_doubles; q0 += (size_t) 8 * T) {
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = q0 + (size_t) u * T < w_doubles ? Wimg[q0 + (size_t) u * T] : 0.;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        if (q0 + (size_t) u * T < w_doubles) scratch[q0 + (size_t) u * T] = x[u];
                }
                __syncthreads();
                if (tid == 0) maxnr_s = W.cnt[2];
                __syncthreads();
            }
            if (2 * m <= tm.tthreads)
                dc_merge_level<2, false, BIG, WIDE>(Q, tm, a, mid, b, rho, dv, Fg, W, stamps, m, Tbuf, top_part);
            else
                dc_merge_level<1, false, BIG, WIDE>(Q, tm, a, mid, b, rho, dv, Fg, W, stamps, m, Tbuf, top_part);
            if (top_part == 1) {
                __syncthreads();
                for (size_t q = tid; q < w_doubles; q += T) Wimg[q] = scratch[q];
                return;
            }
        } else if (4 * m <= tm.tthreads)
            dc_merge_level<4, true, BIG, WIDE>(Q, tm, a, mid, b, rho, dv, Fg, W, stamps, m, Tbuf);
        else if (2 * m <= tm.tthreads)
            dc_merge_level<2, true, BIG, WIDE>(Q, tm, a, mid, b, rho, dv, Fg, W, stamps, m, Tbuf);
        else
            dc_merge_level<1, true, BIG, WIDE>(Q, tm, a, mid, b, rho, dv, Fg, W, stamps, m, Tbuf);
        if (BIG) __syncthreads();
        }
        nc = (nc + 1) >> 1;
        DC_STAMP(19 + (nc == 1 ? 2 : nc == 2 ? 1 : 0));
    }
    __syncthreads();
    // a single leaf (n <= 16) never went through a merge: sort its eigenpairs here
    const bool single = nblk == 1;
    if (single) {
        if (tid < n) {
            const double v = dv[tid], kv = dc_key(v);
            int r = 0;
            for (int j = 0; j < n; j++) {
                const double u = dc_key(dv[j]);
                r += (u < kv) || (u == kv && j < tid);
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
    if (mode == 1) {
        for (int r = wave; r < n; r += NW)
            for (int cc = lane; cc < n; cc += 64) Bout[(size_t) r * ldb + (single ? W.outpos[cc] : cc)] = Q(r, cc);
        __syncthreads();
        return;
    }
    if (ext_top) {
        // (the products run as separate kernels; with stashed reflectors the second one is
        // cma_eig_wy and needs the panels' T factors: G + 2 n^2 = [tau (n) | T (npanel x 256)])
        if (hv && !t_prebuilt) {
            __syncthreads();
            dc_build_T(n, Qh, taug, taug + n, scratch);
        }
        return;
    }
    if (hv) {
        dc_apply_reflectors<TT>(Q, n, Qh, taug, taug + n, scratch, Bout, ldb,
                single ? W.outpos : nullptr, stamps, t_early);
        DC_STAMP(23);
        return;
    }
    // ---- B = Q_house * Q_T on the matrix cores: a wavefront keeps the Q_house fragments of
    // its row tile in registers and sweeps the column tiles ---------------------------------
    const int ntile = (n + 15) >> 4;
    const int ksteps = (n + 3) >> 2;
    const int fr = lane & 15, fk = lane >> 4;
    for (int rt = wave; rt < ntile; rt += NW) {
        double afrag[DC_KSTEPS];
        const int arow = rt * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < DC_KSTEPS; ks++) {
            const int kk = 4 * ks + fk;
            afrag[ks] = (arow < n && kk < n) ? Qh[(size_t) arow * n + kk] : 0.;
        }
        for (int ct = 0; ct < ntile; ct++) {
            dc_d4 acc = { 0., 0., 0., 0. };
            const int col = ct * 16 + fr;
#pragma unroll
            for (int ks = 0; ks < DC_KSTEPS; ks++) {
                if (ks < ksteps) {
                    const int kk = 4 * ks + fk;
                    const double bv = (kk < n && col < n) ? Q(kk, col) : 0.;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[ks], bv, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = rt * 16 + (lane >> 4) + 4 * r;
                if (row < n && col < n)
                    Bout[(size_t) row * ldb + (single ? W.outpos[col] : col)] = acc[r];
            }
        }
    }
    __syncthreads();
    DC_STAMP(23);
#undef DC_STAMP
}

} // namespace bbo
