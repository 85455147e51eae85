// bbo_eig_ql.hpp -- the implicit-shift QL recurrence of the reference (tql2, cmaes.cpp:383-456)
// split into a producer (scalar recurrence on (d, e), records Givens pairs) and a consumer
// (applies recorded pairs to one row of the eigenvector matrix).  Used by cma_eigen's QL path
// (bbo_eig.hpp) and by the leaves of the divide-and-conquer path (bbo_eig_dc.hpp).
#pragma once

#include <hip/hip_runtime.h>

namespace bbo {

constexpr int EIG_MAXSEQ = 64;

struct EigMat {
    double *a;
    int ld;
    __device__ double& operator()(int i, int j) const { return a[(size_t) i * ld + j]; }
};

// wavefront sum on the cross-lane data path (DPP row rotations, then one readlane per
// 16-lane row) instead of LDS permutes; every lane gets the same total, and the fixed
// order makes it identical in every wavefront that sums the same values
template<int CTRL>
__device__ inline double eig_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// partner exchange inside groups of four lanes (DPP quad_perm: a plain vector move, where
// __shfl_xor goes through the LDS crossbar -- ds_bpermute, an address computation and a wait on
// the critical path of every Householder step and every secular iteration)
__device__ inline double eig_quad_xor1(double v) { return eig_dpp<0xB1>(v); }   // quad_perm:[1,0,3,2]
__device__ inline double eig_quad_xor2(double v) { return eig_dpp<0x4E>(v); }   // quad_perm:[2,3,0,1]
__device__ inline double eig_quad_sum(double v)     // all four lanes of a quad get the same sum
{
    v += eig_quad_xor1(v);
    v += eig_quad_xor2(v);
    return v;
}

__device__ inline double eig_readlane(double v, int l)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

__device__ inline double eig_wave_sum(double v)
{
    v += eig_dpp<0x128>(v);   // row_ror:8
    v += eig_dpp<0x124>(v);   // row_ror:4
    v += eig_dpp<0x122>(v);   // row_ror:2
    v += eig_dpp<0x121>(v);   // row_ror:1
    return ((eig_readlane(v, 0) + eig_readlane(v, 16)) + eig_readlane(v, 32))
            + eig_readlane(v, 48);
}

// The same sum as a butterfly that ends with the total in every lane: two half-exchanges
// (v_permlane32_swap / v_permlane16_swap of a value with its own copy: each lane then holds its
// own and its partner's), then the four rotations inside a 16-lane row.  Six dependent steps and
// no detour through scalar registers (eig_wave_sum: four rotations, four readlanes, three adds);
// another order of additions, so another rounding -- but the same in every lane and wavefront.
__device__ inline double eig_wave_sum_bf(double v)
{
    {
        unsigned alo = __double2loint(v), ahi = __double2hiint(v);
        auto r0 = __builtin_amdgcn_permlane32_swap(alo, alo, false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(ahi, ahi, false, false);
        v = __hiloint2double((int) r1[0], (int) r0[0]) + __hiloint2double((int) r1[1], (int) r0[1]);
    }
    {
        unsigned alo = __double2loint(v), ahi = __double2hiint(v);
        auto r0 = __builtin_amdgcn_permlane16_swap(alo, alo, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(ahi, ahi, false, false);
        v = __hiloint2double((int) r1[0], (int) r0[0]) + __hiloint2double((int) r1[1], (int) r0[1]);
    }
    v += eig_dpp<0x128>(v);   // row_ror:8
    v += eig_dpp<0x124>(v);   // row_ror:4
    v += eig_dpp<0x122>(v);   // row_ror:2
    v += eig_dpp<0x121>(v);   // row_ror:1
    return v;
}

// Lanes of one wavefront talk through LDS here (lane 0 walks the recurrence, the others shift
// the diagonal and search for the split).  The hardware keeps a wavefront's LDS operations in
// order, but the COMPILER reasons per thread: on the path that skips the `if (lane == 0)` block
// it may hoist a later load above the block and so read what lane 0 is about to overwrite.
// Every hand-over point therefore carries a wavefront-scope fence.
__device__ inline void ql_wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// reciprocal and reciprocal root from the hardware estimates + Newton corrections (full fp64 to
// a rounding error).  The sweep's shift and its closing quotient sit on the serial path of the
// recurrence -- four IEEE divisions and a hypot per sweep were a third of a leaf's time
__device__ inline double ql_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
}
__device__ inline double ql_hypot1(double p)       // sqrt(p^2 + 1)
{
    const double t = fma(p, p, 1.);
    double y = __builtin_amdgcn_rsq(t);
    const double err = fma(-t * y, y, 1.);
    y = fma(y * err, fma(err, 0.375, 0.5), y);
    double r = t * y;
    return fma(fma(-r, r, t), 0.5 * y, r);         // one more step on the root itself
}

// state of the QL recurrence, uniform across the producer wavefront
struct QlState {
    int l, m, need_m, done;
    double f, tst1;
    int total;            // sweeps so far, over all calls (bounded: see ql_produce)
};

// Producer: advances the QL recurrence on (dv, ev), recording up to `rc` Givens pairs of
// whole sweeps into `rot` (pair for column i of sweep q at rot[off_q + i - l_q]) and their
// descriptors (l, m, off) into `desc`.  Executed by all 64 lanes of wavefront 0; lane 0
// walks the recurrence, the other lanes help with the O(n) shift and the split search.
__device__ inline int ql_produce(QlState &st, int n, double *dv, double *ev, double2 *rot,
        int *desc, int rc, int lane)
{
    const double eps = 0x1.0p-52;
    int count = 0, ns = 0;
    while (!st.done) {
        ql_wave_fence();
        if (st.need_m) {
            const double dl = dv[st.l], el = ev[st.l];
            st.tst1 = fmax(st.tst1, fabs(dl) + fabs(el));
            const double thr = eps * st.tst1;
            int m = n;
            for (int base = st.l; base < n; base += 64) {
                const int idx = base + lane;
                const bool ok = idx < n && fabs(ev[idx]) <= thr;
                const unsigned long long mask = __ballot(ok);
                if (mask) {
                    m = base + __builtin_ctzll(mask);
                    break;
                }
            }
            st.m = m;
            st.need_m = 0;
            if (m >= n) {   // unreachable for finite input: e[n-1] == 0
                st.done = 1;
                break;
            }
            if (m == st.l) {
                if (lane == 0) {
                    dv[st.l] = dl + st.f;
                    ev[st.l] = 0.;
                }
                st.l++;
                st.need_m = 1;
                if (st.l >= n) st.done = 1;
                continue;
            }
        }
        const int l = st.l, m = st.m, len = m - l;
        if (count + len > rc || ns >= EIG_MAXSEQ) break;
        // (the reference's tql2 has no sweep limit; here a matrix that never converges must not
        // hang the GPU: 30 sweeps per eigenvalue, far beyond anything finite input needs)
        if (++st.total > 30 * n) {
            st.done = 1;
            break;
        }
        const double thr = eps * st.tst1;

        // implicit shift (cmaes.cpp:405-417)
        const double g0 = dv[l], d1 = dv[l + 1], el = ev[l];
        const double p0 = (d1 - g0) * ql_rcp(2. * el);
        double r0 = ql_hypot1(p0);
        r0 = p0 >= 0. ? r0 : -r0;
        const double dl_new = el * ql_rcp(p0 + r0);
        const double dl1 = el * (p0 + r0);
        const double h0 = g0 - dl_new;
        ql_wave_fence();
        for (int i = l + 2 + lane; i < n; i += 64) dv[i] -= h0;
        st.f += h0;
        ql_wave_fence();

        if (lane == 0) {
            dv[l] = dl_new;
            dv[l + 1] = dl1;
            // implicit QL sweep (cmaes.cpp:419-449)
            double pp = dv[m];
            double cth = 1., c2 = 1., c3 = 1., s = 0., s2 = 0.;
            const double el1 = ev[l + 1];
            double ei = ev[m - 1], di = dv[m - 1];
            double2 *out = rot + count;
            for (int i = m - 1; i >= l; i--) {
                // next column's (e, d): always a legal address (the vectors carry a front pad)
                const double ein = ev[i - 1], din = dv[i - 1];
                c3 = c2;
                c2 = cth;
                s2 = s;
                const double g = cth * ei;
                const double h = cth * pp;
                const double t = fma(pp, pp, ei * ei);
                // 1/sqrt(t): hardware estimate + one third-order correction (full fp64)
                double y = __builtin_amdgcn_rsq(t);
                const double err = fma(-t * y, y, 1.);
                y = fma(y * err, fma(err, 0.375, 0.5), y);
                const double r = t * y;             // = hypot(pp, ei) to rounding
                ev[i + 1] = s * r;
                s = ei * y;
                cth = pp * y;
                pp = fma(cth, di, -(s * g));
                dv[i + 1] = h + s * fma(cth, g, s * di);
                out[i - l] = make_double2(cth, s);
                ei = ein;
                di = din;
            }
            pp = -s * s2 * c3 * el1 * ev[l] * ql_rcp(dl1);
            ev[l] = s * pp;
            dv[l] = cth * pp;
        }
        ql_wave_fence();
        desc[3 * ns + 0] = l;
        desc[3 * ns + 1] = m;
        desc[3 * ns + 2] = count;
        count += len;
        ns++;
        const double el_new = ev[l];
        if (!(fabs(el_new) > thr)) {
            if (lane == 0) {
                dv[l] += st.f;
                ev[l] = 0.;
            }
            st.l++;
            st.need_m = 1;
            if (st.l >= n) st.done = 1;
        }
    }
    return ns;
}

// The same recurrence for a block of at most 64 rows (the leaves of the divide and conquer, the
// whole matrix for n <= 16) with (d, e) in REGISTERS -- lane i holds d[i] and e[i] -- and every
// lane walking the recurrence on uniform values: an element is fetched by v_readlane and put
// back by a one-lane select, where the LDS form paid an LDS round trip per hand-over (a sweep's
// set-up alone was six of them, one after the other) and a wavefront fence at each.  The same
// operations in the same order: the same bits.  Only the Givens pairs and the sweep descriptors go
// through LDS (the consumer reads them).
__device__ inline double ql_lane(double v, int idx)      // v of lane idx (idx uniform)
{
    const int k = __builtin_amdgcn_readfirstlane(idx);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
    return __hiloint2double(hi, lo);
}

// FUSED: every rotation is applied to the eigenvector block at once -- lane k < n owns row k of
// `blk` (the consumer's streaming form: one LDS read and one write per rotation, the carried
// element in a register) -- in the latency shadow of the recurrence, which issues a dozen
// dependent instructions per rotation and leaves the pipe idle in between; no pairs are recorded,
// no rounds.  Same operations per row in the same order as ql_apply_row.
template<bool FUSED>
__device__ inline int ql_produce_reg(QlState &st, int n, double &d, double &e, double2 *rot,
        int *desc, int rc, int lane, const EigMat &blk)
{
    const double eps = 0x1.0p-52;
    int count = 0, ns = 0;
    double *zrow = blk.a + (size_t) (lane < n ? lane : 0) * blk.ld;
    while (!st.done) {
        if (st.need_m) {
            const double dl = ql_lane(d, st.l), el = ql_lane(e, st.l);
            st.tst1 = fmax(st.tst1, fabs(dl) + fabs(el));
            const double thr = eps * st.tst1;
            const bool ok = lane >= st.l && lane < n && fabs(e) <= thr;
            const unsigned long long mask = __ballot(ok);
            const int m = mask ? (int) __builtin_ctzll(mask) : n;
            st.m = m;
            st.need_m = 0;
            if (m >= n) {   // unreachable for finite input: e[n-1] == 0
                st.done = 1;
                break;
            }
            if (m == st.l) {
                if (lane == st.l) {
                    d = dl + st.f;
                    e = 0.;
                }
                st.l++;
                st.need_m = 1;
                if (st.l >= n) st.done = 1;
                continue;
            }
        }
        const int l = st.l, m = st.m, len = m - l;
        if (!FUSED && (count + len > rc || ns >= EIG_MAXSEQ)) break;
        // Sweep bound of the fused form (one call does the whole block): the reference's tql2
        // has no limit, but here a block that never converges would hang the GPU instead of
        // returning a bad basis.  30 sweeps per eigenvalue is far beyond anything finite input
        // needs (2-3 observed); the uniform counter ends the loop and the caller's repair /
        // eigen_done path deals with what is left.
        if (FUSED && ns >= 30 * n) {
            st.done = 1;
            break;
        }
        const double thr = eps * st.tst1;

        // implicit shift (cmaes.cpp:405-417)
        const double g0 = ql_lane(d, l), d1 = ql_lane(d, l + 1), el = ql_lane(e, l);
        const double p0 = (d1 - g0) * ql_rcp(2. * el);
        double r0 = ql_hypot1(p0);
        r0 = p0 >= 0. ? r0 : -r0;
        const double dl_new = el * ql_rcp(p0 + r0);
        const double dl1 = el * (p0 + r0);
        const double h0 = g0 - dl_new;
        if (lane >= l + 2 && lane < n) d -= h0;
        st.f += h0;
        if (lane == l) d = dl_new;
        if (lane == l + 1) d = dl1;

        // implicit QL sweep (cmaes.cpp:419-449)
        double pp = ql_lane(d, m);
        double cth = 1., c2 = 1., c3 = 1., sn = 0., s2 = 0.;
        const double el1 = ql_lane(e, l + 1);
        double ei = ql_lane(e, m - 1), di = ql_lane(d, m - 1);
        double2 *out = rot + count;
        double hcur = FUSED ? zrow[m] : 0., zx = FUSED ? zrow[m - 1] : 0.;
        for (int i = m - 1; i >= l; i--) {
            const int ip = i > 0 ? i - 1 : 0;              // (the value for i = l is not used)
            const double ein = ql_lane(e, ip), din = ql_lane(d, ip);
            const double zn = FUSED ? zrow[ip] : 0.;       // next column's element of this lane's row
            c3 = c2;
            c2 = cth;
            s2 = sn;
            const double g = cth * ei;
            const double h = cth * pp;
            const double t = fma(pp, pp, ei * ei);
            // 1/sqrt(t): hardware estimate + one third-order correction (full fp64)
            double y = __builtin_amdgcn_rsq(t);
            const double err = fma(-t * y, y, 1.);
            y = fma(y * err, fma(err, 0.375, 0.5), y);
            const double r = t * y;             // = hypot(pp, ei) to rounding
            const double e_up = sn * r;
            sn = ei * y;
            cth = pp * y;
            pp = fma(cth, di, -(sn * g));
            const double d_up = h + sn * fma(cth, g, sn * di);
            if (lane == i + 1) {
                e = e_up;
                d = d_up;
            }
            if (FUSED) {
                if (lane < n) zrow[i + 1] = sn * zx + cth * hcur;
                hcur = cth * zx - sn * hcur;
                zx = zn;
            } else if (lane == 0) {
                out[i - l] = make_double2(cth, sn);
            }
            ei = ein;
            di = din;
        }
        if (FUSED && lane < n) zrow[l] = hcur;
        pp = -sn * s2 * c3 * el1 * ql_lane(e, l) * ql_rcp(dl1);
        const double el_new = sn * pp;
        if (lane == l) {
            e = el_new;
            d = cth * pp;
        }
        if (!FUSED && lane == 0) {
            desc[3 * ns + 0] = l;
            desc[3 * ns + 1] = m;
            desc[3 * ns + 2] = count;
        }
        count += len;
        ns++;
        if (!(fabs(el_new) > thr)) {
            if (lane == l) {
                d += st.f;
                e = 0.;
            }
            st.l++;
            st.need_m = 1;
            if (st.l >= n) st.done = 1;
        }
    }
    return ns;
}

// ---- two leaves per wavefront, one in each 32-lane half ---------------------------------------------
// A leaf of the divide and conquer (<= 32 rows; 8 or 16 in practice) is one chain of Givens
// rotations, ~40 dependent vector instructions each, walked by all 64 lanes on uniform values while
// at most 16 of them hold data: with 16 leaves on 8 wavefronts the kernel solved them two IN TURN
// (61 us of the 400 at n = 128; 220 of 1470 at n = 256).  Here lanes 0-31 walk leaf A and lanes
// 32-63 leaf B in the SAME instruction stream: the state (l, m, f, the sweep's running values) is
// per half, an element of (d, e) is fetched by one v_readlane per half and a select, and where the
// two recurrences part (different sweep lengths, one finished) the hardware masks lanes as for any
// divergent loop.  Every leaf executes exactly the operations of ql_produce_reg<true> in the same
// order: the same bits.  (Tried in round 3 as two leaves interleaved BY HAND in one uniform stream:
// twice the instructions per rotation, slower.)
// v of lane (32 * half + idx): idx is per half (the same in all lanes of a half), 0 <= idx < 32
__device__ inline double ql_half_lane(double v, int idx, int lane)
{
    const int ia = __builtin_amdgcn_readlane(idx, 0) & 31;
    const int ib = 32 + (__builtin_amdgcn_readlane(idx, 32) & 31);
    const int lo_a = __builtin_amdgcn_readlane(__double2loint(v), ia);
    const int hi_a = __builtin_amdgcn_readlane(__double2hiint(v), ia);
    const int lo_b = __builtin_amdgcn_readlane(__double2loint(v), ib);
    const int hi_b = __builtin_amdgcn_readlane(__double2hiint(v), ib);
    return lane < 32 ? __hiloint2double(hi_a, lo_a) : __hiloint2double(hi_b, lo_b);
}

// hl = lane & 31 holds d[hl], e[hl] of its half's block of n rows (n per half, 0: no block);
// zrow: row hl of the half's eigenvector block (LDS), identity on entry.  Returns the sweeps.
__device__ inline int ql_leaf_pair(int n, double &d, double &e, double *zrow, int lane)
{
    const double eps = 0x1.0p-52;
    const int hl = lane & 31;
    int l = 0, m = 0, ns = 0;
    bool need_m = true, done = n <= 0;
    double fsum = 0., tst1 = 0.;
    while (!done) {
        if (need_m) {
            const double dl = ql_half_lane(d, l, lane), el = ql_half_lane(e, l, lane);
            tst1 = fmax(tst1, fabs(dl) + fabs(el));
            const double thr = eps * tst1;
            const bool ok = hl >= l && hl < n && fabs(e) <= thr;
            const unsigned long long mask = __ballot(ok);
            const unsigned mh = lane < 32 ? (unsigned) mask : (unsigned) (mask >> 32);
            m = mh ? (int) __builtin_ctz(mh) : n;
            need_m = false;
            if (m >= n) {   // unreachable for finite input: e[n-1] == 0
                done = true;
                continue;
            }
            if (m == l) {
                if (hl == l) {
                    d = dl + fsum;
                    e = 0.;
                }
                l++;
                need_m = true;
                if (l >= n) done = true;
                continue;
            }
        }
        // (sweep bound: see ql_produce_reg)
        if (ns >= 30 * n) {
            done = true;
            continue;
        }
        const double thr = eps * tst1;
        // implicit shift (cmaes.cpp:405-417)
        const double g0 = ql_half_lane(d, l, lane), d1 = ql_half_lane(d, l + 1, lane),
                el = ql_half_lane(e, l, lane);
        const double p0 = (d1 - g0) * ql_rcp(2. * el);
        double r0 = ql_hypot1(p0);
        r0 = p0 >= 0. ? r0 : -r0;
        const double dl_new = el * ql_rcp(p0 + r0);
        const double dl1 = el * (p0 + r0);
        const double h0 = g0 - dl_new;
        if (hl >= l + 2 && hl < n) d -= h0;
        fsum += h0;
        if (hl == l) d = dl_new;
        if (hl == l + 1) d = dl1;

        // implicit QL sweep (cmaes.cpp:419-449)
        double pp = ql_half_lane(d, m, lane);
        double cth = 1., c2 = 1., c3 = 1., sn = 0., s2 = 0.;
        const double el1 = ql_half_lane(e, l + 1, lane);
        double ei = ql_half_lane(e, m - 1, lane), di = ql_half_lane(d, m - 1, lane);
        double hcur = zrow[m], zx = zrow[m - 1];
        for (int i = m - 1; i >= l; i--) {
            const int ip = i > 0 ? i - 1 : 0;              // (the value for i = l is not used)
            const double ein = ql_half_lane(e, ip, lane), din = ql_half_lane(d, ip, lane);
            const double zn = zrow[ip];
            c3 = c2;
            c2 = cth;
            s2 = sn;
            const double g = cth * ei;
            const double h = cth * pp;
            const double t = fma(pp, pp, ei * ei);
            double y = __builtin_amdgcn_rsq(t);
            const double err = fma(-t * y, y, 1.);
            y = fma(y * err, fma(err, 0.375, 0.5), y);
            const double r = t * y;             // = hypot(pp, ei) to rounding
            const double e_up = sn * r;
            sn = ei * y;
            cth = pp * y;
            pp = fma(cth, di, -(sn * g));
            const double d_up = h + sn * fma(cth, g, sn * di);
            if (hl == i + 1) {
                e = e_up;
                d = d_up;
            }
            if (hl < n) zrow[i + 1] = sn * zx + cth * hcur;
            hcur = cth * zx - sn * hcur;
            zx = zn;
            ei = ein;
            di = din;
        }
        if (hl < n) zrow[l] = hcur;
        pp = -sn * s2 * c3 * el1 * ql_half_lane(e, l, lane) * ql_rcp(dl1);
        const double el_new = sn * pp;
        if (hl == l) {
            e = el_new;
            d = cth * pp;
        }
        ns++;
        if (!(fabs(el_new) > thr)) {
            if (hl == l) {
                d += fsum;
                e = 0.;
            }
            l++;
            need_m = true;
            if (l >= n) done = true;
        }
    }
    return ns;
}

// Consumer: applies the recorded sweeps to row k of the eigenvector matrix
// (the inner k-loop of cmaes.cpp:438-443, one lane per k)
__device__ inline void ql_apply_row(const EigMat &A, int k, const double2 *rot,
        const int *desc, int ns)
{
    for (int q = 0; q < ns; q++) {
        const int l = desc[3 * q], m = desc[3 * q + 1];
        const int cb = desc[3 * q + 2] - l;               // rot[cb + i] = pair of column i
        double hcur = A(k, m);
        int i = m - 1;
        for (; i - 3 >= l; i -= 4) {
            const double x0 = A(k, i), x1 = A(k, i - 1), x2 = A(k, i - 2), x3 = A(k, i - 3);
            const double2 r0 = rot[cb + i], r1 = rot[cb + i - 1], r2 = rot[cb + i - 2],
                    r3 = rot[cb + i - 3];
            A(k, i + 1) = r0.y * x0 + r0.x * hcur;
            hcur = r0.x * x0 - r0.y * hcur;
            A(k, i) = r1.y * x1 + r1.x * hcur;
            hcur = r1.x * x1 - r1.y * hcur;
            A(k, i - 1) = r2.y * x2 + r2.x * hcur;
            hcur = r2.x * x2 - r2.y * hcur;
            A(k, i - 2) = r3.y * x3 + r3.x * hcur;
            hcur = r3.x * x3 - r3.y * hcur;
        }
        for (; i >= l; i--) {
            const double x = A(k, i);
            const double2 r = rot[cb + i];
            A(k, i + 1) = r.y * x + r.x * hcur;
            hcur = r.x * x - r.y * hcur;
        }
        A(k, l) = hcur;
    }
}


} // namespace bbo
