// bbo_cso_kernels.hpp -- one CSO generation as gfx950 kernels.
//
//   kernel            reference lines (cso.cpp)                       bytes per particle
//   cso_init          :84-97 uniform swarm, v = 0                      16n written
//   cso_ring_mean     :117-123 ring neighbourhood mean                 24n read + 8n written
//   cso_colsum/mean   :124-131 swarm mean (also the winners' mean)     8n read (the swarm mean
//                     for n <= 512: cso_wgsum over the sums cso_compete leaves per workgroup)
//   cso_shuffle       :136 Random::shuffle (keyed Feistel bijection)     8
//   cso_groups        :137-143 sort inside every group                 (f only)
//   cso_compete       :219-276 velocity / position / evaluate of the losers: parent x, own x,
//                     own v, one mean row read, x and v written = 48n + 8 per loser, HBM-bound
//   cso_finish        :150-156 incumbent, :177-194 stop test           8 per particle
#pragma once

#include "bbo_cso.hpp"
#include "bbo_objectives.hpp"
#include "bbo_rng.hpp"

namespace bbo {

#define CSO_INF (__builtin_huge_val())

__device__ inline bool cso_frozen(const CsoConst &c, const CsoScal *sc)
{
    return c.honor_stop && sc->stop != 0;
}

// orders a wavefront's LDS accesses among its own lanes (hardware keeps them in order; the fence
// stops the compiler from moving loads across the point)
__device__ inline void cso_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template<int G>
__device__ inline double cso_group_sum(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

// grid (ceil(np/16), P), 256 threads, LDS 16 * ld doubles
__global__ __launch_bounds__(256) void cso_init(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * (blockDim.x >> 4) + r, ld = c.ld;
    double *row = lds + r * ld;
    const size_t base = ((size_t) p * c.np + i) * ld;
    double ssq = 0.;
    if (i < c.np) {
        for (int pj = g; pj < ld / 2; pj += 16) {
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) pj, 0,
                    stream_word(STREAM_INIT, (uint32_t) p));
            const int j = 2 * pj;
            double2 v = make_double2(0., 0.);
            if (j < c.n) v.x = u01(w.x, w.y) * (d.upper[j] - d.lower[j]) + d.lower[j];
            if (j + 1 < c.n) v.y = u01(w.z, w.w) * (d.upper[j + 1] - d.lower[j + 1]) + d.lower[j + 1];
            *reinterpret_cast<double2*>(&row[j]) = v;
            *reinterpret_cast<double2*>(&d.X[base + j]) = v;
            *reinterpret_cast<double2*>(&d.V[base + j]) = make_double2(0., 0.);
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    __syncthreads();
    ssq = cso_group_sum<16>(ssq);
    double f = CSO_INF;
    if (c.obj >= 0) {
        f = eval_row_group<16>(c.obj, c.n, row, d.aux, g);
        if (f != f) f = CSO_INF;
    }
    if (g == 0 && i < c.np) {
        d.f[(size_t) p * c.np + i] = f;
        d.radius[(size_t) p * c.np + i] = sqrt(ssq);
        d.occ[(size_t) p * c.np + i] = i;
    }
}

// ring topology: mean of the particle and of the occupants of the two slots next to its
// birth slot (the reference's _left / _right are pointers to SLOTS).  grid (ceil(np/16), P)
__global__ __launch_bounds__(256) void cso_ring_mean(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y;
    if (cso_frozen(c, d.scal + p)) return;
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * 16 + r, ld = c.ld, np = c.np;
    if (i >= np) return;
    const size_t pb = (size_t) p * np;
    const double *X = d.X + pb * ld;
    const double *xl = X + (size_t) d.occ[pb + (i - 1 + np) % np] * ld;
    const double *xr = X + (size_t) d.occ[pb + (i + 1) % np] * ld;
    const double *xi = X + (size_t) i * ld;
    double *pm = d.PM + (pb + i) * ld;
    for (int pj = g; pj < ld / 2; pj += 16) {
        const double2 a = *reinterpret_cast<const double2*>(&xl[2 * pj]);
        const double2 b = *reinterpret_cast<const double2*>(&xi[2 * pj]);
        const double2 e = *reinterpret_cast<const double2*>(&xr[2 * pj]);
        double2 m;
        m.x = (a.x + b.x + e.x) / 3.;
        m.y = (a.y + b.y + e.y) / 3.;
        *reinterpret_cast<double2*>(&pm[2 * pj]) = m;
    }
}

// column sums of the rows listed by `occ` with stride `step` (step = 1: all particles in
// slot order; step = pc: the winners).  grid (parts, P), 256 threads
__global__ __launch_bounds__(256) void cso_colsum(CsoDev d, CsoConst c, int step, int count)
{
    const int p = blockIdx.y, part = blockIdx.x;
    if (cso_frozen(c, d.scal + p)) return;
    const int per = (count + c.parts - 1) / c.parts;
    const int q0 = part * per, q1 = min(count, q0 + per);
    const size_t pb = (size_t) p * c.np;
    // two columns per thread (16-byte loads; ld is a multiple of 16), every column summed over the
    // rows in the same order as before
    for (int j = 2 * threadIdx.x; j < c.ld; j += 512) {
        double s0 = 0., s1 = 0.;
        int q = q0;
        for (; q + 8 <= q1; q += 8) {          // eight independent row reads in flight
            double2 x[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                x[u] = *reinterpret_cast<const double2*>(
                        &d.X[(pb + d.occ[pb + (size_t) (q + u) * step]) * c.ld + j]);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                s0 += x[u].x;
                s1 += x[u].y;
            }
        }
        for (; q < q1; q++) {
            const double2 x = *reinterpret_cast<const double2*>(
                    &d.X[(pb + d.occ[pb + (size_t) q * step]) * c.ld + j]);
            s0 += x.x;
            s1 += x.y;
        }
        *reinterpret_cast<double2*>(&d.colpart[((size_t) p * c.parts + part) * c.ld + j]) =
                make_double2(s0, s1);
    }
}

__global__ __launch_bounds__(256) void cso_mean(CsoDev d, CsoConst c, int winners, int count)
{
    const int p = blockIdx.x;
    if (cso_frozen(c, d.scal + p)) return;
    double *dst = (winners ? d.meanw : d.mean) + (size_t) p * c.ld;
    for (int j = threadIdx.x; j < c.ld; j += 256) {
        double s = 0.;
        int q = 0;
        // (four workgroups in all: what this kernel costs is load latency, so many loads at once)
        for (; q + 32 <= c.parts; q += 32) {
            double x[32];
#pragma unroll
            for (int u = 0; u < 32; u++) x[u] = d.colpart[((size_t) p * c.parts + q + u) * c.ld + j];
#pragma unroll
            for (int u = 0; u < 32; u++) s += x[u];
        }
        for (; q + 8 <= c.parts; q += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = d.colpart[((size_t) p * c.parts + q + u) * c.ld + j];
#pragma unroll
            for (int u = 0; u < 8; u++) s += x[u];
        }
        for (; q < c.parts; q++) s += d.colpart[((size_t) p * c.parts + q) * c.ld + j];
        dst[j] = s / count;
    }
}

// the shuffle (cso.cpp:136 Random::shuffle): a keyed bijection of the slots instead of a sort --
// new slot s takes the occupant of slot perm(s), perm = 4-round Feistel network on 2 kb bits
// with one Philox word per round, cycle-walked back into [0, np).  O(1) per slot, no
// communication; oracle twin: Cso::feistel_perm.  grid (ceil(np/256), P), 256 threads
__global__ __launch_bounds__(256) void cso_shuffle(CsoDev d, CsoConst c, int kb)
{
    const int p = blockIdx.y;
    const CsoScal *sc = d.scal + p;
    if (cso_frozen(c, sc)) return;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= c.np) return;
    const size_t pb = (size_t) p * c.np;
    const uint32_t src = cso_perm((uint32_t) s, kb, (uint32_t) c.np, c.seed, (uint32_t) sc->gen,
            stream_word(STREAM_PSO_CTRL, (uint32_t) p));
    d.occ2[pb + s] = d.occ[pb + src];
}

// sort inside every group of pc consecutive slots by fitness (stable: ties keep the shuffled
// order, like the insertion sort std::sort runs on such short ranges).  One thread per group.
__global__ __launch_bounds__(256) void cso_groups(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y;
    if (cso_frozen(c, d.scal + p)) return;
    const int gI = blockIdx.x * 256 + threadIdx.x;
    if (gI >= c.ngroup) return;
    const size_t pb = (size_t) p * c.np;
    const int *src = d.occ2 + pb + (size_t) gI * c.pc;
    int *dst = d.occ + pb + (size_t) gI * c.pc;
    const double *f = d.f + pb;
    for (int k = 0; k < c.pc; k++) {
        const int row = src[k];
        const double fr = f[row];
        int q = k;
        while (q > 0 && fr < f[dst[q - 1]]) {
            dst[q] = dst[q - 1];
            q--;
        }
        dst[q] = row;
    }
}

// the losers of 16 groups per workgroup, 16 lanes per group, worst first (each learns from the
// next better particle of its group BEFORE that one moves, cso.cpp:222-228).
// grid (ceil(ngroup/16), P), 256 threads, LDS 16 * ld doubles
// the workgroup's team rows (LDS, `teams` x ld) summed in team order -> wgpart[p][blockIdx.x]
__device__ inline void cso_team_rows_to_wgpart(const CsoDev &d, const CsoConst &c, int p,
        const double *lds, int teams)
{
    const int ld = c.ld;
    double *dst = d.wgpart + ((size_t) p * c.nwg + blockIdx.x) * ld;
    for (int j = threadIdx.x; j < ld; j += blockDim.x) {
        double sum = lds[j];
        for (int rr = 1; rr < teams; rr++) sum += lds[rr * ld + j];
        dst[j] = sum;
    }
}

// G lanes per group (16, 32 or 64: one or several groups per wavefront), blockDim.x / G groups
// per workgroup.  FUSE (the host picks G so that a lane owns at most four column pairs, i.e.
// ld <= 8 G): the kernel also leaves the column sums of ALL rows of its groups -- the losers'
// new positions and the winners -- in wgpart[p][workgroup][ld], from which the next generation
// takes the swarm mean (cso_wgsum + cso_mean) instead of reading the whole swarm again (a quarter
// of the generation's HBM traffic).  Per column: a team adds its group's rows from the worst
// loser up to the winner, the workgroup its teams in order.
template<int G, bool FUSE>
__global__ __launch_bounds__(256) void cso_compete(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y;
    const CsoScal *sc = d.scal + p;
    if (cso_frozen(c, sc)) return;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid / G, g = tid % G;
    const int gI = blockIdx.x * (blockDim.x / G) + r, ld = c.ld, n = c.n, gen = sc->gen;
    const bool live = gI < c.ngroup;
    double *trial = lds + r * ld;
    // The box in LDS, once per workgroup, behind the teams' rows: read from global memory under
    // `if (j < n)` inside the update loop, each column pair waited for its own loads -- and, the
    // memory counter being in order, for the STORES of the pair before it (round 3, from the ISA).
    double *lob = lds + (size_t) (blockDim.x / G) * ld, *upb = lob + ld;
    for (int j = tid; j < ld; j += blockDim.x) {
        lob[j] = d.lower[j];
        upb[j] = d.upper[j];
    }
    __syncthreads();
    const size_t pb = (size_t) p * c.np;
    const int *occ = d.occ + pb + (size_t) (live ? gI : 0) * c.pc;
    const double *gmean = d.mean + (size_t) p * ld, *wmean = d.meanw + (size_t) p * ld;
    double2 csum[4];
#pragma unroll
    for (int u = 0; u < 4; u++) csum[u] = make_double2(0., 0.);
    for (int k = c.pc - 1; k >= 1; k--) {
        const int slot = gI * c.pc + k;
        const int row = occ[k], prow = occ[k - 1];
        double *x = d.X + (pb + row) * ld, *v = d.V + (pb + row) * ld;
        const double *xp = d.X + (pb + prow) * ld;
        const double *xm = k == 1 ? (c.ring ? d.PM + (pb + row) * ld : gmean) : wmean;
        double ssq = 0.;
        if (live) {
            const u32x4 wp = philox4x32_10(c.seed, (uint32_t) slot, 1, (uint32_t) gen,
                    stream_word(STREAM_PSO_CTRL, (uint32_t) p));
            const double phi = u01(wp.x, wp.y) * (c.phih - c.phil) + c.phil;
            const uint32_t swr = stream_word(STREAM_PSO_R, (uint32_t) p);
            // four column pairs per lane at a time, their sixteen row loads issued first
            const int npair = ld >> 1;
            for (int pj0 = g; pj0 < npair; pj0 += 4 * G) {
                double2 xi4[4], vi4[4], pa4[4], me4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int pj = pj0 + G * u;
                    const int j = pj < npair ? 2 * pj : 0;
                    xi4[u] = *reinterpret_cast<const double2*>(&x[j]);
                    vi4[u] = *reinterpret_cast<const double2*>(&v[j]);
                    pa4[u] = *reinterpret_cast<const double2*>(&xp[j]);
                    me4[u] = *reinterpret_cast<const double2*>(&xm[j]);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int pj = pj0 + G * u;
                    if (pj >= npair) continue;
                    const int j = 2 * pj;
                    const double2 xi = xi4[u], vi = vi4[u], pa = pa4[u], me = me4[u];
                    const u32x4 wa = philox4x32_10(c.seed, (uint32_t) slot, (uint32_t) (3 * pj),
                            (uint32_t) gen, swr);
                    const u32x4 wb = philox4x32_10(c.seed, (uint32_t) slot, (uint32_t) (3 * pj + 1),
                            (uint32_t) gen, swr);
                    const u32x4 wc = philox4x32_10(c.seed, (uint32_t) slot, (uint32_t) (3 * pj + 2),
                            (uint32_t) gen, swr);
                    double2 xn = make_double2(0., 0.), vn = make_double2(0., 0.);
                    if (j < n) {
                        double vv = u01(wa.x, wa.y) * vi.x + u01(wa.z, wa.w) * (pa.x - xi.x)
                                + phi * u01(wc.x, wc.y) * (me.x - xi.x);
                        const double maxv = c.vmax * (upb[j] - lob[j]);
                        vv = fmax(-maxv, fmin(vv, maxv));
                        double xx = xi.x + vv;
                        if (c.correct) xx = fmax(lob[j], fmin(xx, upb[j]));
                        vn.x = vv;
                        xn.x = xx;
                    }
                    if (j + 1 < n) {
                        double vv = u01(wb.x, wb.y) * vi.y + u01(wb.z, wb.w) * (pa.y - xi.y)
                                + phi * u01(wc.z, wc.w) * (me.y - xi.y);
                        const double maxv = c.vmax * (upb[j + 1] - lob[j + 1]);
                        vv = fmax(-maxv, fmin(vv, maxv));
                        double xx = xi.y + vv;
                        if (c.correct) xx = fmax(lob[j + 1], fmin(xx, upb[j + 1]));
                        vn.y = vv;
                        xn.y = xx;
                    }
                    *reinterpret_cast<double2*>(&x[j]) = xn;
                    *reinterpret_cast<double2*>(&v[j]) = vn;
                    *reinterpret_cast<double2*>(&trial[j]) = xn;
                    ssq += xn.x * xn.x + xn.y * xn.y;
                    if (FUSE) {                  // (one pass over the row: u names the column pair)
                        csum[u].x += xn.x;
                        csum[u].y += xn.y;
                        if (k == 1) {            // the winner's row is this loser's parent
                            csum[u].x += pa.x;
                            csum[u].y += pa.y;
                        }
                    }
                }
            }
        }
        cso_wave_sync();          // (a team sits inside one wavefront: its row is its own)
        ssq = cso_group_sum<G>(ssq);
        double f = CSO_INF;
        if (c.obj >= 0) {
            f = eval_row_group<G>(c.obj, n, trial, d.aux, g);
            if (f != f) f = CSO_INF;
        }
        if (live && g == 0) {
            if (c.obj >= 0) d.f[pb + row] = f;
            d.radius[pb + row] = sqrt(ssq);
        }
        // the next (better) loser of this group reads x of its own parent only; its own row
        // was last written in an earlier generation.  The wavefront barrier orders the LDS reuse.
        cso_wave_sync();
    }
    if (FUSE) {
        // the teams' sums through their (now free) LDS rows, then the workgroup's teams in order
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int pj = g + G * u;
            if (pj < (ld >> 1))
                *reinterpret_cast<double2*>(&trial[2 * pj]) = live ? csum[u] : make_double2(0., 0.);
        }
        __syncthreads();
        cso_team_rows_to_wgpart(d, c, p, lds, blockDim.x / G);
    }
}

// The same sums from the swarm as it stands (after cso_init: no cso_compete has run yet), in the
// same order: a team adds its group's rows from the last slot of the group up to the first, the
// workgroup its teams in order.  grid (nwg, P), 256 threads, LDS (256 / G) * ld doubles
template<int G>
__global__ __launch_bounds__(256) void cso_team_colsum(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid / G, g = tid % G, ld = c.ld;
    const int gI = blockIdx.x * (blockDim.x / G) + r;
    const bool live = gI < c.ngroup;
    const size_t pb = (size_t) p * c.np;
    const int *occ = d.occ + pb + (size_t) (live ? gI : 0) * c.pc;
    double *trial = lds + r * ld;
    for (int pj = g; pj < (ld >> 1); pj += G) {
        double2 sum = make_double2(0., 0.);
        if (live)
            for (int k = c.pc - 1; k >= 0; k--) {
                const double2 x = *reinterpret_cast<const double2*>(
                        &d.X[(pb + occ[k]) * ld + 2 * pj]);
                sum.x += x.x;
                sum.y += x.y;
            }
        *reinterpret_cast<double2*>(&trial[2 * pj]) = sum;
    }
    __syncthreads();
    cso_team_rows_to_wgpart(d, c, p, lds, blockDim.x / G);
}

// colpart[p][part] = the workgroup sums [part * per, (part + 1) * per) of wgpart, in order
// (cso_mean finishes).  grid (parts, P), 256 threads
__global__ __launch_bounds__(256) void cso_wgsum(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y, part = blockIdx.x;
    if (cso_frozen(c, d.scal + p)) return;
    const int per = (c.nwg + c.parts - 1) / c.parts;
    const int q0 = part * per, q1 = min(c.nwg, q0 + per);
    const double *src = d.wgpart + (size_t) p * c.nwg * c.ld;
    for (int j = threadIdx.x; j < c.ld; j += 256) {
        double s = 0.;
        int q = q0;
        for (; q + 8 <= q1; q += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = src[(size_t) (q + u) * c.ld + j];
#pragma unroll
            for (int u = 0; u < 8; u++) s += x[u];
        }
        for (; q < q1; q++) s += src[(size_t) q * c.ld + j];
        d.colpart[((size_t) p * c.parts + part) * c.ld + j] = s;
    }
}

// incumbent (first slot holding the smallest f, cso.cpp:150-156) and the spread of the radii
// (:177-194), first over slabs of the swarm: slab s covers slots AND rows [s*per, (s+1)*per).
// fpart[p][s] = { best f, its slot, count, mean radius, sum of squared deviations }.
// grid (fparts, P), 256 threads
constexpr int CSO_FPART = 5;
__global__ __launch_bounds__(256) void cso_finish_part(CsoDev d, CsoConst c)
{
    const int p = blockIdx.y, part = blockIdx.x;
    const CsoScal *sc = d.scal + p;
    if (cso_frozen(c, sc)) return;
    __shared__ double sval[4];
    __shared__ int sidx[4];
    __shared__ double scratch[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t pb = (size_t) p * c.np;
    const int per = (c.np + c.fparts - 1) / c.fparts;
    const int lo = part * per, hi = min(c.np, lo + per);
    double best = CSO_INF;
    int bslot = 0x7fffffff;
    for (int s0 = lo + tid; s0 < hi; s0 += 4 * 256) {      // four gathers in flight
        double fv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int s = s0 + 256 * u;
            fv[u] = s < hi ? d.f[pb + d.occ[pb + s]] : CSO_INF;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int s = s0 + 256 * u;
            if (s < hi && (fv[u] < best || (fv[u] == best && s < bslot))) {
                best = fv[u];
                bslot = s;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bslot, off, 64);
        if (ov < best || (ov == best && oi < bslot)) {
            best = ov;
            bslot = oi;
        }
    }
    if (lane == 0) {
        sval[wave] = best;
        sidx[wave] = bslot;
    }
    __syncthreads();
    best = sval[0];
    bslot = sidx[0];
    for (int w = 1; w < 4; w++)
        if (sval[w] < best || (sval[w] == best && sidx[w] < bslot)) {
            best = sval[w];
            bslot = sidx[w];
        }
    auto block_sum = [&](double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if (lane == 0) scratch[wave] = v;
        __syncthreads();
        return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    };
    // (radii by ROW: the same set as the slots, coalesced and without the gather)
    double rv[4];
    double s = 0.;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int q = lo + tid + 256 * u;
        rv[u] = q < hi ? d.radius[pb + q] : 0.;
        s += rv[u];
    }
    for (int q = lo + tid + 1024; q < hi; q += 256) s += d.radius[pb + q];
    const int cnt = max(hi - lo, 0);
    const double mean = cnt > 0 ? block_sum(s) / cnt : 0.;
    double m2 = 0.;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int q = lo + tid + 256 * u;
        const double dd = rv[u] - mean;
        if (q < hi) m2 += dd * dd;
    }
    for (int q = lo + tid + 1024; q < hi; q += 256) {
        const double dd = d.radius[pb + q] - mean;
        m2 += dd * dd;
    }
    m2 = block_sum(m2);
    if (tid == 0) {
        double *out = d.fpart + ((size_t) p * c.fparts + part) * CSO_FPART;
        out[0] = best;
        out[1] = (double) bslot;
        out[2] = (double) cnt;
        out[3] = mean;
        out[4] = m2;
    }
}

// combines the slabs (arg-min lexicographic in (f, slot); mean / M2 by the pairwise update of
// Chan et al.), then evaluation count and stop test.  One wavefront per population.
__global__ __launch_bounds__(64) void cso_finish(CsoDev d, CsoConst c, int init_only)
{
    const int p = blockIdx.x;
    CsoScal *sc = d.scal + p;
    if (cso_frozen(c, sc)) return;
    if (threadIdx.x != 0) return;
    const size_t pb = (size_t) p * c.np;
    const double *in = d.fpart + (size_t) p * c.fparts * CSO_FPART;
    double best = CSO_INF, bslot = 2147483647., cnt = 0., mean = 0., m2 = 0.;
    for (int s = 0; s < c.fparts; s++) {
        const double *e = in + (size_t) s * CSO_FPART;
        if (e[0] < best || (e[0] == best && e[1] < bslot)) {
            best = e[0];
            bslot = e[1];
        }
        if (e[2] > 0.) {
            const double nb = e[2], tot = cnt + nb, delta = e[3] - mean;
            m2 += e[4] + delta * delta * (cnt * nb / tot);
            mean += delta * (nb / tot);
            cnt = tot;
        }
    }
    const int islot = (int) bslot;
    sc->ibest = islot < c.np ? d.occ[pb + islot] : 0;
    sc->fbest = best;
    sc->m2 = m2;
    if (init_only) {
        sc->conv = m2 <= (c.np - 1) * c.stol * c.stol ? 1 : 0;
        return;
    }
    sc->fev += c.np - c.ngroup;
    sc->gen += 1;
    const int conv = m2 <= (c.np - 1) * c.stol * c.stol ? 1 : 0;
    sc->conv = conv;
    if (conv) sc->stop = 1;
    else if (sc->fev >= c.mfev) sc->stop = 2;
}

} // namespace bbo
