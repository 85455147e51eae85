// bbo_ccpso_kernels.hpp -- one CCPSO2 generation as gfx950 kernels.
//
//   kernel          reference lines (ccpso.cpp)
//   ccp_init        :76-100 uniform swarm, Y = X, yhat = the best particle
//   ccp_regroup     :188-231 subset size (kept while yhat improves), random regrouping
//                   (std::shuffle -> the keyed Feistel bijection of bbo_cso_kernels.hpp)
//   ccp_eval<G>     :233-247 + evaluate :150-168: 2 nswarm np context-vector evaluations (yhat
//                   with one swarm's coordinates replaced), teams of G lanes per SWARM walking
//                   its candidates in one LDS row: evaluation latency x occupancy, the
//                   throughput driver
//   ccp_update      :251-290 personal bests, swarm bests into yhat (last qualifying particle
//                   wins, stale fY -- both kept), ring local bests
//   ccp_yhat        :292-303 re-evaluation of a moved yhat, accept / revert; :306-332 Cauchy rate
//   ccp_strategy + ccp_position   :335-371 Cauchy / normal resampling around y_i / y_lbest
//   ccp_finish      :170-186 radius spread, stop
#pragma once

#include "bbo_ccpso.hpp"
#include "bbo_objectives.hpp"
#include "bbo_rng.hpp"

namespace bbo {

#define CCP_INF (__builtin_huge_val())
constexpr double CCP_PI = 3.14159265358979323846;

__device__ inline bool ccp_frozen(const CcpConst &c, const CcpScal *sc)
{
    return c.honor_stop && sc->stop != 0;
}

template<int G>
__device__ inline double ccp_group_sum(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

// grid (ceil(np/16), P), 256 threads, LDS 16 * ld doubles (n <= 1024) -- init only
__global__ __launch_bounds__(256) void ccp_init(CcpDev d, CcpConst c)
{
    const int p = blockIdx.y;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * 16 + r, ld = c.ld;
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
            *reinterpret_cast<double2*>(&d.Y[base + j]) = v;
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    __syncthreads();
    ssq = ccp_group_sum<16>(ssq);
    double f = CCP_INF;
    if (c.obj >= 0) {
        f = eval_row_group<16>(c.obj, c.n, row, d.aux, g);
        if (f != f) f = CCP_INF;
    }
    if (g == 0 && i < c.np) {
        d.fX[(size_t) p * c.n * c.np + i] = f;        // scratch: the initial fitness
        d.radius[(size_t) p * c.np + i] = sqrt(ssq);
    }
}

// yhat = the first particle with the smallest initial fitness.  grid (P), 256 threads
__global__ __launch_bounds__(256) void ccp_init_yhat(CcpDev d, CcpConst c)
{
    const int p = blockIdx.x;
    __shared__ int sbest;
    CcpScal *sc = d.scal + p;
    if (threadIdx.x == 0) {
        const double *f = d.fX + (size_t) p * c.n * c.np;
        double best = CCP_INF;
        int ib = 0;
        for (int i = 0; i < c.np; i++)
            if (f[i] < best) {
                best = f[i];
                ib = i;
            }
        sbest = ib;
        sc->fyhat = best;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < c.ld; j += 256)
        d.yhat[(size_t) p * c.ld + j] = d.X[((size_t) p * c.np + sbest) * c.ld + j];
}

// grid (P), 256 threads
__global__ __launch_bounds__(256) void ccp_regroup(CcpDev d, CcpConst c)
{
    const int p = blockIdx.x;
    CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    __shared__ int s_changed, s_cp;
    const int tid = threadIdx.x, gen = sc->gen;
    const uint32_t sw = stream_word(STREAM_PSO_CTRL, (uint32_t) p);
    if (tid == 0) {
        const int is0 = sc->is;
        int is = is0;
        if (!sc->improved) {
            const u32x4 w = philox4x32_10(c.seed, 0, 0, (uint32_t) gen, sw);
            is = uint_below(w.x, c.npps);
        }
        s_changed = is != is0;
        if (is != is0) {
            sc->is = is;
            sc->cpswarm = c.pps[is];
            sc->nswarm = c.n / c.pps[is];
        }
        s_cp = sc->cpswarm;
        sc->yupd = 0;
        sc->fyhat0 = sc->fyhat;
    }
    __syncthreads();
    const int cp = s_cp;
    if (s_changed)   // _strat is re-created (zeros) with the new shape, ccpso.cpp:209-210
        for (int q = tid; q < c.n * c.np; q += 256) d.strat[(size_t) p * c.n * c.np + q] = 0;
    int bits = 1;
    while ((1u << bits) < (unsigned) c.n) bits++;
    const int kb = (bits + 1) / 2;
    for (int q = tid; q < c.n; q += 256) {
        const int coord = (int) cso_perm((uint32_t) q, kb, (uint32_t) c.n, c.seed, (uint32_t) gen, sw);
        d.range[(size_t) p * c.n + q] = coord;
        d.grp_of[(size_t) p * c.n + coord] = q / cp;
    }
    for (int j = tid; j < c.ld; j += 256) d.ysave[(size_t) p * c.ld + j] = d.yhat[(size_t) p * c.ld + j];
}

// orders a wavefront's LDS accesses among its own lanes
__device__ inline void ccp_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// candidate (j, i, which): yhat with the coordinates of swarm j taken from X_i (which = 0) or
// Y_i (1) -- ccpso.cpp:241-260, 2 np nswarm context evaluations per generation.  CCP_SPLIT TEAMS
// of G lanes per SWARM j: a team stages yhat in its LDS row once and then walks its share of the
// swarm's 2 np candidates, which all replace the SAME coordinates -- a candidate costs the gather of its cp
// values and one evaluation, not a copy of the n-vector (the form with one team per candidate
// spent its time staging 8 KB per evaluation: 749 us per generation at n = 1000 against
// the figure in HISTORY.md section 3).  The values in the row and the reduction are what they
// were: f is bit-identical.  Teams sit inside one wavefront: syncs at wavefront scope.
// grid (ceil(CCP_SPLIT nswarm_max / (256/G)), P), 256 threads, LDS (256/G) * ld doubles -- swarms past
// nswarm, or outside this rank's shard [nswarm r / W, nswarm (r + 1) / W), return
constexpr int CCP_SPLIT = 4;          // teams per swarm (candidate t goes to team t mod CCP_SPLIT)

template<int G>
__global__ __launch_bounds__(256) void ccp_eval(CcpDev d, CcpConst c)
{
    const int p = blockIdx.y;
    const CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    extern __shared__ double lds[];
    constexpr int R = 256 / G;
    constexpr int MAXQ = 256 / G;                // coordinates of the swarm held in registers: 256
    const int tid = threadIdx.x, r = tid / G, g = tid % G;
    const int team = blockIdx.x * R + r, ld = c.ld, np = c.np;
    const int j = team / CCP_SPLIT, part = team - j * CCP_SPLIT;
    const int nswarm = sc->nswarm, cp = sc->cpswarm;
    const int jlo = (int) ((long) nswarm * c.shard_rank / c.shard_world);
    const int jhi = (int) ((long) nswarm * (c.shard_rank + 1) / c.shard_world);
    if ((int) (blockIdx.x * R) >= jhi * CCP_SPLIT || (int) (blockIdx.x * R + R) <= jlo * CCP_SPLIT)
        return;
    if (j < jlo || j >= jhi) return;             // (whole teams: nobody waits for them)
    double *row = lds + (size_t) r * ld;
    const double *yh = d.yhat + (size_t) p * ld;
    for (int q = g; q < ld; q += G) row[q] = yh[q];
    // this lane's coordinates of the swarm (the same for every candidate) and, one candidate
    // ahead, the values that go there: the gather is a chain of two dependent loads, taken off
    // the evaluation's path
    const int *rg = d.range + (size_t) p * c.n + (size_t) j * cp;
    int coord[MAXQ];
#pragma unroll
    for (int u = 0; u < MAXQ; u++) coord[u] = g + G * u < cp ? rg[g + G * u] : -1;
    const size_t fb = (size_t) p * c.n * np + (size_t) j * np;
    auto source = [&](int t) {
        return ((t & 1) ? d.Y : d.X) + ((size_t) p * np + (t >> 1)) * ld;
    };
    double val[MAXQ];
    {
        const double *src = source(part);
#pragma unroll
        for (int u = 0; u < MAXQ; u++) val[u] = coord[u] >= 0 ? src[coord[u]] : 0.;
    }
    for (int t = part; t < 2 * np; t += CCP_SPLIT) {
        ccp_wave_sync();                         // the previous evaluation has read the row
#pragma unroll
        for (int u = 0; u < MAXQ; u++)
            if (coord[u] >= 0) row[coord[u]] = val[u];
        if (cp > G * MAXQ) {                     // (swarms wider than 256 coordinates: the rest, plainly)
            const double *src = source(t);
            for (int q = g + G * MAXQ; q < cp; q += G) {
                const int co = rg[q];
                row[co] = src[co];
            }
        }
        if (t + CCP_SPLIT < 2 * np) {
            const double *src = source(t + CCP_SPLIT);
#pragma unroll
            for (int u = 0; u < MAXQ; u++) val[u] = coord[u] >= 0 ? src[coord[u]] : 0.;
        }
        ccp_wave_sync();
        if (c.obj >= 0) {
            double f = eval_row_group<G, false, 4>(c.obj, c.n, row, d.aux, g);
            if (f != f) f = CCP_INF;
            if (g == 0) ((t & 1) ? d.fY : d.fX)[fb + (t >> 1)] = f;
        }
    }
}

// sharded swarm groups.  A rank's record holds ONLY the rows of the swarms it evaluated: with
// `stride` = ceil(max swarms / world) * np doubles per table, record r = [fX block | fY block],
// block element (j - j0_r) * np + i for swarm j of rank r's range [j0_r, j1_r), particle i.
// (Round 2 exchanged the two full-capacity tables per rank: W times the live data.)
__device__ inline int ccp_shard_lo(int nswarm, int r, int world)
{
    return (int) ((long) nswarm * r / world);
}

// this rank's block into `dst` (device memory).  grid (ceil(stride / 256)), 256 threads
__global__ __launch_bounds__(256) void ccp_export(CcpDev d, CcpConst c, double *dst, int stride)
{
    const int q = blockIdx.x * 256 + threadIdx.x, np = c.np, nswarm = d.scal->nswarm;
    if (q >= stride) return;
    const int j0 = ccp_shard_lo(nswarm, c.shard_rank, c.shard_world);
    const int j1 = ccp_shard_lo(nswarm, c.shard_rank + 1, c.shard_world);
    const bool live = q < (j1 - j0) * np;
    dst[q] = live ? d.fX[(size_t) j0 * np + q] : 0.;
    dst[stride + q] = live ? d.fY[(size_t) j0 * np + q] : 0.;
}

// take every swarm's rows from the record of its owner.
// grid (ceil(n * np / 256)), 256 threads; population 0 only
__global__ __launch_bounds__(256) void ccp_merge(CcpDev d, CcpConst c, const double *gathered,
        int world, int stride)
{
    const CcpScal *sc = d.scal;
    const int q = blockIdx.x * 256 + threadIdx.x, np = c.np, nswarm = sc->nswarm;
    if (q >= nswarm * np) return;
    const int j = q / np;
    int owner = 0;
    for (int r = 0; r < world; r++)
        if (j >= ccp_shard_lo(nswarm, r, world) && j < ccp_shard_lo(nswarm, r + 1, world))
            owner = r;
    const double *rec = gathered + (size_t) owner * 2 * stride;
    const int off = q - ccp_shard_lo(nswarm, owner, world) * np;
    d.fX[q] = rec[off];
    d.fY[q] = rec[stride + off];
}

// one WAVEFRONT per swarm j: personal bests, the swarm's contribution to yhat, ring local bests
// (np cp elements, often a few dozen: a wider workgroup only adds idle wavefronts to schedule).
// grid (n (>= nswarm), P), 64 threads
__global__ __launch_bounds__(64) void ccp_update(CcpDev d, CcpConst c)
{
    const int p = blockIdx.y, j = blockIdx.x;
    CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    if (j >= sc->nswarm) return;
    const int tid = threadIdx.x, np = c.np, ld = c.ld, cp = sc->cpswarm;
    const size_t fb = (size_t) p * c.n * np + (size_t) j * np;
    const double *fX = d.fX + fb, *fY = d.fY + fb;
    const int *rg = d.range + (size_t) p * c.n + (size_t) j * cp;
    const double fyhat = sc->fyhat0;
    // personal bests: Y_i <- X_i on this swarm's coordinates where X scored better
    for (int q = tid; q < np * cp; q += 64) {
        const int i = q / cp, coord = rg[q - i * cp];
        if (fX[i] < fY[i]) d.Y[((size_t) p * np + i) * ld + coord] = d.X[((size_t) p * np + i) * ld + coord];
    }
    // the LAST particle whose (stale) fY beats fyhat gives the swarm's coordinates to yhat
    int last = -1;
    for (int i = tid; i < np; i += 64)
        if (fY[i] < fyhat) last = i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) last = max(last, __shfl_xor(last, off, 64));
    // (one wavefront: the reduction above is the whole workgroup's; the fence orders the Y writes
    // above before the reads below)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (last >= 0) {
        for (int q = tid; q < cp; q += 64) {
            const int coord = rg[q];
            d.yhat[(size_t) p * ld + coord] = d.Y[((size_t) p * np + last) * ld + coord];
        }
        if (tid == 0) atomicOr(&sc->yupd, 1);
    }
    // ring local best: the first smallest of (i-1, i, i+1)
    int *ib = d.ibest + fb;
    for (int i = tid; i < np; i += 64) {
        const int a = (i - 1 + np) % np, b = (i + 1) % np;
        int im = a;
        if (fY[i] < fY[im]) im = i;
        if (fY[b] < fY[im]) im = b;
        ib[i] = im;
    }
}

// a moved yhat is re-evaluated, kept only if better; then the Cauchy rate.  grid (P), 256
// threads, LDS ld doubles (the objective reads yhat from there)
__global__ __launch_bounds__(256) void ccp_yhat(CcpDev d, CcpConst c)
{
    const int p = blockIdx.x;
    CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    extern __shared__ double lds[];
    __shared__ double red[4][4];
    const int tid = threadIdx.x, ld = c.ld, np = c.np;
    const int nswarm = sc->nswarm;
    double *yh = d.yhat + (size_t) p * ld;
    double fnew = sc->fyhat0;
    const int yupd = sc->yupd;
    if (yupd && c.obj >= 0) {
        for (int q = tid; q < ld; q += 256) lds[q] = yh[q];
        __syncthreads();
        double f = 0.;
        if (tid < 64) {
            f = eval_row_group<64>(c.obj, c.n, lds, d.aux, tid);
            if (f != f) f = CCP_INF;
        }
        f = __shfl(f, 0, 64);
        if (tid == 0) red[0][0] = f;
        __syncthreads();
        fnew = red[0][0];
        __syncthreads();
    } else if (yupd) {
        fnew = sc->fyhat;                // host objective: the host stored f(yhat) here
    }
    const bool improved = yupd && fnew < sc->fyhat0;
    if (!improved)
        for (int q = tid; q < ld; q += 256) yh[q] = d.ysave[(size_t) p * ld + q];
    // success rates of the two resampling kinds (ccpso.cpp:306-332)
    double cs = 0., ns = 0., ct = 0., nt = 0.;
    const size_t fb = (size_t) p * c.n * np;
    for (int q = tid; q < nswarm * np; q += 256) {
        const int st = d.strat[fb + q];
        const bool ok = d.fX[fb + q] < d.fY[fb + q];
        if (st == 0) {
            ct += 1.;
            cs += ok ? 1. : 0.;
        } else {
            nt += 1.;
            ns += ok ? 1. : 0.;
        }
    }
    double v[4] = { cs, ns, ct, nt };
#pragma unroll
    for (int u = 0; u < 4; u++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[u] += __shfl_xor(v[u], off, 64);
        if ((tid & 63) == 0) red[tid >> 6][u] = v[u];
    }
    __syncthreads();
    if (tid == 0) {
        double tot[4];
        for (int u = 0; u < 4; u++) tot[u] = red[0][u] + red[1][u] + red[2][u] + red[3][u];
        sc->fev += 2 * nswarm * np + (yupd ? 1 : 0);
        sc->improved = improved ? 1 : 0;
        sc->fyhat = improved ? fnew : sc->fyhat0;
        if (sc->gen > 0 && c.adaptp) {
            const double crate = tot[0] / fmax(1., tot[2]);
            const double nrate = tot[1] / fmax(1., tot[3]);
            sc->phat = fmax(0.05, fmin(crate / fmax(1., crate + nrate), 0.95));
        }
    }
}

// resampling kind of every (swarm, particle).  grid (ceil(n*np/256), P)
__global__ __launch_bounds__(256) void ccp_strategy(CcpDev d, CcpConst c)
{
    const int p = blockIdx.y;
    const CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    const int q = blockIdx.x * 256 + threadIdx.x, np = c.np;
    if (q >= sc->nswarm * np) return;
    const int j = q / np, i = q - j * np;
    const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (16 + j), (uint32_t) sc->gen,
            stream_word(STREAM_PSO_CTRL, (uint32_t) p));
    d.strat[(size_t) p * c.n * np + q] = u01(w.x, w.y) < sc->phat ? 0 : 1;
}

// new positions: 16 lanes per particle sweep its coordinates.  grid (ceil(np/16), P)
__global__ __launch_bounds__(256) void ccp_position(CcpDev d, CcpConst c)
{
    // one thread per coordinate pair: grid (ceil(ld / 512), np, P).  (With 16 lanes per particle
    // the whole resampling was 2 np workgroups -- 32 for the benchmark's 16 populations -- walking
    // chains of dependent look-ups, tangents and Box-Muller pairs: 90 us of latency.)  The
    // squared radius leaves as one partial per workgroup; ccp_finish adds them in order.
    const int p = blockIdx.z, i = blockIdx.y;
    const CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    __shared__ double red[4];
    const int tid = threadIdx.x, np = c.np, ld = c.ld, gen = sc->gen;
    const int pj = blockIdx.x * 256 + tid;
    double ssq = 0.;
    if (pj < ld / 2) {
        const size_t fb = (size_t) p * c.n * np;
        const int *grp = d.grp_of + (size_t) p * c.n;
        const double *Y = d.Y + (size_t) p * np * ld;
        double *x = d.X + ((size_t) p * np + i) * ld;
        const uint32_t sw = stream_word(STREAM_PSO_R, (uint32_t) p);
        const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) pj, (uint32_t) gen, sw);
        double z0 = 0., z1 = 0.;
        bool have_z = false;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int dd = 2 * pj + h;
            if (dd >= c.n) continue;
            const int j = grp[dd];
            const int st = d.strat[fb + (size_t) j * np + i];
            const int ihat = d.ibest[fb + (size_t) j * np + i];
            const double yi = Y[(size_t) i * ld + dd], yl = Y[(size_t) ihat * ld + dd];
            double dev;
            if (st == 0) {
                const double u = h ? u01(w.z, w.w) : u01(w.x, w.y);
                dev = tan(CCP_PI * (u - 0.5));
            } else {
                if (!have_z) {
                    normal_pair(c.seed, (uint32_t) i, (uint32_t) pj, (uint32_t) gen, sw, z0, z1);
                    have_z = true;
                }
                dev = h ? z1 : z0;
            }
            double v = (st == 0 ? yi : yl) + dev * fabs(yi - yl);
            if (c.correct) v = fmax(d.lower[dd], fmin(v, d.upper[dd]));
            x[dd] = v;
            ssq += v * v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ssq += __shfl_xor(ssq, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
    __syncthreads();
    if (tid == 0)
        d.rpart[((size_t) p * np + i) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// grid (P), 256 threads
// rparts > 0: the radii come as `rparts` partial sums of squares per particle (ccp_position)
__global__ __launch_bounds__(256) void ccp_finish(CcpDev d, CcpConst c, int rparts)
{
    const int p = blockIdx.x;
    CcpScal *sc = d.scal + p;
    if (ccp_frozen(c, sc)) return;
    __shared__ double red[4];
    const int tid = threadIdx.x, np = c.np;
    if (rparts > 0)
        for (int q = tid; q < np; q += 256) {
            const double *rp = d.rpart + ((size_t) p * np + q) * rparts;
            double s2 = rp[0];
            for (int k = 1; k < rparts; k++) s2 += rp[k];
            d.radius[(size_t) p * np + q] = sqrt(s2);
        }
    auto block_sum = [&](double v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    double s = 0.;
    for (int q = tid; q < np; q += 256) s += d.radius[(size_t) p * np + q];
    const double mean = block_sum(s) / np;
    double m2 = 0.;
    for (int q = tid; q < np; q += 256) {
        const double dd = d.radius[(size_t) p * np + q] - mean;
        m2 += dd * dd;
    }
    m2 = block_sum(m2);
    if (tid == 0) {
        sc->gen += 1;
        sc->m2 = m2;
        const int conv = m2 <= (np - 1) * c.stol * c.stol ? 1 : 0;
        sc->conv = conv;
        // ccpso.cpp:137-146: the budget test comes first
        if (sc->fev >= c.mfev) sc->stop = 2;
        else if (conv) sc->stop = 1;
    }
}

} // namespace bbo
