// bbo_de_kernels.hpp -- one L-SHADE / JADE generation as gfx950 kernels.
//
//   kernel          reference lines it replaces                               bound
//   de_init         shade.cpp:77-93 / jade.cpp:84-95 (uniform population)      HBM (8n B/ind.)
//   de_generation   shade.cpp:98-184 / jade.cpp:104-178 (the i-loop, fused)    HBM: 4 row reads + 1
//                                                                              row write = 40n+16 B
//   de_bookkeep     archive :164-172, success memory :188-212 / jade :180-205  latency (1 WG)
//   de_archive_copy archive row copies                                         HBM (16n B/success)
//   de_rank         std::sort shade.cpp:215 / jade.cpp:101                     L2 (np^2 compares)
//   de_finish       LPSR :218-225, archive trim :228-235, stop test :258-275   latency (1 WG)
//
// 16 lanes share one individual; lane g owns the coordinate PAIRS g, g+16, ... so every row
// access is a 16-byte-per-lane coalesced stream.  No MFMA: the path is byte-bound.
#pragma once

#include "bbo_de.hpp"
#include "bbo_objectives.hpp"
#include "bbo_rng.hpp"
#include "bbo_rank.hpp"

namespace bbo {

#define BBO_INF_D (__builtin_huge_val())
constexpr double DE_PI = 3.14159265358979323846;
constexpr int DE_MAX_TRIES = 64;

__device__ inline bool de_frozen(const DeConst &c, const DeScal *sc)
{
    return c.honor_stop && sc->stop != 0;
}

template<int G>
__device__ inline double group_sum_d(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

// ---------------------------------------------------------------------------
// initial population: x = lb + u (ub - lb), f = objective(x)
// grid (ceil(npinit/16), P), 256 threads; dynamic LDS 16 * ld doubles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void de_init(DeDev d, DeConst c)
{
    const int p = blockIdx.y;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * (blockDim.x >> 4) + r;
    const int ld = c.ld;
    double *row = lds + r * ld;
    double *X = d.X[0] + (size_t) p * c.npinit * ld;
    double ssq = 0.;
    if (i < c.npinit) {
        for (int pj = g; pj < ld / 2; pj += 16) {
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) pj, 0,
                    stream_word(STREAM_INIT, (uint32_t) p));
            double2 v = make_double2(0., 0.);
            const int j = 2 * pj;
            if (j < c.n) v.x = u01(w.x, w.y) * (d.upper[j] - d.lower[j]) + d.lower[j];
            if (j + 1 < c.n) v.y = u01(w.z, w.w) * (d.upper[j + 1] - d.lower[j + 1]) + d.lower[j + 1];
            *reinterpret_cast<double2*>(&row[j]) = v;
            *reinterpret_cast<double2*>(&X[(size_t) i * ld + j]) = v;
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    __syncthreads();
    ssq = group_sum_d<16>(ssq);
    if (c.obj >= 0) {
        double f = eval_row_group<16>(c.obj, c.n, row, d.aux, g);
        if (g == 0 && i < c.npinit) {
            if (f != f) f = BBO_INF_D;
            d.f[0][(size_t) p * c.npinit + i] = f;
        }
    }
    if (g == 0 && i < c.npinit) d.radius[(size_t) p * c.npinit + i] = sqrt(ssq);
}

// ---------------------------------------------------------------------------
// rank of f[which] over the first np individuals (stable in the row index)
// grid (ceil(np_launch/32), P), 256 threads = 32 individuals x 8 slices
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void de_rank(DeDev d, DeConst c, int which_next)
{
    const int p = blockIdx.y;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    __shared__ __attribute__((aligned(16))) double tile[RANK_TILE];
    const int np = sc->np;
    const int which = which_next ? (sc->cur ^ 1) : sc->cur;
    const int tid = threadIdx.x;
    const int cand = blockIdx.x * 32 + (tid >> 3), slice = tid & 7;
    const double *f = d.f[which] + (size_t) p * c.npinit;
    const int cnt = rank_by_counting(f, np, cand, slice, tile);
    if (cand < np && slice == 0) {
        d.rank[(size_t) p * c.npinit + cand] = cnt;
        d.order[(size_t) p * c.npinit + cnt] = cand;
    }
}

// in-LDS bitonic sort per population (np <= SORT_LDS_MAX): grid (P), 1024 threads
__global__ __launch_bounds__(1024) void de_rank_sort(DeDev d, DeConst c, int which_next, int m)
{
    const int p = blockIdx.x;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double sortbuf[];
    double *keys = sortbuf;
    int *idx = reinterpret_cast<int*>(sortbuf + max(m, 1024));   // the sort pads to >= 1024
    const int which = which_next ? (sc->cur ^ 1) : sc->cur;
    const double *f = d.f[which] + (size_t) p * c.npinit;
    if (m == 2048 || m == 4096) {
        int *ibuf = reinterpret_cast<int*>(sortbuf + 2 * m);
        if (m == 2048)
            merge_sort_lds<2>(f, sc->np, sortbuf, ibuf, d.order + (size_t) p * c.npinit,
                    d.rank + (size_t) p * c.npinit, &keys, &idx);
        else
            merge_sort_lds<4>(f, sc->np, sortbuf, ibuf, d.order + (size_t) p * c.npinit,
                    d.rank + (size_t) p * c.npinit, &keys, &idx);
        return;
    }
    bitonic_sort_lds(f, sc->np, m, keys, idx, d.order + (size_t) p * c.npinit,
            d.rank + (size_t) p * c.npinit);
}

// np <= 64: one wavefront per population, the bitonic network in registers and lane exchanges
// (no LDS, no barrier), four populations per workgroup.  grid (ceil(P / 4)), 256 threads
__global__ __launch_bounds__(256) void de_rank_wave(DeDev d, DeConst c, int which_next)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= c.npop) return;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    const int which = which_next ? (sc->cur ^ 1) : sc->cur;
    const double *f = d.f[which] + (size_t) p * c.npinit;
    const int np = sc->np;
    double kf[1] = { lane < np ? f[lane] : __builtin_huge_val() };
    int ki[1] = { lane < np ? lane : 0x7fffffff };
    for (int k = 2; k <= 64; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            switch (j) {
            case 1: sort_wave_stage<1, 1>(kf, ki, lane, j, k); break;
            case 2: sort_wave_stage<1, 2>(kf, ki, lane, j, k); break;
            case 4: sort_wave_stage<1, 4>(kf, ki, lane, j, k); break;
            case 8: sort_wave_stage<1, 8>(kf, ki, lane, j, k); break;
            case 16: sort_wave_stage<1, 16>(kf, ki, lane, j, k); break;
            default: sort_wave_stage<1, 32>(kf, ki, lane, j, k); break;
            }
        }
    if (lane < np) {
        d.order[(size_t) p * c.npinit + lane] = ki[0];
        d.rank[(size_t) p * c.npinit + ki[0]] = lane;
    }
}

// ---------------------------------------------------------------------------
// the fused generation.  grid (ceil(np_launch/16), P), 256 threads; LDS (16 + 2) * ld doubles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void de_generation(DeDev d, DeConst c)
{
    const int p = blockIdx.y;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * (blockDim.x >> 4) + r;
    const int ld = c.ld, n = c.n, np = sc->np, larch = sc->larch, gen = sc->gen, cur = sc->cur;
    const bool live = i < np;
    double *trial = lds + r * ld;
    // The box, once per workgroup, behind the rows (the host sizes the stage for two more): read
    // from global memory inside the crossover loop -- where the scheduler puts a load next to its
    // use -- it was a round trip to L2 per group of column pairs, on the critical path of a
    // workgroup that lives for about ten such round trips (round 3, read in the ISA).
    // (requested here, stored to LDS after the draws: the loads then cost no wait of their own;
    // rows of more than two columns per thread take the plain loop)
    double *lob = lds + (size_t) (blockDim.x >> 4) * ld, *upb = lob + ld;
    const bool box_regs = ld <= 2 * (int) blockDim.x;
    double blo[2] = { 0., 0. }, bup[2] = { 0., 0. };
    if (box_regs) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = min(tid + u * (int) blockDim.x, ld - 1);
            blo[u] = d.lower[j];
            bup[u] = d.upper[j];
        }
    } else {
        for (int j = tid; j < ld; j += blockDim.x) {
            lob[j] = d.lower[j];
            upb[j] = d.upper[j];
        }
    }
    const size_t pbase = (size_t) p * c.npinit;
    const double *Xc = d.X[cur] + pbase * ld;
    double *Xn = d.X[cur ^ 1] + pbase * ld;
    const double *fc = d.f[cur] + pbase;
    const int *order = d.order + pbase;
    const uint32_t sub = (uint32_t) p;
    // (the parent's row index: requested before the draws, used after them -- unsigned and
    // unconditional, so that nothing has to wait for it here, not even a sign extension)
    const unsigned oi = (unsigned) order[live ? i : 0];

    // ---- per-individual draws, shade.cpp:105-131 / jade.cpp:107-133.  The draws of one
    // individual are independent Philox calls, so the 16 lanes of its group make one each
    // (lane 0 the normal for CR, lane 1 the parameter word, lane 2 the pbest word, lanes 3-6
    // the first four Cauchy tries for F, 7-10 the first four tries for r1, 11-15 the first five
    // for r2) instead of lane 0 walking through all of them; the group then picks the first
    // admissible try exactly as the sequential rule does, and only falls back to a loop when
    // all speculative tries were rejected ---------------------------------------------------
    const uint32_t sw_param = stream_word(STREAM_DE_PARAM, sub);
    double mine_d = 0.;        // lane 0: z0; lanes 3-6: Cauchy deviate * 0.1
    uint32_t mine_x = 0, mine_y = 0, mine_z = 0, mine_w = 0;
    if (live) {
        if (g == 0) {
            double z1;
            normal_pair(c.seed, (uint32_t) i, 0, (uint32_t) gen, sw_param, mine_d, z1);
        } else {
            const uint32_t ctr = g == 1 ? 1u : g == 2 ? 2u : g <= 6 ? (uint32_t) (16 + g - 3)
                    : g <= 10 ? (uint32_t) (96 + g - 7) : (uint32_t) (160 + g - 11);
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, ctr, (uint32_t) gen, sw_param);
            mine_x = w.x; mine_y = w.y; mine_z = w.z; mine_w = w.w;
            if (g >= 3 && g <= 6) mine_d = tan(DE_PI * (u01(w.x, w.y) - 0.5)) * 0.1;
        }
    }
    const double z0 = __shfl(mine_d, 0, 16);
    const uint32_t p1x = __shfl(mine_x, 1, 16), p1y = __shfl(mine_y, 1, 16),
            p1z = __shfl(mine_z, 1, 16), p1w = __shfl(mine_w, 1, 16);
    const uint32_t p2x = __shfl(mine_x, 2, 16);
    double CR = 0., F = 0.;
    int ibest = 0, r1 = 0, r2 = 0, jrand = 0;
    {
        const int ri = uint_below(p1x, c.h);
        jrand = uint_below(p1y, n);
        const double up = u01(p1z, p1w);
        const double mcr = !live ? 0. : c.variant == 0 ? d.MCR[(size_t) p * c.h + ri] : sc->mucr;
        const double mf = !live ? 0.5 : c.variant == 0 ? d.MF[(size_t) p * c.h + ri] : sc->muf;
        CR = fmax(0., fmin(z0 * 0.1 + mcr, 1.));
        bool got = false;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double dev = __shfl(mine_d, 3 + t, 16);
            const double Ft = fmin(1., dev + mf);
            const bool ok = c.variant == 0 ? (Ft > 0.) : !(Ft < 0.);
            if (!got && ok) {
                F = Ft;
                got = true;
            }
        }
        for (int t = 4; t < DE_MAX_TRIES && !got && live; t++) {
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (16 + t), (uint32_t) gen,
                    sw_param);
            F = fmin(1., tan(DE_PI * (u01(w.x, w.y) - 0.5)) * 0.1 + mf);
            got = c.variant == 0 ? (F > 0.) : !(F < 0.);
        }
        if (!got) F = fmax(1e-8, fmin(1., mf));
        int nelite;
        if (c.variant == 0) {
            const double plo = fmin(2. / n, 0.2);
            const double pi = up * (0.2 - plo) + plo;
            nelite = max(1, (int) (pi * np));
        } else {
            nelite = max(1, (int) (c.pelite * np));
        }
        ibest = uint_below(p2x, nelite);
        r1 = -1;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int cnd = uint_below(__shfl(mine_x, 7 + t, 16), np);
            if (r1 < 0 && cnd != i) r1 = cnd;
        }
        for (int t = 4; t < DE_MAX_TRIES && r1 < 0 && live; t++) {
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (96 + t), (uint32_t) gen,
                    sw_param);
            const int cnd = uint_below(w.x, np);
            if (cnd != i) r1 = cnd;
        }
        if (r1 < 0) r1 = (i + 1) % max(np, 1);
        r2 = -1;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const int cnd = uint_below(__shfl(mine_x, 11 + t, 16), np + larch);
            if (r2 < 0 && cnd != i && cnd != r1) r2 = cnd;
        }
        for (int t = 5; t < DE_MAX_TRIES && r2 < 0 && live; t++) {
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (160 + t), (uint32_t) gen,
                    sw_param);
            const int cnd = uint_below(w.x, np + larch);
            if (cnd != i && cnd != r1) r2 = cnd;
        }
        if (r2 < 0) {
            r2 = 0;
            while (r2 == i || r2 == r1) r2++;
        }
    }

    // ---- mutation + binomial crossover + midpoint bound repair, into LDS.  Four column pairs
    // per lane at a time: all sixteen row loads of the four partners go out before the first
    // Philox call (the gathers are what this kernel waits for), the arithmetic is select-only ---
    if (box_regs) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = tid + u * (int) blockDim.x;
            if (j < ld) {
                lob[j] = blo[u];
                upb[j] = bup[u];
            }
        }
    }
    __syncthreads();           // lob / upb are staged
    double cnt = 0.;
    const double *xi = nullptr;
    double fold = 0.;          // the parent's fitness: requested with the partner rows
    const int npair = ld >> 1;
    double2 akeep[4];          // the parent's columns of the first pass (all of them if ld <= 128)
#pragma unroll
    for (int u = 0; u < 4; u++) akeep[u] = make_double2(0., 0.);
    if (live) {
        xi = Xc + (size_t) oi * ld;
        fold = fc[oi];
        const double *xb = Xc + (size_t) order[ibest] * ld;
        const double *x1 = Xc + (size_t) order[r1] * ld;
        const double *x2 = r2 >= np ? d.arch + (pbase + (r2 - np)) * ld
                                    : Xc + (size_t) order[r2] * ld;
        const uint32_t swc = stream_word(STREAM_DE_CROSS, sub);
        for (int pj0 = g; pj0 < npair; pj0 += 64) {
            double2 a[4], b[4], q1[4], q2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pj = pj0 + 16 * u;
                const int j = pj < npair ? 2 * pj : 0;
                a[u] = *reinterpret_cast<const double2*>(&xi[j]);
                b[u] = *reinterpret_cast<const double2*>(&xb[j]);
                q1[u] = *reinterpret_cast<const double2*>(&x1[j]);
                q2[u] = *reinterpret_cast<const double2*>(&x2[j]);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pj = pj0 + 16 * u;
                if (pj < npair) {
                    const int j = 2 * pj;
                    if (pj0 == g) akeep[u] = a[u];
                    const double2 lo = *reinterpret_cast<const double2*>(&lob[j]);
                    const double2 up = *reinterpret_cast<const double2*>(&upb[j]);
                    const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) pj,
                            (uint32_t) gen, swc);
                    const bool cx = j < n && (j == jrand || u01(w.x, w.y) < CR);
                    const bool cy = j + 1 < n && (j + 1 == jrand || u01(w.z, w.w) < CR);
                    const double mx = a[u].x + F * (b[u].x - a[u].x) + F * (q1[u].x - q2[u].x);
                    const double my = a[u].y + F * (b[u].y - a[u].y) + F * (q1[u].y - q2[u].y);
                    double2 v;
                    v.x = cx ? mx : a[u].x;
                    v.y = cy ? my : a[u].y;
                    cnt += (cx ? 1. : 0.) + (cy ? 1. : 0.);
                    const double rlx = (lo.x + a[u].x) / 2., rux = (up.x + a[u].x) / 2.;
                    const double rly = (lo.y + a[u].y) / 2., ruy = (up.y + a[u].y) / 2.;
                    v.x = j < n ? (v.x < lo.x ? rlx : (v.x > up.x ? rux : v.x)) : v.x;
                    v.y = j + 1 < n ? (v.y < lo.y ? rly : (v.y > up.y ? ruy : v.y)) : v.y;
                    *reinterpret_cast<double2*>(&trial[j]) = v;
                }
            }
        }
    }
    __syncthreads();
    cnt = group_sum_d<16>(cnt);

    // ---- evaluate, select into the other buffer ---------------------------------------------
    double ft = BBO_INF_D;
    if (c.obj >= 0) {
        ft = eval_row_group<16>(c.obj, n, trial, d.aux, g);
        if (ft != ft) ft = BBO_INF_D;
    }
    if (c.obj < 0) {
        // host objective: park the trial in the other buffer; de_select finishes the job
        if (live)
            for (int pj = g; pj < ld / 2; pj += 16)
                *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + 2 * pj]) =
                        *reinterpret_cast<const double2*>(&trial[2 * pj]);
        if (live && g == 0) {
            d.rec_cr[pbase + i] = c.repaircr ? cnt / n : CR;
            d.rec_f[pbase + i] = F;
        }
        return;
    }
    double ssq = 0.;
    bool accept = false;
    if (live) {
        accept = ft <= fold;
        // (the parent's first 128 columns are still in registers)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int pj = g + 16 * u;
            if (pj < npair) {
                const int j = 2 * pj;
                const double2 v = accept ? *reinterpret_cast<const double2*>(&trial[j]) : akeep[u];
                *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + j]) = v;
                ssq += v.x * v.x + v.y * v.y;
            }
        }
        for (int pj = g + 64; pj < npair; pj += 16) {
            const int j = 2 * pj;
            const double2 v = accept ? *reinterpret_cast<const double2*>(&trial[j])
                                     : *reinterpret_cast<const double2*>(&xi[j]);
            *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + j]) = v;
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    ssq = group_sum_d<16>(ssq);
    if (live && g == 0) {
        d.f[cur ^ 1][pbase + i] = accept ? ft : fold;
        d.radius[pbase + i] = sqrt(ssq);
        d.rec_cr[pbase + i] = c.repaircr ? cnt / n : CR;
        d.rec_f[pbase + i] = F;
        d.rec_df[pbase + i] = fold - ft;
        d.rec_flag[pbase + i] = (accept ? 1 : 0) | ((accept && ft < fold) ? 2 : 0);
    }
}

// ---------------------------------------------------------------------------
// SaNSDE (sansde.cpp:96-211): the fused generation.  Per individual: CR (persistent, redrawn
// every ncrref generations around crm), F from N(0.5, 0.3) or Cauchy(0, 1) by probability fp,
// three distinct partners, DE/rand/1 or DE/current-to-best/2 by probability p, binomial
// crossover (`<=`), midpoint repair, evaluation, replacement on strict improvement.
// Generation-synchronous like de_generation (oracle: Sansde::iterate_sync).
// grid (ceil(np/16), P), 256 threads; LDS (16 + 2) * ld doubles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sansde_generation(DeDev d, DeConst c)
{
    const int p = blockIdx.y;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * (blockDim.x >> 4) + r;
    const int ld = c.ld, n = c.n, np = sc->np, gen = sc->gen, cur = sc->cur;
    const bool live = i < np;
    double *trial = lds + r * ld;
    const size_t pbase = (size_t) p * c.npinit;
    const double *Xc = d.X[cur] + pbase * ld;
    double *Xn = d.X[cur ^ 1] + pbase * ld;
    const double *fc = d.f[cur] + pbase;
    const int *order = d.order + pbase;
    const uint32_t sw = stream_word(STREAM_DE_PARAM, (uint32_t) p);
    // as in de_generation: the box requested here and stored to LDS after the draws, the parent's
    // index requested first (unsigned, unconditional), its fitness with the partner rows
    double *lob = lds + (size_t) (blockDim.x >> 4) * ld, *upb = lob + ld;
    const bool box_regs = ld <= 2 * (int) blockDim.x;
    double blo[2] = { 0., 0. }, bup[2] = { 0., 0. };
    if (box_regs) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = min(tid + u * (int) blockDim.x, ld - 1);
            blo[u] = d.lower[j];
            bup[u] = d.upper[j];
        }
    } else {
        for (int j = tid; j < ld; j += blockDim.x) {
            lob[j] = d.lower[j];
            upb[j] = d.upper[j];
        }
    }
    const unsigned oi = (unsigned) order[live ? i : 0];

    double CR = 0., F = 0.5;
    int strat = 0, r1 = 0, r2 = 0, r3 = 0, jrand = 0;
    if (live && g == 0) {
        CR = d.crow[cur][pbase + oi];
        if (gen % c.ncrref == 0) {
            double z0, z1;
            normal_pair(c.seed, (uint32_t) i, 0, (uint32_t) gen, sw, z0, z1);
            CR = fmax(0., fmin(z0 * 0.1 + sc->crm, 1.));
        }
        u32x4 w = philox4x32_10(c.seed, (uint32_t) i, 1, (uint32_t) gen, sw);
        const int ifstrat = u01(w.x, w.y) < sc->sfp ? 0 : 1;
        const int istrat = u01(w.z, w.w) < sc->sp ? 0 : 1;
        strat = istrat | (ifstrat << 1);
        bool got = false;
        for (int t = 0; t < DE_MAX_TRIES && !got; t++) {
            if (ifstrat == 0) {
                double z0, z1;
                normal_pair(c.seed, (uint32_t) i, (uint32_t) (16 + t), (uint32_t) gen, sw, z0, z1);
                F = z0 * 0.3 + 0.5;
            } else {
                w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (16 + t), (uint32_t) gen, sw);
                F = tan(DE_PI * (u01(w.x, w.y) - 0.5));
            }
            F = fmin(F, 1.);
            got = !(F < 0.);
        }
        if (!got) F = 0.5;
        int rr[3] = { -1, -1, -1 };
#pragma unroll
        for (int q = 0; q < 3; q++) {
            for (int t = 0; t < DE_MAX_TRIES && rr[q] < 0; t++) {
                w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (96 + 64 * q + t),
                        (uint32_t) gen, sw);
                const int cnd = uint_below(w.x, np);
                if (cnd != i && cnd != rr[0] && cnd != rr[1]) rr[q] = cnd;
            }
            for (int cnd = 0; rr[q] < 0; cnd++)
                if (cnd != i && cnd != rr[0] && cnd != rr[1]) rr[q] = cnd;
        }
        r1 = rr[0];
        r2 = rr[1];
        r3 = rr[2];
        w = philox4x32_10(c.seed, (uint32_t) i, 2, (uint32_t) gen, sw);
        jrand = uint_below(w.x, n);
    }
    CR = __shfl(CR, 0, 16);
    F = __shfl(F, 0, 16);
    strat = __shfl(strat, 0, 16);
    r1 = __shfl(r1, 0, 16);
    r2 = __shfl(r2, 0, 16);
    r3 = __shfl(r3, 0, 16);
    jrand = __shfl(jrand, 0, 16);

    if (box_regs) {
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = tid + u * (int) blockDim.x;
            if (j < ld) {
                lob[j] = blo[u];
                upb[j] = bup[u];
            }
        }
    }
    __syncthreads();           // lob / upb are staged
    // (same shape as de_generation: four column pairs per lane, the sixteen gathers first)
    double cnt = 0.;
    double fold = 0.;
    const int npair = ld >> 1;
    double2 akeep[4];
#pragma unroll
    for (int u = 0; u < 4; u++) akeep[u] = make_double2(0., 0.);
    if (live) {
        const double *xi = Xc + (size_t) oi * ld;
        fold = fc[oi];
        const double *x1 = Xc + (size_t) order[r1] * ld;
        const double *x2 = Xc + (size_t) order[r2] * ld;
        const bool rand1 = (strat & 1) == 0;
        // fourth partner: r3 for DE/rand/1, the best individual for DE/current-to-best/2
        const double *x4 = Xc + (size_t) order[rand1 ? r3 : 0] * ld;
        const uint32_t swc = stream_word(STREAM_DE_CROSS, (uint32_t) p);
        for (int pj0 = g; pj0 < npair; pj0 += 64) {
            double2 a[4], q1[4], q2[4], q4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pj = pj0 + 16 * u;
                const int j = pj < npair ? 2 * pj : 0;
                a[u] = *reinterpret_cast<const double2*>(&xi[j]);
                q1[u] = *reinterpret_cast<const double2*>(&x1[j]);
                q2[u] = *reinterpret_cast<const double2*>(&x2[j]);
                q4[u] = *reinterpret_cast<const double2*>(&x4[j]);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int pj = pj0 + 16 * u;
                if (pj < npair) {
                    const int j = 2 * pj;
                    if (pj0 == g) akeep[u] = a[u];
                    const double2 lo = *reinterpret_cast<const double2*>(&lob[j]);
                    const double2 up = *reinterpret_cast<const double2*>(&upb[j]);
                    double2 m;
                    if (rand1) {
                        m.x = q1[u].x + F * (q2[u].x - q4[u].x);
                        m.y = q1[u].y + F * (q2[u].y - q4[u].y);
                    } else {
                        m.x = a[u].x + F * (q4[u].x - a[u].x) + F * (q1[u].x - q2[u].x);
                        m.y = a[u].y + F * (q4[u].y - a[u].y) + F * (q1[u].y - q2[u].y);
                    }
                    const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) pj,
                            (uint32_t) gen, swc);
                    const bool cx = j < n && (u01(w.x, w.y) <= CR || j == jrand);
                    const bool cy = j + 1 < n && (u01(w.z, w.w) <= CR || j + 1 == jrand);
                    double2 v;
                    v.x = cx ? m.x : a[u].x;
                    v.y = cy ? m.y : a[u].y;
                    cnt += (cx ? 1. : 0.) + (cy ? 1. : 0.);
                    const double rlx = (lo.x + a[u].x) / 2., rux = (up.x + a[u].x) / 2.;
                    const double rly = (lo.y + a[u].y) / 2., ruy = (up.y + a[u].y) / 2.;
                    v.x = j < n ? (v.x < lo.x ? rlx : (v.x > up.x ? rux : v.x)) : v.x;
                    v.y = j + 1 < n ? (v.y < lo.y ? rly : (v.y > up.y ? ruy : v.y)) : v.y;
                    *reinterpret_cast<double2*>(&trial[j]) = v;
                }
            }
        }
    }
    __syncthreads();
    cnt = group_sum_d<16>(cnt);
    if (live && g == 0) {
        d.crow[cur ^ 1][pbase + i] = CR;
        d.rec_cr[pbase + i] = c.repaircr ? cnt / n : CR;
        d.rec_f[pbase + i] = F;
        d.slot_of[pbase + i] = strat;
    }
    if (c.obj < 0) {
        // host objective: park the trial in the other buffer; de_select finishes the job
        if (live)
            for (int pj = g; pj < ld / 2; pj += 16)
                *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + 2 * pj]) =
                        *reinterpret_cast<const double2*>(&trial[2 * pj]);
        return;
    }
    double ft = eval_row_group<16>(c.obj, n, trial, d.aux, g);
    if (ft != ft) ft = BBO_INF_D;
    double ssq = 0.;
    bool accept = false;
    if (live) {
        const double *xi = Xc + (size_t) oi * ld;
        accept = ft < fold;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int pj = g + 16 * u;
            if (pj < npair) {
                const int j = 2 * pj;
                const double2 v = accept ? *reinterpret_cast<const double2*>(&trial[j]) : akeep[u];
                *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + j]) = v;
                ssq += v.x * v.x + v.y * v.y;
            }
        }
        for (int pj = g + 64; pj < npair; pj += 16) {
            const int j = 2 * pj;
            const double2 v = accept ? *reinterpret_cast<const double2*>(&trial[j])
                                     : *reinterpret_cast<const double2*>(&xi[j]);
            *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + j]) = v;
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    ssq = group_sum_d<16>(ssq);
    if (live && g == 0) {
        d.f[cur ^ 1][pbase + i] = accept ? ft : fold;
        d.radius[pbase + i] = sqrt(ssq);
        d.rec_df[pbase + i] = fold - ft;
        d.rec_flag[pbase + i] = accept ? 3 : 0;
    }
}

// host-objective path: the trials sit in X[cur^1], their fitness in f[cur^1]; finish the
// selection (same arithmetic as the tail of de_generation)
__global__ __launch_bounds__(256) void de_select(DeDev d, DeConst c)
{
    const int p = blockIdx.y;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * 16 + r;
    const int ld = c.ld, np = sc->np, cur = sc->cur;
    const bool live = i < np;
    const size_t pbase = (size_t) p * c.npinit;
    const double *Xc = d.X[cur] + pbase * ld;
    double *Xn = d.X[cur ^ 1] + pbase * ld;
    double ssq = 0.;
    bool accept = false;
    double fold = 0., ft = 0.;
    if (live) {
        const int row = d.order[pbase + i];
        fold = d.f[cur][pbase + row];
        ft = d.f[cur ^ 1][pbase + i];
        if (ft != ft) ft = BBO_INF_D;
        accept = c.variant == 2 ? ft < fold : ft <= fold;   // SaNSDE replaces on `<` (sansde.cpp:164)
        for (int pj = g; pj < ld / 2; pj += 16) {
            const int j = 2 * pj;
            double2 v = *reinterpret_cast<const double2*>(&Xn[(size_t) i * ld + j]);
            if (!accept) {
                v = *reinterpret_cast<const double2*>(&Xc[(size_t) row * ld + j]);
                *reinterpret_cast<double2*>(&Xn[(size_t) i * ld + j]) = v;
            }
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    ssq = group_sum_d<16>(ssq);
    if (live && g == 0) {
        d.f[cur ^ 1][pbase + i] = accept ? ft : fold;
        d.radius[pbase + i] = sqrt(ssq);
        d.rec_df[pbase + i] = fold - ft;
        d.rec_flag[pbase + i] = (accept ? 1 : 0) | ((accept && ft < fold) ? 2 : 0);
    }
}

// ---------------------------------------------------------------------------
// bookkeeping: archive slots (successes in index order: push while there is room, else a
// drawn slot, the later index wins) and the success-history / adaptive-mean update.
// one workgroup of 1024 threads per population
// ---------------------------------------------------------------------------
__device__ inline double block_sum_1024(double v, double *scratch)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    double s = 0.;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; w++) s += scratch[w];
    return s;
}

__global__ __launch_bounds__(1024) void de_bookkeep(DeDev d, DeConst c)
{
    const int p = blockIdx.x;
    DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    __shared__ int wave_tot[16];
    __shared__ int carry;
    __shared__ double scratch[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = blockDim.x, NWV = (T + 63) >> 6;   // 64 / 256 / 1024 threads by np (launch)
    const int np = sc->np, gen = sc->gen, larch0 = sc->larch;
    const size_t pbase = (size_t) p * c.npinit;
    const int *flag = d.rec_flag + pbase;
    const int mask = c.variant == 0 ? 2 : 1;   // SHADE archives strict improvements only

    // ---- archive slots ------------------------------------------------------------------
    if (c.archive) {
        for (int s = tid; s < np; s += T) d.claim[pbase + s] = -1;
        if (tid == 0) carry = 0;
        __syncthreads();
        for (int base = 0; base < np; base += T) {
            const int i = base + tid;
            const int rec = (i < np && (flag[i] & mask)) ? 1 : 0;
            // exclusive prefix of rec over the block
            const unsigned long long bal = __ballot(rec);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_tot[wave] = __popcll(bal);
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; w++) woff += wave_tot[w];
            const int s_idx = carry + woff + before;
            int slot = -1;
            if (rec) {
                if (larch0 + s_idx < np) {
                    slot = larch0 + s_idx;
                } else {
                    const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, 0, (uint32_t) gen,
                            stream_word(STREAM_DE_ARCH, (uint32_t) p));
                    slot = uint_below(w.x, np);
                }
                atomicMax(&d.claim[pbase + slot], i);
            }
            if (i < np) d.slot_of[pbase + i] = slot;
            __syncthreads();
            if (tid == 0) {
                int tot = 0;
                for (int w = 0; w < NWV; w++) tot += wave_tot[w];
                carry += tot;
            }
            __syncthreads();
        }
    }

    // ---- parameter adaptation --------------------------------------------------------------
    const double *cr = d.rec_cr + pbase, *ff = d.rec_f + pbase, *df = d.rec_df + pbase;
    if (c.variant == 0) {
        // weighted arithmetic mean of CR, weighted Lehmer mean of F (shade.cpp:188-205)
        double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
        int cntl = 0;
        for (int i = tid; i < np; i += T)
            if (flag[i] & 2) {
                const double w = df[i];
                a0 += w * cr[i];
                a1 += w;
                a2 += w * ff[i] * ff[i];
                a3 += w * ff[i];
                cntl++;
            }
        a0 = block_sum_1024(a0, scratch);
        a1 = block_sum_1024(a1, scratch);
        a2 = block_sum_1024(a2, scratch);
        a3 = block_sum_1024(a3, scratch);
        const double nsd = block_sum_1024((double) cntl, scratch);
        if (tid == 0) {
            sc->nsucc = (int) nsd;
            if (nsd > 0.) {
                d.MCR[(size_t) p * c.h + sc->k - 1] = a0 / a1;
                d.MF[(size_t) p * c.h + sc->k - 1] = a2 / a3;
                sc->k = sc->k + 1 > c.h ? 1 : sc->k + 1;
            }
        }
    } else {
        // jade.cpp:180-205: mean / root-mean-square of the successful CR, Lehmer mean of F
        double s1 = 0., s2 = 0., f1 = 0., f2 = 0.;
        int cntl = 0;
        for (int i = tid; i < np; i += T)
            if (flag[i] & 1) {
                s1 += cr[i];
                s2 += cr[i] * cr[i];
                f1 += ff[i];
                f2 += ff[i] * ff[i];
                cntl++;
            }
        s1 = block_sum_1024(s1, scratch);
        s2 = block_sum_1024(s2, scratch);
        f1 = block_sum_1024(f1, scratch);
        f2 = block_sum_1024(f2, scratch);
        const double ns = block_sum_1024((double) cntl, scratch);
        double dev = 0.;
        if (ns > 0.) {
            const double mean = s1 / ns;
            for (int i = tid; i < np; i += T)
                if (flag[i] & 1) dev += (cr[i] - mean) * (cr[i] - mean);
        }
        dev = block_sum_1024(dev, scratch);
        if (tid == 0) {
            sc->nsucc = (int) ns;
            double meancr = 0., meanf = 0.;
            if (ns > 0.) {
                meancr = sqrt(dev / ns) > c.jsigma ? sqrt(s2 / ns) : s1 / ns;
                meanf = (f2 / ns) / (f1 / ns);
            }
            sc->mucr = (1. - c.cdamp) * sc->mucr + c.cdamp * meancr;
            sc->muf = (1. - c.cdamp) * sc->muf + c.cdamp * meanf;
        }
    }
    if (tid == 0) {
        sc->fev += np;
        if (c.archive) {
            int tot = carry;
            sc->larch = min(np, larch0 + tot);
        }
    }
}

// SaNSDE bookkeeping (sansde.cpp:166-183, :213-234): success / failure tallies per strategy
// and per F distribution, the CR record, then the periodic re-estimation of p, crm, fp.
// one workgroup of 1024 threads per population
__global__ __launch_bounds__(1024) void sansde_bookkeep(DeDev d, DeConst c)
{
    const int p = blockIdx.x;
    DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    __shared__ double scratch[16];
    const int tid = threadIdx.x, T = blockDim.x;
    const int np = sc->np;
    const size_t pbase = (size_t) p * c.npinit;
    const int *flag = d.rec_flag + pbase, *strat = d.slot_of + pbase;
    const double *cr = d.rec_cr + pbase, *ff = d.rec_f + pbase, *df = d.rec_df + pbase;
    double ns[2] = { 0., 0. }, nf[2] = { 0., 0. }, fs[2] = { 0., 0. }, ffl[2] = { 0., 0. };
    double rec = 0., del = 0.;
    for (int i = tid; i < np; i += T) {
        const int is = strat[i] & 1, fi = (strat[i] >> 1) & 1;
        if (flag[i] & 1) {
            ns[is] += 1.;
            fs[fi] += ff[i];
            rec += cr[i] * df[i];
            del += df[i];
        } else {
            nf[is] += 1.;
            ffl[fi] += ff[i];
        }
    }
    double tot[10];
    tot[0] = block_sum_1024(ns[0], scratch);
    tot[1] = block_sum_1024(ns[1], scratch);
    tot[2] = block_sum_1024(nf[0], scratch);
    tot[3] = block_sum_1024(nf[1], scratch);
    tot[4] = block_sum_1024(fs[0], scratch);
    tot[5] = block_sum_1024(fs[1], scratch);
    tot[6] = block_sum_1024(ffl[0], scratch);
    tot[7] = block_sum_1024(ffl[1], scratch);
    tot[8] = block_sum_1024(rec, scratch);
    tot[9] = block_sum_1024(del, scratch);
    if (tid == 0) {
        sc->pns[0] += (int) tot[0];
        sc->pns[1] += (int) tot[1];
        sc->pnf[0] += (int) tot[2];
        sc->pnf[1] += (int) tot[3];
        sc->fpns[0] += tot[4];
        sc->fpns[1] += tot[5];
        sc->fpnf[0] += tot[6];
        sc->fpnf[1] += tot[7];
        sc->crrec += tot[8];
        sc->crdeltaf += tot[9];
        sc->nsucc = (int) (tot[0] + tot[1]);
        sc->fev += np;
        const int it = sc->gen + 1;     // de_finish commits gen + 1 afterwards
        if (it % c.npup == 0) {
            const int pnsf0 = sc->pns[0] + sc->pnf[0], pnsf1 = sc->pns[1] + sc->pnf[1];
            sc->sp = (1. * sc->pns[0] * pnsf1) / (sc->pns[1] * pnsf0 + sc->pns[0] * pnsf1);
            sc->pns[0] = sc->pns[1] = sc->pnf[0] = sc->pnf[1] = 0;
        }
        if (it % c.ncrup == 0) {
            if (sc->crdeltaf > 0) sc->crm = sc->crrec / sc->crdeltaf;
            sc->crrec = sc->crdeltaf = 0.;
            const double f0 = sc->fpns[0] + sc->fpnf[0], f1 = sc->fpns[1] + sc->fpnf[1];
            sc->sfp = (1. * sc->fpns[0] * f1) / (sc->fpns[1] * f0 + sc->fpns[0] * f1);
            sc->fpns[0] = sc->fpns[1] = sc->fpnf[0] = sc->fpnf[1] = 0.;
        }
    }
}

// copies the replaced parents into the archive slots they won
// grid (ceil(np_launch/16), P), 256 threads
__global__ __launch_bounds__(256) void de_archive_copy(DeDev d, DeConst c)
{
    const int p = blockIdx.y;
    const DeScal *sc = d.scal + p;
    if (de_frozen(c, sc) || !c.archive) return;
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * 16 + r;
    if (i >= sc->np) return;
    const size_t pbase = (size_t) p * c.npinit;
    const int slot = d.slot_of[pbase + i];
    if (slot < 0 || d.claim[pbase + slot] != i) return;
    const double *src = d.X[sc->cur] + (pbase + d.order[pbase + i]) * c.ld;
    double *dst = d.arch + (pbase + slot) * c.ld;
    for (int pj = g; pj < c.ld / 2; pj += 16)
        *reinterpret_cast<double2*>(&dst[2 * pj]) = *reinterpret_cast<const double2*>(&src[2 * pj]);
}

// ---------------------------------------------------------------------------
// finish: linear population-size reduction, archive trim, stop test, buffer flip.
// Runs after de_rank(next).  One workgroup of 1024 threads per population.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void de_finish(DeDev d, DeConst c)
{
    const int p = blockIdx.x;
    DeScal *sc = d.scal + p;
    if (de_frozen(c, sc)) return;
    __shared__ double scratch[16];
    __shared__ int sh_idx;
    const int tid = threadIdx.x, T = blockDim.x;
    const size_t pbase = (size_t) p * c.npinit;
    int np = sc->np;
    const int gen = sc->gen;
    int larch = sc->larch;
    if (c.variant == 0) {
        const int npnew = (int) round((c.npmin - c.npinit) * ((1. * sc->fev) / c.mfev) + c.npinit);
        if (npnew < np) np = npnew;
        if (c.archive) {
            // swap-with-last removal of drawn slots until the archive fits (shade.cpp:228-235)
            int t = 0;
            while (larch > npnew) {
                if (tid == 0) {
                    const u32x4 w = philox4x32_10(c.seed, (uint32_t) t, 1, (uint32_t) gen,
                            stream_word(STREAM_DE_ARCH, (uint32_t) p));
                    sh_idx = uint_below(w.x, larch);
                }
                __syncthreads();
                const int ir = sh_idx;
                if (ir != larch - 1)
                    for (int j = tid; j < c.ld; j += T)
                        d.arch[(pbase + ir) * c.ld + j] = d.arch[(pbase + larch - 1) * c.ld + j];
                __syncthreads();
                larch--;
                t++;
            }
        }
    }
    // radius spread of the surviving population (two passes)
    const int *order = d.order + pbase;
    const double *rad = d.radius + pbase;
    double s = 0.;
    for (int q = tid; q < np; q += T) s += rad[order[q]];
    const double mean = block_sum_1024(s, scratch) / np;
    double m2 = 0.;
    for (int q = tid; q < np; q += T) {
        const double dd = rad[order[q]] - mean;
        m2 += dd * dd;
    }
    m2 = block_sum_1024(m2, scratch);
    if (tid == 0) {
        sc->np = np;
        sc->larch = larch;
        sc->cur ^= 1;
        sc->gen = gen + 1;
        sc->m2 = m2;
        const int conv = m2 <= (np - 1) * c.tol * c.tol ? 1 : 0;
        sc->conv = conv;
        if (conv) sc->stop = 1;
        else if (sc->fev >= c.mfev) sc->stop = 2;
    }
}

} // namespace bbo
