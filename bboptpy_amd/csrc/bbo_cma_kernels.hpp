// bbo_cma_kernels.hpp -- the CMA-ES generation as hand-written gfx950 kernels.
//
//   kernel                 reference lines it replaces                    bound
//   cma_sample_eval        cmaes.cpp:65-80 + base_cmaes.cpp:214-217        fp64 MFMA (2 n^2 flop/cand)
//   cma_rank               std::sort in base_cmaes.cpp:221                 L2 / VALU (lambda^2 compares)
//   cma_whiten             active_cmaes.cpp:115-132 (ycoeff norms)         fp64 MFMA (2 n^2 flop/cand, worst mu)
//   cma_gram               active_cmaes.cpp:135-161 rank-mu +/- terms,     fp64 MFMA (2 n^2 flop/cand)
//   (cma_gram128s, n = 128) and the weighted mean :75-85
//   cma_paths              active_cmaes.cpp:75-112 + base_cmaes.cpp:176-189 latency (1 workgroup)
//   cma_cov                active_cmaes.cpp:137-160 (assembly of C)        HBM/L2 (slab reduce)
//   cma_eigen              cmaes.cpp:229-478 (tred2 + tql2 + repair)       serial latency (1 workgroup)
//   cma_post               cmaes.cpp:274-282 (C^-1/2) + operand packing    L2 (on demand where lazy_isc)
//   cma_history_stop       base_cmaes.cpp:191-209, cmaes.cpp:151-227       latency
//
// All arithmetic is fp64.  MFMA is v_mfma_f64_16x16x4_f64: A fragment = one double
// per lane, A[row = lane & 15][k = lane >> 4]; B fragment B[k = lane >> 4][col = lane & 15];
// C/D = 4 doubles per lane, col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma once

#include "bbo_cma.hpp"
#include "bbo_objectives.hpp"
#include "bbo_rng.hpp"
#include "bbo_eig.hpp"
#include "bbo_rank.hpp"

namespace bbo {

typedef double d4_t __attribute__((ext_vector_type(4)));

#define BBO_INF (__builtin_huge_val())

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// hand-over point between the lanes of ONE wavefront (single-wavefront bodies: the hardware
// keeps a wavefront's memory operations in order, the fence stops the compiler from moving
// loads across the point)
__device__ inline void cma_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ inline bool pop_frozen(const CmaConst &c, const CmaScal *sc)
{
    return c.honor_stop && sc->stop != 0;
}

// The four normals Philox call q of candidate `row` delivers: columns cma_quad_col0(q) + 4 i,
// i = 0..3 -- one lane's A-fragment elements of four consecutive k-steps.  Injected Z
// (parity tests) and the Z recorder go through the same mapping.  tab: normal_table_fill'ed.
__device__ inline void cma_draw_quad(const CmaDev &d, const CmaConst &c, int p, int row, int q,
        int gen, uint32_t sw, const double2 *tab, double (&z)[4])
{
    const int j0 = cma_quad_col0(q);
    z[0] = z[1] = z[2] = z[3] = 0.;
    if (row < c.lambda && j0 < c.n) {
        if (d.zinject) {
            const double *zi = d.zinject + ((size_t) p * c.lambda + row) * c.n;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (j0 + 4 * i < c.n) z[i] = zi[j0 + 4 * i];
        } else {
            normal_quad(c.seed, (uint32_t) row, (uint32_t) q, (uint32_t) gen, sw, tab, z[0], z[1],
                    z[2], z[3]);
#pragma unroll
            for (int i = 1; i < 4; i++)
                if (j0 + 4 * i >= c.n) z[i] = 0.;
        }
        if (d.zrecord) {
            double *zr = d.zrecord + ((size_t) p * c.lambda + row) * c.n;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (j0 + 4 * i < c.n) zr[j0 + 4 * i] = z[i];
        }
    }
}

// The same in two steps (normal_quad_fast / normal_quad_settle): the candidates of call q and
// the mask of the draws still to settle -- the caller collects the masks of all the calls it
// holds and settles them together, cma_settle_draw per set bit.
__device__ inline uint32_t cma_draw_quad_fast(const CmaDev &d, const CmaConst &c, int p, int row,
        int q, int gen, uint32_t sw, const double2 *tab, double (&z)[4])
{
    const int j0 = cma_quad_col0(q);
    uint32_t pend = 0;
    z[0] = z[1] = z[2] = z[3] = 0.;
    if (row < c.lambda && j0 < c.n) {
        if (d.zinject) {
            const double *zi = d.zinject + ((size_t) p * c.lambda + row) * c.n;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (j0 + 4 * i < c.n) z[i] = zi[j0 + 4 * i];
        } else {
            pend = normal_quad_fast(c.seed, (uint32_t) row, (uint32_t) q, (uint32_t) gen, sw, tab,
                    z[0], z[1], z[2], z[3]);
#pragma unroll
            for (int i = 1; i < 4; i++)
                if (j0 + 4 * i >= c.n) {
                    z[i] = 0.;
                    pend &= ~(1u << i);
                }
        }
        if (d.zrecord) {
            double *zr = d.zrecord + ((size_t) p * c.lambda + row) * c.n;
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (j0 + 4 * i < c.n) zr[j0 + 4 * i] = z[i];
        }
    }
    return pend;
}

__device__ inline double cma_settle_draw(const CmaDev &d, const CmaConst &c, int p, int row, int q,
        int slot, int gen, uint32_t sw, const double2 *tab)
{
    const double v = normal_quad_settle(c.seed, (uint32_t) row, (uint32_t) q, (uint32_t) slot,
            (uint32_t) gen, sw, tab, zig_global_f());
    if (d.zrecord)
        d.zrecord[((size_t) p * c.lambda + row) * c.n + cma_quad_col0(q) + 4 * slot] = v;
    return v;
}

// ---------------------------------------------------------------------------
// sample + evaluate: X = m + sigma * Z (B diag D)^T, f = objective(X)
// grid (lambda_pad/16, P), 64 NW threads; dynamic LDS 16*(ld+2) doubles
// NW wavefronts, MAXT column tiles each (NW * MAXT >= ld / 16).  NW = 4 is the form for many row
// tiles; with a handful of candidates (C5: lambda = 20 at n = 256, two workgroups on the whole
// chip) the sweep is a chain of L2 round trips for the operand, one per group of k-steps and
// wavefront, and NW = ld / 16 wavefronts of ONE column tile each put four times the requests in
// flight (round 4: 60 -> 3x us).  Same products in the same order for every NW.
// ---------------------------------------------------------------------------
template<int MAXT, int NW = 4>
__global__ __launch_bounds__(64 * NW) void cma_sample_eval(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y, mt = blockIdx.x;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = c.ld, ldz = ld + 2;
    const int gen = sc->it;

    // 1. the standard normals of these 16 candidates (four per Philox call)
    __shared__ double2 ntab[NORMAL_TABLE_N];
    normal_table_fill(ntab, tid, 64 * NW);
    __syncthreads();
    const int nquads = ld >> 2;
    const uint32_t sw = stream_word(STREAM_CMA_NORMAL, (uint32_t) p);
    uint64_t pend = 0;                       // 4 bits per call this thread draws (<= 16 calls)
    for (int qi = tid, it = 0; qi < 16 * nquads; qi += 64 * NW, it++) {
        const int r = qi / nquads, q = qi - r * nquads;
        double z[4];
        pend |= (uint64_t) cma_draw_quad_fast(d, c, p, mt * 16 + r, q, gen, sw, ntab, z) << (4 * it);
        const int j0 = cma_quad_col0(q);
#pragma unroll
        for (int i = 0; i < 4; i++) lds[r * ldz + j0 + 4 * i] = z[i];
    }
    while (pend) {
        const int b = __ffsll((unsigned long long) pend) - 1;
        pend &= pend - 1;
        const int qi = tid + 64 * NW * (b >> 2), r = qi / nquads, q = qi - r * nquads;
        lds[r * ldz + cma_quad_col0(q) + 4 * (b & 3)] =
                cma_settle_draw(d, c, p, mt * 16 + r, q, b & 3, gen, sw, ntab);
    }
    __syncthreads();

    // 2. 16 x ld tile of Z (B D)^T on the matrix cores; wave w owns column tiles w, w + NW, ...
    d4_t acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) acc[t] = d4_t { 0., 0., 0., 0. };
    const int NT = ld >> 4, KS = ld >> 2;
    const double *bdp = d.BDp + (size_t) p * ld * ld;
    const int ar = lane & 15, ak = lane >> 4;
    double zz = 0.;
    // four k-steps of operand fragments requested before the first of their MFMAs: with one
    // 16-row tile per workgroup (small lambda: a handful of workgroups on the whole chip) the
    // fragments come from L2 one dependent round trip per k-step otherwise.  (ld is a multiple
    // of 16: KS is a multiple of 4.)  Same products in the same order.
    for (int ks0 = 0; ks0 < KS; ks0 += 4) {
        double a[4], b[4][MAXT];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            a[u] = lds[ar * ldz + 4 * (ks0 + u) + ak];
#pragma unroll
            for (int t = 0; t < MAXT; t++) {
                const int nt = wave + NW * t;
                b[u][t] = nt < NT ? bdp[((size_t) nt * KS + ks0 + u) * 64 + lane] : 0.;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            zz = __builtin_fma(a[u], a[u], zz);
#pragma unroll
            for (int t = 0; t < MAXT; t++) {
                const int nt = wave + NW * t;
                if (nt < NT)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][t], acc[t], 0, 0, 0);
            }
        }
    }
    // ||z||^2 of the 16 rows of this tile (by-product for the whitened-norm shortcut, as in
    // sample_eval64_body: same sums in the same order)
    zz += __shfl_xor(zz, 16, 64);
    zz += __shfl_xor(zz, 32, 64);
    if (wave == 0 && ak == 0) d.zn2[(size_t) p * c.lambda_pad + mt * 16 + ar] = zz;
    __syncthreads();

    // 3. x = m + sigma * y (clipped to the box when bound), to HBM and to LDS
    const double sigma = sc->sigma;
    const double *xm = d.xmean + (size_t) p * ld;
    double *Xp = d.X + (size_t) p * c.lambda_pad * ld;
    // (the lane's columns of the mean and of the box first, all of them: read next to their use
    // they sit between the row stores, and each is then waited for together with the stores in
    // front of it -- one in-order memory counter)
    double xmc[MAXT], loc[MAXT], upc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int col = min((wave + NW * t) * 16 + (lane & 15), ld - 1);
        xmc[t] = xm[col];
        loc[t] = c.bound ? d.lower[col] : 0.;
        upc[t] = c.bound ? d.upper[col] : 0.;
    }
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int nt = wave + NW * t;
        if (nt < NT) {
            const int col = nt * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rl = (lane >> 4) + 4 * r;
                double v = 0.;
                if (col < c.n) {
                    v = xmc[t] + sigma * acc[t][r];
                    if (c.bound) v = fmax(loc[t], fmin(v, upc[t]));
                }
                lds[rl * ldz + col] = v;
                Xp[((size_t) mt * 16 + rl) * ld + col] = v;
            }
        }
    }
    __syncthreads();

    // 4. objective: 16 lanes per candidate
    if (c.obj >= 0 && tid < 256) {
        const int r = tid >> 4, g = tid & 15;
        double f = eval_row_group<16>(c.obj, c.n, &lds[r * ldz], d.aux, g);
        if (g == 0) {
            const int row = mt * 16 + r;
            if (!(row < c.lambda) || f != f) f = BBO_INF;   // padding rows; NaN -> +inf
            d.f[(size_t) p * c.lambda_pad + row] = f;
        }
    }
}

// ---------------------------------------------------------------------------
// sample + evaluate for ld <= 128: 64 candidates per workgroup.  A wavefront loads the
// (B D) fragments of its column tiles ONCE into registers and sweeps four 16-row tiles of
// normals with them, so the packed operand is read once per 64 candidates instead of once
// per 16.  grid (ceil(lambda_pad/64), P), 256 threads; dynamic LDS 64*(ld+2) doubles
// ---------------------------------------------------------------------------
template<int MAXT, int KSM = 32>     // KSM: k-steps held (ld <= 4 KSM); 8 keeps ld <= 32 lean in registers
__device__ __forceinline__ void sample_eval64_body(const CmaDev &d, const CmaConst &c, int p, int bx,
        double *lds, int psub,      // psub: the population's Philox sub-stream (= p unless d is a local view)
        bool stage_table = true)    // false: an earlier call of this workgroup staged the generator's table
{
    const int row0 = bx * 64;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = c.ld, ldz = ld + 2;
    const int gen = sc->it;
    const int NT = ld >> 4, KS = ld >> 2;

    // (B D) fragments of this wavefront's column tiles
    const double *bdp = d.BDp + (size_t) p * ld * ld;
    double bfr[MAXT][KSM];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int nt = wave + 4 * t;
#pragma unroll
        for (int ks = 0; ks < KSM; ks++)
            bfr[t][ks] = (nt < NT && ks < KS) ? bdp[((size_t) nt * KS + ks) * 64 + lane] : 0.;
    }

    // standard normals of the 64 candidates (four per Philox call)
    __shared__ double2 ntab[NORMAL_TABLE_N];
    if (stage_table) {
        normal_table_fill(ntab, tid, 256);
        __syncthreads();
    }
    const int nquads = ld >> 2;
    const uint32_t sw = stream_word(STREAM_CMA_NORMAL, (uint32_t) psub);
    uint32_t pend = 0;                       // 4 bits per call this thread draws (<= 8 calls)
    for (int qi = tid, it = 0; qi < 64 * nquads; qi += 256, it++) {
        const int r = qi / nquads, q = qi - r * nquads;
        double z[4];
        pend |= cma_draw_quad_fast(d, c, p, row0 + r, q, gen, sw, ntab, z) << (4 * it);
        const int j0 = cma_quad_col0(q);
#pragma unroll
        for (int i = 0; i < 4; i++) lds[r * ldz + j0 + 4 * i] = z[i];
    }
    while (pend) {
        const int b = __ffs(pend) - 1;
        pend &= pend - 1;
        const int qi = tid + 256 * (b >> 2), r = qi / nquads, q = qi - r * nquads;
        lds[r * ldz + cma_quad_col0(q) + 4 * (b & 3)] =
                cma_settle_draw(d, c, p, row0 + r, q, b & 3, gen, sw, ntab);
    }
    __syncthreads();

    d4_t acc[4][MAXT];
    const int ar = lane & 15, ak = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
        double zz = 0.;
#pragma unroll
        for (int t = 0; t < MAXT; t++) acc[mt][t] = d4_t { 0., 0., 0., 0. };
#pragma unroll
        for (int ks = 0; ks < KSM; ks++) {
            if (ks < KS) {
                const double a = lds[(mt * 16 + ar) * ldz + 4 * ks + ak];
                zz = __builtin_fma(a, a, zz);
#pragma unroll
                for (int t = 0; t < MAXT; t++)
                    acc[mt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bfr[t][ks], acc[mt][t], 0, 0, 0);
            }
        }
        // ||z||^2 of the 16 rows of this tile (by-product for the whitened-norm shortcut)
        zz += __shfl_xor(zz, 16, 64);
        zz += __shfl_xor(zz, 32, 64);
        if (wave == 0 && ak == 0 && row0 + mt * 16 + ar < c.lambda_pad)
            d.zn2[(size_t) p * c.lambda_pad + row0 + mt * 16 + ar] = zz;
    }
    __syncthreads();

    const double sigma = sc->sigma;
    const double *xm = d.xmean + (size_t) p * ld;
    double *Xp = d.X + (size_t) p * c.lambda_pad * ld;
    // (the lane's columns of the mean and of the box first: see sample_eval_body)
    double xmc[MAXT], loc[MAXT], upc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        const int col = min((wave + 4 * t) * 16 + (lane & 15), ld - 1);
        xmc[t] = xm[col];
        loc[t] = c.bound ? d.lower[col] : 0.;
        upc[t] = c.bound ? d.upper[col] : 0.;
    }
#pragma unroll
    for (int mt = 0; mt < 4; mt++) {
#pragma unroll
        for (int t = 0; t < MAXT; t++) {
            const int nt = wave + 4 * t;
            if (nt < NT) {
                const int col = nt * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int rl = mt * 16 + (lane >> 4) + 4 * r;
                    double v = 0.;
                    if (col < c.n) {
                        v = xmc[t] + sigma * acc[mt][t][r];
                        if (c.bound) v = fmax(loc[t], fmin(v, upc[t]));
                    }
                    lds[rl * ldz + col] = v;
                    if (row0 + rl < c.lambda_pad) Xp[((size_t) row0 + rl) * ld + col] = v;
                }
            }
        }
    }
    __syncthreads();

    if (c.obj >= 0) {
        const int g = tid & 15;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int r = (tid >> 4) + 16 * rr;
            double f = eval_row_group<16>(c.obj, c.n, &lds[r * ldz], d.aux, g);
            const int row = row0 + r;
            if (g == 0 && row < c.lambda_pad) {
                if (!(row < c.lambda) || f != f) f = BBO_INF;
                d.f[(size_t) p * c.lambda_pad + row] = f;
            }
        }
    }
}

template<int MAXT, int KSM = 32>
__global__ __launch_bounds__(256) void cma_sample_eval64(CmaDev d, CmaConst c)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    sample_eval64_body<MAXT, KSM>(d, c, blockIdx.y, blockIdx.x, lds, blockIdx.y);
}

// ---------------------------------------------------------------------------
// sample + evaluate for ld == 128 (the headline shape), whole populations in flight:
// the packed (B D) operand of the population sits in LDS (128 KB, loaded once per
// workgroup), each wavefront owns whole 16-candidate tiles: it DRAWS the normals straight
// into its MFMA A fragments (the Philox column layout is the fragment layout, see
// cma_quad_col0), sweeps the 8 column tiles, and evaluates the objective on the
// accumulators.  No LDS traffic besides the B fragments, no barrier after the fill.
// grid (ceil(lambda_pad / rows_per_wg), P), 512 threads, dynamic LDS 128 KB
// ---------------------------------------------------------------------------
// FULL: the dimension is exactly 128, no box, lambda a multiple of 16, no injected / recorded
// normals (the benchmark's M and C3): every bounds test, clamp and per-store branch of the general
// form is gone at compile time -- ~400 of the ~3100 vector instructions a 16-row tile costs next
// to its 256 MFMAs, all on the same pipe.  Same arithmetic, same bits.
template<bool FULL>
__device__ inline void sample128_epilogue(const CmaDev &d, const CmaConst &c, int p, int rowbase,
        double sigma, const d4_t (&acc)[8], int lane, const double *xm)
{
    // xm: the mean, from the workgroup's LDS copy.  Read from global memory here -- eight loads,
    // each next to its use between two groups of four row stores -- every one of them was waited
    // for with vmcnt(0), i.e. together with the stores in front of it (one in-order counter): eight
    // store acknowledgements per tile on the wavefront's critical path (round 3, from the ISA).
    const int fr = lane & 15, fk = lane >> 4;
    const int n = FULL ? 128 : c.n;
    double *Xp = d.X + (size_t) p * c.lambda_pad * 128 + ((size_t) rowbase + fk) * 128 + fr;
    double x[8][4];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int col = t * 16 + fr;
        const double xmc = xm[col];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = 0.;
            if (FULL) {
                v = xmc + sigma * acc[t][r];
            } else if (col < n) {
                v = xmc + sigma * acc[t][r];
                if (c.bound) v = fmax(d.lower[col], fmin(v, d.upper[col]));
            }
            x[t][r] = v;
            Xp[(size_t) (4 * r) * 128 + t * 16] = v;
        }
    }
    if (c.obj >= 0) {
        double f[4];
        eval_frag_rows<8>(c.obj, n, x, d.aux, lane, f);
        if (fr == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rr = rowbase + fk + 4 * r;
                double fv = f[r];
                if ((!FULL && !(rr < c.lambda)) || fv != fv) fv = BBO_INF;
                d.f[(size_t) p * c.lambda_pad + rr] = fv;
            }
        }
    }
}

template<bool FULL>
__device__ __forceinline__ void sample_eval128_body(const CmaDev &d, const CmaConst &c,
        int rows_per_wg, double *bd, const double2 *ntab, const double *ftab, const double *xms)
{
    const int p = blockIdx.y, row0 = blockIdx.x * rows_per_wg;
    const CmaScal *sc = d.scal + p;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (fp64 MFMA and vector ALU work of any kind do not overlap on gfx950 -- measured, DESIGN.md
    // section 3 -- so the draw is priced in VALU cycles next to the sweep, not hidden behind it.)
    const int gen = sc->it;
    const double sigma = sc->sigma;
    const int tiles = min(rows_per_wg, c.lambda_pad - row0) >> 4;
    const int fr = lane & 15, fk = lane >> 4;
    const uint32_t sw = stream_word(STREAM_CMA_NORMAL, (uint32_t) p);

    for (int tile = wave; tile < tiles; tile += 8) {
        const int rowbase = row0 + tile * 16;
        const int row = rowbase + fr;
        d4_t acc[8];
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = d4_t { 0., 0., 0., 0. };
        double zz = 0.;
        if (FULL) {
            // the 32 normals this lane feeds: z[4 q + i] = Z[row][16 q + fk + 4 i], i.e. the A
            // elements of k-steps 4 q .. 4 q + 3; candidates first, the unsettled draws of all
            // eight calls together afterwards (normal_quad_fast)
            double z[32];
            uint32_t pend = normal_quads_fast<8>(c.seed, (uint32_t) row, (uint32_t) fk, 4u,
                    (uint32_t) gen, sw, ntab, z);
            while (pend) {
                const int b = __ffs(pend) - 1;
                pend &= pend - 1;
                const double v = normal_quad_settle(c.seed, (uint32_t) row,
                        (uint32_t) (4 * (b >> 2) + fk), (uint32_t) (b & 3), (uint32_t) gen, sw,
                        ntab, ftab);
#pragma unroll
                for (int i = 0; i < 32; i++) z[i] = (i == b) ? v : z[i];
            }
#pragma unroll
            for (int i = 0; i < 32; i++) zz = __builtin_fma(z[i], z[i], zz);
            // (two bases, 64 KB apart: every operand read then is base + a 16-bit immediate; from
            // one base the upper half costs a vector add per read)
            int hi = lane + 8192;
            asm volatile("" : "+v"(hi));          // (opaque: keeps the second base in its register)
            auto frag = [&](int i, int t) {
                const int idx = (i >> 3) * 8 + t * 32 + (i & 7);          // fragment number
                return idx < 128 ? bd[lane + idx * 64] : bd[hi + (idx - 128) * 64];
            };
            // The scheduling barrier every four k-steps (it keeps the scheduler from hoisting the
            // operand reads of all 32 k-steps at once) also makes every group START with its reads:
            // its first MFMA waited for a fresh LDS round trip, eight times per tile.  The eight
            // fragments of a group's first k-step are therefore read at the end of the group before.
            double pre[8];
#pragma unroll
            for (int t = 0; t < 8; t++) pre[t] = frag(0, t);
#pragma unroll
            for (int i = 0; i < 32; i++) {
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const double bv = (i & 3) == 0 ? pre[t] : frag(i, t);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(z[i], bv, acc[t], 0, 0, 0);
                }
                if ((i & 3) == 3) {
                    if (i + 1 < 32) {
#pragma unroll
                        for (int t = 0; t < 8; t++) pre[t] = frag(i + 1, t);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // eight k-steps at a time: draw a[i] = z[row][4 (8 kc + i) + fk] (two Philox calls),
            // then sweep them
#pragma unroll 1
            for (int kc = 0; kc < 4; kc++) {
                double a[8];
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    double z[4];
                    cma_draw_quad(d, c, p, row, 4 * (2 * kc + h) + fk, gen, sw, ntab, z);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        a[4 * h + i] = z[i];
                        zz = __builtin_fma(z[i], z[i], zz);
                    }
                }
                const double *bk = bd + kc * 8 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 8; i++) {
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bk[(t * 32 + i) * 64],
                                acc[t], 0, 0, 0);
                }
            }
        }
        // ||z||^2 of the row: the four k-groups of a row sit 16 lanes apart
        zz += __shfl_xor(zz, 16, 64);
        zz += __shfl_xor(zz, 32, 64);
        if (fk == 0) d.zn2[(size_t) p * c.lambda_pad + row] = zz;
        sample128_epilogue<FULL>(d, c, p, rowbase, sigma, acc, lane, xms);
    }
}

__global__ __launch_bounds__(512, 1) void cma_sample_eval128(CmaDev d, CmaConst c, int rows_per_wg,
        int full)
{
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double bd[];
    const int tid = threadIdx.x;
    {
        const double2 *src = reinterpret_cast<const double2*>(d.BDp + (size_t) p * 128 * 128);
        double2 *dst = reinterpret_cast<double2*>(bd);
#pragma unroll
        for (int i = 0; i < 16; i++) dst[tid + 512 * i] = src[tid + 512 * i];
    }
    __shared__ double2 ntab[NORMAL_TABLE_N];
    __shared__ double ftab[NORMAL_FTABLE_N];
    __shared__ double xms[128];
    normal_table_fill(ntab, tid, 512);
    normal_ftable_fill(ftab, tid, 512);
    if (tid < 128) xms[tid] = d.xmean[(size_t) p * 128 + tid];
    __syncthreads();
    if (full) sample_eval128_body<true>(d, c, rows_per_wg, bd, ntab, ftab, xms);
    else sample_eval128_body<false>(d, c, rows_per_wg, bd, ntab, ftab, xms);
}

// ---------------------------------------------------------------------------
// rank: rank[i] = #{ j : f_j < f_i  or (f_j == f_i and j < i) }, order[rank[i]] = i
// grid (ceil(lambda SL / 256), P), 256 threads = 256 / SL candidates x SL slices of the population
// Active CMA-ES with unclamped samples and a consistent basis: the whitened norms of the worst mu
// are the sampler's sigma^2 ||z||^2 (cma_whiten128's shortcut) and leave with the ranking -- every
// candidate knows its rank here -- instead of through a launch of cma_whiten (round 5: 5 us of a
// single-population generation)
// ---------------------------------------------------------------------------
template<int SL>
__device__ __forceinline__ void cma_rank_body(const CmaDev &d, const CmaConst &c)
{
    const int p = blockIdx.y;
    CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    __shared__ __attribute__((aligned(16))) double tile[RANK_TILE];
    const int tid = threadIdx.x;
    const int cand = blockIdx.x * (256 / SL) + tid / SL, slice = tid % SL;
    const double *f = d.f + (size_t) p * c.lambda_pad;
    const bool live = cand < c.lambda;
    const int cnt = rank_by_counting<SL>(f, c.lambda, cand, slice, tile);
    const bool norms = c.variant == 1 && c.use_zn && sc->basis_ok;
    if (live && slice == 0) {
        const double fi = f[cand];
        d.rank[(size_t) p * c.lambda_pad + cand] = cnt;
        d.order[(size_t) p * c.lambda_pad + cnt] = cand;
        if (cnt == 0) { sc->ibw[0] = cand; sc->ybw[0] = fi; }
        if (cnt == 1) { sc->ibw[1] = cand; sc->ybw[1] = fi; }
        if (cnt == c.lambda - 2) { sc->ibw[2] = cand; sc->ybw[2] = fi; }
        if (cnt == c.lambda - 1) { sc->ibw[3] = cand; sc->ybw[3] = fi; }
        if (norms && cnt >= c.lambda - c.mu)
            d.S[(size_t) p * c.mu_pad + cnt - (c.lambda - c.mu)] =
                    sc->sigma * sc->sigma * d.zn2[(size_t) p * c.lambda_pad + cand];
    }
    if (norms && blockIdx.x == 0)
        for (int wr = c.mu + tid; wr < c.mu_pad; wr += 256) d.S[(size_t) p * c.mu_pad + wr] = 0.;
    if (blockIdx.x == 0 && tid == 0) sc->fev += c.lambda;   // base_cmaes.cpp:218
}

__global__ __launch_bounds__(256) void cma_rank(CmaDev d, CmaConst c)
{
    cma_rank_body<8>(d, c);
}
__global__ __launch_bounds__(256) void cma_rank32(CmaDev d, CmaConst c)
{
    cma_rank_body<32>(d, c);
}
__global__ __launch_bounds__(256) void cma_rank64(CmaDev d, CmaConst c)
{
    cma_rank_body<64>(d, c);
}

// the same ranking by one in-LDS sort per population (lambda <= SORT_LDS_MAX): merge sort by
// merge path for 2048 / 4096 padded keys (`merge`: two buffers, 24 m bytes of dynamic LDS), the
// bitonic network otherwise (12 max(m, 1024) bytes).  grid (P), 1024 threads (256 for m <= 256)
__host__ __device__ inline bool rank_sort_merges(int m, int dbg)
{
    return (m == 2048 || m == 4096) && !(dbg & 262144);
}

__global__ __launch_bounds__(1024) void cma_rank_sort(CmaDev d, CmaConst c, int m)
{
    const int p = blockIdx.x;
    CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double sortbuf[];
    double *keys = sortbuf;
    int *idx = reinterpret_cast<int*>(sortbuf + max(m, 1024));   // the sort pads to >= 1024
    const double *f = d.f + (size_t) p * c.lambda_pad;
    int *order = d.order + (size_t) p * c.lambda_pad, *rank = d.rank + (size_t) p * c.lambda_pad;
    if (rank_sort_merges(m, d.dbg)) {
        int *ibuf = reinterpret_cast<int*>(sortbuf + 2 * m);
        if (m == 2048) merge_sort_lds<2>(f, c.lambda, sortbuf, ibuf, order, rank, &keys, &idx);
        else merge_sort_lds<4>(f, c.lambda, sortbuf, ibuf, order, rank, &keys, &idx);
    } else
    bitonic_sort_lds(f, c.lambda, m, keys, idx, order, rank);
    if (c.variant == 1 && c.use_zn && sc->basis_ok) {
        // active CMA-ES, unclamped samples, consistent basis: || C^-1/2 (x - m) ||^2 of the worst
        // mu is sigma^2 ||z||^2 (cma_whiten128), gathered through the ranking that sits in LDS here
        const double s2 = sc->sigma * sc->sigma;
        for (int wr = threadIdx.x; wr < c.mu_pad; wr += blockDim.x)
            d.S[(size_t) p * c.mu_pad + wr] = wr < c.mu
                    ? s2 * d.zn2[(size_t) p * c.lambda_pad + idx[c.lambda - c.mu + wr]] : 0.;
    }
    if (threadIdx.x == 0) {
        const int L = c.lambda;
        sc->ibw[0] = idx[0]; sc->ybw[0] = keys[0];
        sc->ibw[1] = idx[1]; sc->ybw[1] = keys[1];
        sc->ibw[2] = idx[L - 2]; sc->ybw[2] = keys[L - 2];
        sc->ibw[3] = idx[L - 1]; sc->ybw[3] = keys[L - 1];
        sc->fev += L;   // base_cmaes.cpp:218
    }
}

// the same ranking for lambda <= 64: one WAVEFRONT per population, the bitonic network entirely
// in registers and lane exchanges (no LDS, no barrier); four populations per workgroup.
// grid (ceil(P / 4)), 256 threads
__device__ __forceinline__ void rank_wave_body(const CmaDev &d, const CmaConst &c, int p, int lane)
{
    CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const double *f = d.f + (size_t) p * c.lambda_pad;
    double kf[1] = { lane < c.lambda ? f[lane] : __builtin_huge_val() };
    int ki[1] = { lane < c.lambda ? lane : 0x7fffffff };
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
    const int L = c.lambda;
    if (lane < L) {
        d.order[(size_t) p * c.lambda_pad + lane] = ki[0];
        d.rank[(size_t) p * c.lambda_pad + ki[0]] = lane;
        if (lane == 0) { sc->ibw[0] = ki[0]; sc->ybw[0] = kf[0]; }
        if (lane == 1) { sc->ibw[1] = ki[0]; sc->ybw[1] = kf[0]; }
        if (lane == L - 2) { sc->ibw[2] = ki[0]; sc->ybw[2] = kf[0]; }
        if (lane == L - 1) { sc->ibw[3] = ki[0]; sc->ybw[3] = kf[0]; }
    }
    if (lane == 0) sc->fev += L;   // base_cmaes.cpp:218
}

__global__ __launch_bounds__(256) void cma_rank_wave(CmaDev d, CmaConst c)
{
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= c.npop) return;
    rank_wave_body(d, c, p, lane);
}

// ---------------------------------------------------------------------------
// whiten: S[r] = || C^-1/2 (x_{(lambda-mu+r):lambda} - xold) ||^2 for the worst mu
// grid (mu_pad/16, P), 256 threads; dynamic LDS 16*(ld+2) doubles + 64
// ---------------------------------------------------------------------------
template<int MAXT>
__device__ __forceinline__ void whiten_body(const CmaDev &d, const CmaConst &c, int p, int mt,
        double *lds)
{
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = c.ld, ldz = ld + 2;
    if (c.use_zn && sc->basis_ok) {
        // sigma^2 ||z||^2 from the sampler stands in for the GEMM (see cma_whiten128)
        if (tid < 16) {
            const int wr = mt * 16 + tid;
            if (wr < c.mu_pad)
                d.S[(size_t) p * c.mu_pad + wr] = wr < c.mu
                        ? sc->sigma * sc->sigma * d.zn2[(size_t) p * c.lambda_pad
                                + d.order[(size_t) p * c.lambda_pad + c.lambda - c.mu + wr]]
                        : 0.;
        }
        return;
    }
    double *part = lds + 16 * ldz;   // [4][16]
    const double *Xp = d.X + (size_t) p * c.lambda_pad * ld;
    const double *xold = d.xmean + (size_t) p * ld;   // the mean has not moved yet
    const int *order = d.order + (size_t) p * c.lambda_pad;

    for (int q = tid; q < 16 * ld; q += 256) {
        const int r = q / ld, j = q - r * ld;
        const int wr = mt * 16 + r;
        double v = 0.;
        if (wr < c.mu && j < c.n) {
            const int cand = order[c.lambda - c.mu + wr];
            v = Xp[(size_t) cand * ld + j] - xold[j];
        }
        lds[r * ldz + j] = v;
    }
    __syncthreads();

    d4_t acc[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; t++) acc[t] = d4_t { 0., 0., 0., 0. };
    const int NT = ld >> 4, KS = ld >> 2;
    const double *isp = d.ISp + (size_t) p * ld * ld;
    const int ar = lane & 15, ak = lane >> 4;
    for (int ks = 0; ks < KS; ks++) {
        const double a = lds[ar * ldz + 4 * ks + ak];
#pragma unroll
        for (int t = 0; t < MAXT; t++) {
            const int nt = wave + 4 * t;
            if (nt < NT) {
                const double b = isp[((size_t) nt * KS + ks) * 64 + lane];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // row sums of squares: lane holds rows (lane>>4)+4r, one column per tile
    double ss[4] = { 0., 0., 0., 0. };
#pragma unroll
    for (int t = 0; t < MAXT; t++) {
        if (wave + 4 * t < NT) {
#pragma unroll
            for (int r = 0; r < 4; r++) ss[r] += acc[t][r] * acc[t][r];
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) ss[r] += __shfl_xor(ss[r], off, 16);
    }
    if ((lane & 15) == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) part[wave * 16 + (lane >> 4) + 4 * r] = ss[r];
    }
    __syncthreads();
    if (tid < 16) {
        const int wr = mt * 16 + tid;
        if (wr < c.mu_pad)
            d.S[(size_t) p * c.mu_pad + wr] = part[tid] + part[16 + tid] + part[32 + tid]
                    + part[48 + tid];
    }
}

template<int MAXT>
__global__ __launch_bounds__(256) void cma_whiten(CmaDev d, CmaConst c)
{
    extern __shared__ double lds[];
    whiten_body<MAXT>(d, c, blockIdx.y, blockIdx.x, lds);
}

// ---------------------------------------------------------------------------
// whiten for ld == 128, whole populations in flight: the packed C^-1/2 operand sits in LDS
// (loaded once per workgroup), a wavefront owns whole 16-row tiles of the mu worst
// candidates, gathers y = x - xold straight into its MFMA A fragments and reduces the
// squared norms on the accumulators.  grid (ceil(mu_pad / rows_per_wg), P), 512 threads,
// dynamic LDS 128 KB + 1 KB
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void cma_whiten128(CmaDev d, CmaConst c, int rows_per_wg)
{
    const int p = blockIdx.y, wr0 = blockIdx.x * rows_per_wg;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double is[];
    double *xo = is + 128 * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (c.use_zn && sc->basis_ok) {
        // x - m = sigma B D z and C^-1/2 = B D^-1 B^T from the same (B, D), so
        // C^-1/2 (x - m) = sigma B z and the squared norm is sigma^2 ||z||^2: the sampler's own
        // by-product.  (Not so for clamped x, nor before the first decomposition of a
        // re-initialised object: those take the GEMM below.)
        const double s2 = sc->sigma * sc->sigma;
        const int *ord = d.order + (size_t) p * c.lambda_pad;
        const int rows = min(rows_per_wg, c.mu_pad - wr0);
        for (int r = tid; r < rows; r += 512) {
            const int wr = wr0 + r;
            d.S[(size_t) p * c.mu_pad + wr] = wr < c.mu
                    ? s2 * d.zn2[(size_t) p * c.lambda_pad + ord[c.lambda - c.mu + wr]] : 0.;
        }
        return;
    }
    {
        const double2 *src = reinterpret_cast<const double2*>(d.ISp + (size_t) p * 128 * 128);
        double2 *dst = reinterpret_cast<double2*>(is);
#pragma unroll
        for (int i = 0; i < 16; i++) dst[tid + 512 * i] = src[tid + 512 * i];
        if (tid < 128) xo[tid] = d.xmean[(size_t) p * 128 + tid];   // the mean has not moved yet
    }
    __syncthreads();
    const double *Xp = d.X + (size_t) p * c.lambda_pad * 128;
    const int *order = d.order + (size_t) p * c.lambda_pad;
    const int tiles = min(rows_per_wg, c.mu_pad - wr0) >> 4;
    const int fr = lane & 15, fk = lane >> 4;

    for (int tile = wave; tile < tiles; tile += 8) {
        const int wr = wr0 + tile * 16 + fr;
        const bool in = wr < c.mu;
        const double *xrow = Xp + (size_t) (in ? order[c.lambda - c.mu + wr] : 0) * 128 + fk;
        d4_t acc[8];
#pragma unroll
        for (int t = 0; t < 8; t++) acc[t] = d4_t { 0., 0., 0., 0. };
        // eight k-steps at a time; the next chunk's gather is in flight during the sweep
        double raw[8];
#pragma unroll
        for (int i = 0; i < 8; i++) raw[i] = xrow[4 * i];
#pragma unroll 1
        for (int kc = 0; kc < 4; kc++) {
            double a[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int kk = 4 * (8 * kc + i) + fk;
                a[i] = (in && kk < c.n) ? raw[i] - xo[kk] : 0.;
            }
            if (kc < 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) raw[i] = xrow[4 * (8 * (kc + 1) + i)];
            }
            const double *bk = is + kc * 8 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 8; i++) {
#pragma unroll
                for (int t = 0; t < 8; t++)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bk[(t * 32 + i) * 64],
                            acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double ss = 0.;
#pragma unroll
            for (int t = 0; t < 8; t++) ss += acc[t][r] * acc[t][r];
            ss = row16_sum(ss);
            const int orow = wr0 + tile * 16 + fk + 4 * r;
            if (fr == 0 && orow < c.mu_pad) d.S[(size_t) p * c.mu_pad + orow] = ss;
        }
    }
}

// the rank-mu (+) / active (-) coefficient of the candidate with this rank
// (active_cmaes.cpp:136,150,158; cmaes.cpp:137), and its recombination weight
__device__ inline void rank_coefficients(const CmaDev &d, const CmaConst &c, int p, int rank,
        double &wmean, double &vgram)
{
    wmean = 0.;
    vgram = 0.;
    if (rank < c.mu) {
        wmean = d.weights[rank];
        const double cmu1 = c.variant == 1 ? c.cmu + c.cneg * (1. - c.alphaold) : c.cmu;
        vgram = cmu1 * d.weights[rank];
    } else if (c.variant == 1 && rank >= c.lambda - c.mu) {
        const int k = c.lambda - 1 - rank;
        const double *S = d.S + (size_t) p * c.mu_pad;
        const double ycoeff = S[k] / fmax(S[c.mu - 1 - k], 1e-8);
        vgram = -c.cneg * d.weights[k] * ycoeff;
    }
}

// ---------------------------------------------------------------------------
// gram: slab s of  G = sum_k v_k y_k y_k^T  (lower 16x16 tiles) and of sum_k w_k y_k
// grid (splits, tile_groups, P), 256 threads; dynamic LDS rps*ldy + rps doubles
// ---------------------------------------------------------------------------
__device__ inline void tri_tile(int q, int &ti, int &tj)
{
    ti = (int) ((sqrt(8. * q + 1.) - 1.) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= q) ti++;
    while (ti * (ti + 1) / 2 > q) ti--;
    tj = q - ti * (ti + 1) / 2;
}

constexpr int GRAM_TPW = 9;   // lower 16x16 tiles per wavefront (36 per workgroup)

__device__ __forceinline__ void gram_body(const CmaDev &d, const CmaConst &c, int p, int s, int tg,
        int ldy, double *lds)
{
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ld = c.ld, rps = c.rps;
    double *Y = lds;                 // [rps][ldy]   y = (x - xold) / sigma
    double *V = lds + rps * ldy;     // [rps]        gram coefficient of the row
    double *W = V + rps;             // [rps]        recombination weight of the row
    double *R = W + rps;             // [rows per pass][ld] partial sums of the mean
    const double *Xp = d.X + (size_t) p * c.lambda_pad * ld;
    const double *xold = d.xmean + (size_t) p * ld;
    const int *rank = d.rank + (size_t) p * c.lambda_pad;
    const double isig = 1. / sc->sigma;
    const int row0 = s * rps;
    // (rows of this slab that exist at all: lambda_pad is a multiple of 16, so are the slab's live
    // rows -- at lambda = 20 the slab of 64 is half empty, and staging and sweeping the empty half
    // was a third of this kernel's time at n = 256)
    const int rlive = min(rps, c.lambda_pad - row0);

    if (tid < rps) {
        const int row = row0 + tid;
        double wm = 0., vg = 0.;
        if (row < c.lambda) rank_coefficients(d, c, p, rank[row], wm, vg);
        V[tid] = vg;
        W[tid] = wm;
    }
    __syncthreads();
    // stage the slab: 4 columns per thread, `rpp` rows per pass, all loads independent;
    // the same pass accumulates this slab's share of the weighted mean
    const int tpr = ld >> 2, rpp = 256 / tpr;
    const int c4 = (tid % tpr) * 4, r0 = tid / tpr;
    if (r0 < rpp) {
        double m0 = 0., m1 = 0., m2 = 0., m3 = 0.;
        const double2 xo01 = *reinterpret_cast<const double2*>(&xold[c4]);
        const double2 xo23 = *reinterpret_cast<const double2*>(&xold[c4 + 2]);
        for (int r = r0; r < rlive; r += rpp) {
            const int row = row0 + r;
            double2 a01 = make_double2(0., 0.), a23 = make_double2(0., 0.);
            if (row < c.lambda) {
                a01 = *reinterpret_cast<const double2*>(&Xp[(size_t) row * ld + c4]);
                a23 = *reinterpret_cast<const double2*>(&Xp[(size_t) row * ld + c4 + 2]);
            }
            const double wr = W[r];
            const bool in = row < c.lambda;
            double2 y01, y23;
            y01.x = (in && c4 < c.n) ? (a01.x - xo01.x) * isig : 0.;
            y01.y = (in && c4 + 1 < c.n) ? (a01.y - xo01.y) * isig : 0.;
            y23.x = (in && c4 + 2 < c.n) ? (a23.x - xo23.x) * isig : 0.;
            y23.y = (in && c4 + 3 < c.n) ? (a23.y - xo23.y) * isig : 0.;
            m0 += wr * y01.x; m1 += wr * y01.y; m2 += wr * y23.x; m3 += wr * y23.y;
            *reinterpret_cast<double2*>(&Y[r * ldy + c4]) = y01;
            *reinterpret_cast<double2*>(&Y[r * ldy + c4 + 2]) = y23;
        }
        if (tg == 0) {
            double *rr = R + r0 * ld + c4;
            rr[0] = m0; rr[1] = m1; rr[2] = m2; rr[3] = m3;
        }
    }
    __syncthreads();
    if (tg == 0)
        for (int col = tid; col < ld; col += 256) {
            double msum = 0.;
            for (int g = 0; g < rpp; g++) msum += R[g * ld + col];
            d.mean_part[((size_t) p * c.splits + s) * ld + col] = msum;
        }

    const int NT = ld >> 4, LT = NT * (NT + 1) / 2;
    d4_t acc[GRAM_TPW];
    int ti[GRAM_TPW], tj[GRAM_TPW];
#pragma unroll
    for (int t = 0; t < GRAM_TPW; t++) {
        acc[t] = d4_t { 0., 0., 0., 0. };
        const int q = tg * (4 * GRAM_TPW) + wave + 4 * t;
        ti[t] = tj[t] = 0;
        if (q < LT) tri_tile(q, ti[t], tj[t]);
    }
    const int fr = lane & 15, fk = lane >> 4;
    for (int ks = 0; ks < (rlive >> 2); ks++) {
        const int k = 4 * ks + fk;
        const double vk = V[k];
        const double *yk = Y + k * ldy;
#pragma unroll
        for (int t = 0; t < GRAM_TPW; t++) {
            const int q = tg * (4 * GRAM_TPW) + wave + 4 * t;
            if (q < LT) {
                const double a = vk * yk[ti[t] * 16 + fr];
                const double b = yk[tj[t] * 16 + fr];
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    double *G = d.gram_part + ((size_t) p * c.splits + s) * ld * ld;
#pragma unroll
    for (int t = 0; t < GRAM_TPW; t++) {
        const int q = tg * (4 * GRAM_TPW) + wave + 4 * t;
        if (q < LT) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = ti[t] * 16 + (lane >> 4) + 4 * r;
                const int j = tj[t] * 16 + (lane & 15);
                G[(size_t) i * ld + j] = acc[t][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void cma_gram(CmaDev d, CmaConst c, int ldy)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    gram_body(d, c, blockIdx.z, blockIdx.x, blockIdx.y, ldy, lds);
}

// ---------------------------------------------------------------------------
// gram for ld == 128 (the n = 128 headline shape): the slab streams through LDS in 32-row
// chunks, double buffered, the next chunk's global loads in flight while the matrix cores
// work on the current one.  The 36 lower tiles are dealt to the four wavefronts as
// rectangular blocks, so a k-step costs 5-6 LDS reads for 9 MFMAs:
//     wave 0: tile rows 5-7 x cols 0-2        wave 1: tile rows 2-4 x cols 0-2
//     wave 2: rows 5-7 x cols 3-4, (3,3) (4,3) (4,4)
//     wave 3: (0,0) (1,0) (1,1), lower triangle of rows 5-7 x cols 5-7
// grid (splits, P), 256 threads, 2 workgroups per CU
// ---------------------------------------------------------------------------
constexpr int G128_CH = 32;          // rows per chunk
constexpr int G128_PACE = 2;         // cma_gram128s: chunks between two pacing barriers (1, 2: 366 us; 4: 370; 8: 371; none: 374)
constexpr int G128_LDY = 128 + 16;
constexpr int G128_TI[4][9] = { { 5, 5, 5, 6, 6, 6, 7, 7, 7 }, { 2, 2, 2, 3, 3, 3, 4, 4, 4 },
        { 5, 5, 6, 6, 7, 7, 3, 4, 4 }, { 0, 1, 1, 5, 6, 6, 7, 7, 7 } };
constexpr int G128_TJ[4][9] = { { 0, 1, 2, 0, 1, 2, 0, 1, 2 }, { 0, 1, 2, 0, 1, 2, 0, 1, 2 },
        { 3, 4, 3, 4, 3, 4, 3, 3, 4 }, { 0, 0, 1, 5, 5, 6, 5, 6, 7 } };

constexpr bool g128_mean(int wv, int i) { return (wv == 1 && i <= 4) || (wv == 2 && i >= 5); }
constexpr bool g128_needs_a(int wv, int i)
{
    for (int t = 0; t < 9; t++) if (G128_TI[wv][t] == i) return true;
    return false;
}
constexpr bool g128_needs_y(int wv, int i)
{
    for (int t = 0; t < 9; t++) if (G128_TI[wv][t] == i || G128_TJ[wv][t] == i) return true;
    return g128_mean(wv, i);
}

template<int WV>
__device__ inline void gram128_chunk(const double *Yb, const double *Vb, const double *Wb,
        d4_t (&acc)[9], double (&macc)[8], int fr, int fk)
{
#pragma unroll
    for (int ks = 0; ks < G128_CH / 4; ks++) {
        const int k = 4 * ks + fk;
        const double vk = Vb[k];
        const double *yk = Yb + k * G128_LDY + fr;
        double y[8], a[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            y[i] = a[i] = 0.;
            if (g128_needs_y(WV, i)) y[i] = yk[i * 16];
            if (g128_needs_a(WV, i)) a[i] = vk * y[i];
        }
        if (WV == 1 || WV == 2) {
            const double wk = Wb[k];
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (g128_mean(WV, i)) macc[i] = fma(wk, y[i], macc[i]);
        }
#pragma unroll
        for (int t = 0; t < 9; t++)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[G128_TI[WV][t]], y[G128_TJ[WV][t]],
                    acc[t], 0, 0, 0);
    }
}

// The sums are formed on x - m (not on y = (x - m) / sigma): the division by sigma is applied ONCE,
// to the finished sums (isig^2 for the products, isig for the mean) -- a multiplication per fragment
// and k-step less on the pipe the matrix instruction needs (round 4: 6 of the ~25 vector
// instructions a wavefront issues next to the 9 MFMAs of a k-step).
template<int WV>
__device__ inline void gram128_store(double *G, double *mp, const d4_t (&acc)[9],
        double (&macc)[8], int lane, double isig)
{
    const double isig2 = isig * isig;
#pragma unroll
    for (int t = 0; t < 9; t++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = G128_TI[WV][t] * 16 + (lane >> 4) + 4 * r;
            const int j = G128_TJ[WV][t] * 16 + (lane & 15);
            G[(size_t) i * 128 + j] = acc[t][r] * isig2;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (g128_mean(WV, i)) {
            double m = macc[i];
            m += __shfl_xor(m, 16, 64);
            m += __shfl_xor(m, 32, 64);
            if (lane < 16) mp[i * 16 + lane] = m * isig;
        }
    }
}

__global__ __launch_bounds__(256, 2) void cma_gram128(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y, s = blockIdx.x;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double (*Y)[G128_CH * G128_LDY] = reinterpret_cast<double (*)[G128_CH * G128_LDY]>(lds);
    double (*V)[G128_CH] = reinterpret_cast<double (*)[G128_CH]>(lds + 2 * G128_CH * G128_LDY);
    double (*W)[G128_CH] = V + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Xp = d.X + (size_t) p * c.lambda_pad * 128;
    const double *xold = d.xmean + (size_t) p * 128;
    const int *rank = d.rank + (size_t) p * c.lambda_pad;
    const double isig = 1. / sc->sigma;
    const int row0 = s * c.rps;
    const int nch = (min(c.rps, c.lambda_pad - row0) + G128_CH - 1) / G128_CH;

    const int c4 = (tid & 31) * 4, r0 = tid >> 5;
    const double2 xo01 = *reinterpret_cast<const double2*>(&xold[c4]);
    const double2 xo23 = *reinterpret_cast<const double2*>(&xold[c4 + 2]);
    const bool in0 = c4 < c.n, in1 = c4 + 1 < c.n, in2 = c4 + 2 < c.n, in3 = c4 + 3 < c.n;
    double2 pf[G128_CH / 8][2];
    double pv = 0., pw = 0.;

    auto fetch = [&](int ch) {
        const int base = row0 + ch * G128_CH;
#pragma unroll
        for (int i = 0; i < G128_CH / 8; i++) {
            const int row = base + r0 + 8 * i;
            const double *src = Xp + (size_t) row * 128 + c4;
            pf[i][0] = pf[i][1] = make_double2(0., 0.);
            if (row < c.lambda_pad) {
                pf[i][0] = *reinterpret_cast<const double2*>(src);
                pf[i][1] = *reinterpret_cast<const double2*>(src + 2);
            }
        }
        if (tid < G128_CH) {
            const int row = base + tid;
            pv = pw = 0.;
            if (row < c.lambda) rank_coefficients(d, c, p, rank[row], pw, pv);
        }
    };
    auto stash = [&](int ch, int buf) {
        const int base = row0 + ch * G128_CH;
#pragma unroll
        for (int i = 0; i < G128_CH / 8; i++) {
            const int r = r0 + 8 * i;
            const bool in = base + r < c.lambda;
            double2 y01, y23;
            y01.x = (in && in0) ? pf[i][0].x - xo01.x : 0.;
            y01.y = (in && in1) ? pf[i][0].y - xo01.y : 0.;
            y23.x = (in && in2) ? pf[i][1].x - xo23.x : 0.;
            y23.y = (in && in3) ? pf[i][1].y - xo23.y : 0.;
            *reinterpret_cast<double2*>(&Y[buf][r * G128_LDY + c4]) = y01;
            *reinterpret_cast<double2*>(&Y[buf][r * G128_LDY + c4 + 2]) = y23;
        }
        if (tid < G128_CH) {
            V[buf][tid] = pv;
            W[buf][tid] = pw;
        }
    };

    d4_t acc[9];
    double macc[8];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = d4_t { 0., 0., 0., 0. };
#pragma unroll
    for (int i = 0; i < 8; i++) macc[i] = 0.;
    const int fr = lane & 15, fk = lane >> 4;

    fetch(0);
    stash(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nch; ch++) {
        const int buf = ch & 1;
        if (ch + 1 < nch) fetch(ch + 1);
        switch (wave) {
        case 0: gram128_chunk<0>(Y[buf], V[buf], W[buf], acc, macc, fr, fk); break;
        case 1: gram128_chunk<1>(Y[buf], V[buf], W[buf], acc, macc, fr, fk); break;
        case 2: gram128_chunk<2>(Y[buf], V[buf], W[buf], acc, macc, fr, fk); break;
        default: gram128_chunk<3>(Y[buf], V[buf], W[buf], acc, macc, fr, fk); break;
        }
        if (ch + 1 < nch) stash(ch + 1, buf ^ 1);
        __syncthreads();
    }

    double *G = d.gram_part + ((size_t) p * c.splits + s) * 128 * 128;
    double *mp = d.mean_part + ((size_t) p * c.splits + s) * 128;
    switch (wave) {
    case 0: gram128_store<0>(G, mp, acc, macc, lane, isig); break;
    case 1: gram128_store<1>(G, mp, acc, macc, lane, isig); break;
    case 2: gram128_store<2>(G, mp, acc, macc, lane, isig); break;
    default: gram128_store<3>(G, mp, acc, macc, lane, isig); break;
    }
}

// ---------------------------------------------------------------------------
// The same Gram slab WITHOUT the shared LDS tile and its barriers: every wavefront loads the
// fragments its tile set needs (5-6 of the 8 column tiles of a row) straight from X in MFMA
// layout -- lane (fr, fk) of k-step ks reads X[base + 4 ks + fk][16 i + fr], 128-byte segments,
// the re-reads of the four wavefronts served by L1/L2 -- and turns them into y = (x - m) / sigma
// itself.  With cma_gram128 the chunk barrier cost a quarter of the kernel (measured by running
// it without: 420 -> 305 us at M): two wavefronts per SIMD cannot cover a workgroup that waits.
// Here the wavefronts of a workgroup never wait for each other; the loads run four k-steps
// (~2300 cycles of MFMA) ahead in a register ring of four slots, the rank -> coefficient
// look-ups a whole chunk ahead in a wavefront-private LDS strip.  Same products in the same
// order: bit-identical to cma_gram128.
// grid (splits, P), 256 threads, no dynamic LDS
// ---------------------------------------------------------------------------
template<int WV>
__device__ __forceinline__ void gram128_stream(const CmaDev &d, const CmaConst &c, int p, int s,
        double *coef, int lane)
{
    const CmaScal *sc = d.scal + p;
    const double *Xp = d.X + (size_t) p * c.lambda_pad * 128;
    const double *xold = d.xmean + (size_t) p * 128;
    const int *rank = d.rank + (size_t) p * c.lambda_pad;
    const double isig = 1. / sc->sigma;
    const int row0 = s * c.rps;
    const int nrows = min(c.rps, c.lambda_pad - row0);
    const int nch = (nrows + G128_CH - 1) / G128_CH;
    const int fr = lane & 15, fk = lane >> 4;
    const bool pace = !(d.dbg & 65536);

    double xo[8];
#pragma unroll
    for (int i = 0; i < 8; i++) xo[i] = g128_needs_y(WV, i) ? xold[16 * i + fr] : 0.;
    d4_t acc[9];
    double macc[8];
#pragma unroll
    for (int t = 0; t < 9; t++) acc[t] = d4_t { 0., 0., 0., 0. };
#pragma unroll
    for (int i = 0; i < 8; i++) macc[i] = 0.;

    // raw X fragments, four k-steps deep: slot j holds k-step j, then j + 4 (reloaded as soon as
    // it has been swept), then k-step j of the next chunk, ...
    double ring[4][8];
    const int lane_off = fk * 128 + fr;
    auto load_step = [&](int ch, int ks, int slot) {
        // four whole rows or none (lambda_pad is a multiple of 16): a uniform clamp keeps the
        // loads unconditional and the address a scalar base + a per-lane constant; rows past the
        // population carry zero coefficients in the sweep, whatever is loaded for them
        const int blk = min(row0 + ch * G128_CH + 4 * ks, c.lambda_pad - 4);
        const double *src = Xp + (size_t) blk * 128;
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (g128_needs_y(WV, i)) ring[slot][i] = src[lane_off + 16 * i];
    };
    // Coefficients of a chunk's 32 rows (rank_coefficients, same arithmetic): lanes 0..31 look
    // them up, the LDS strip hands them round.  The look-up is a chain of two dependent loads; it
    // runs TWO chunks ahead, one link per chunk, branch-free with clamped indices -- the memory
    // counter retires in order, so a load that is consumed long after it was issued costs no wait,
    // while one consumed at once would drain the whole prefetch ring first.
    const double cmu1 = c.variant == 1 ? c.cmu + c.cneg * (1. - c.alphaold) : c.cmu;
    const double *Sp = d.S + (size_t) p * c.mu_pad;
    int rk = 0;                           // link 1: the rank of the row (chunk ch + 2)
    int rk2 = 0;                          // ... one chunk on, while link 2 is in flight
    double wa = 0., sk = 1., sm = 1.;     // link 2: weights[.], S[k], S[mu - 1 - k] (chunk ch + 1)
    auto coef_link1 = [&](int ch) {
        const int row = row0 + ch * G128_CH + lane;
        const bool ok = lane < G128_CH && ch < nch && row < c.lambda;
        rk = rank[ok ? row : 0];
        if (!ok) rk = -1;
    };
    auto coef_link2 = [&]() {             // from the rank loaded one chunk ago
        rk2 = rk;
        const bool pos = rk2 >= 0 && rk2 < c.mu;
        const int k = min(max(c.lambda - 1 - rk2, 0), c.mu - 1);
        wa = d.weights[pos ? rk2 : k];
        sk = Sp[k];
        sm = Sp[c.mu - 1 - k];
    };
    auto put_coef = [&](int buf) {        // from the values loaded one chunk ago
        double pv = 0., pw = 0.;
        if (rk2 >= 0 && rk2 < c.mu) {
            pw = wa;
            pv = cmu1 * wa;
        } else if (rk2 >= 0 && c.variant == 1 && rk2 >= c.lambda - c.mu) {
            const double ycoeff = sk / fmax(sm, 1e-8);
            pv = -c.cneg * wa * ycoeff;
        }
        if (lane < G128_CH) {
            coef[buf * 2 * G128_CH + lane] = pv;
            coef[buf * 2 * G128_CH + G128_CH + lane] = pw;
        }
    };
    auto sweep_step = [&](int slot, double vk, double wk) {
        double y[8], a[8];
        // (no masks: columns >= n are zero in X and in the mean alike, rows >= lambda have
        // v = w = 0 -- cma_gram128 zeroes y there instead, the products are the same zeros)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            y[i] = a[i] = 0.;
            if (g128_needs_y(WV, i)) y[i] = ring[slot][i] - xo[i];
            if (g128_needs_a(WV, i)) a[i] = vk * y[i];
        }
        if (WV == 1 || WV == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (g128_mean(WV, i)) macc[i] = fma(wk, y[i], macc[i]);
        }
#pragma unroll
        for (int t = 0; t < 9; t++)
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[G128_TI[WV][t]], y[G128_TJ[WV][t]],
                    acc[t], 0, 0, 0);
    };

    // prologue: coefficients of chunks 0 and 1, the first four k-steps
    coef_link1(0);
    coef_link2();
    put_coef(0);
    coef_link1(1);
    coef_link2();                         // (chunk 1's values: written to the strip at the end of chunk 0)
    coef_link1(2);
#pragma unroll
    for (int ks = 0; ks < 4; ks++) load_step(0, ks, ks);
    cma_wave_sync();
    for (int ch = 0; ch < nch; ch++) {
        const int buf = ch & 1;
        // (this k-step's coefficients were read from the strip a k-step ago: with the scheduling
        // barrier below the LDS latency would otherwise sit in front of every k-step)
        double vk = coef[buf * 2 * G128_CH + fk], wk = coef[buf * 2 * G128_CH + G128_CH + fk];
#pragma unroll
        for (int ks = 0; ks < 8; ks++) {
            double vn = 0., wn = 0.;
            if (ks < 7) {
                vn = coef[buf * 2 * G128_CH + 4 * (ks + 1) + fk];
                if (WV == 1 || WV == 2) wn = coef[buf * 2 * G128_CH + G128_CH + 4 * (ks + 1) + fk];
            }
            sweep_step(ks & 3, vk, wk);
            vk = vn;
            wk = wn;
            if (ks < 4) load_step(ch, ks + 4, ks & 3);
            else load_step(ch + 1, ks - 4, ks & 3);
            // (keeps a slot's subtraction next to ITS k-step: hoisted to the top of the chunk, as
            // the scheduler would, the four slots are waited for at once and nothing is ahead)
            __builtin_amdgcn_sched_barrier(0);
        }
        put_coef(buf ^ 1);                // chunk ch + 1, loaded during chunk ch - 1 / the prologue
        coef_link2();                     // chunk ch + 2, from the rank loaded a chunk ago
        coef_link1(ch + 3);
        cma_wave_sync();
        // Pacing, not synchronisation: the four wavefronts share no data, but they read the same
        // rows, and left alone they drift apart until a row one of them fetched has left L1 / L2
        // when the next one asks for it (1.49 x the algorithmic bytes from HBM, round-3 counters).
        // A bare s_barrier every PACE chunks (no memory wait attached) bounds the drift.
        if (pace && (ch & (G128_PACE - 1)) == G128_PACE - 1) __builtin_amdgcn_s_barrier();
    }
    double *G = d.gram_part + ((size_t) p * c.splits + s) * 128 * 128;
    double *mp = d.mean_part + ((size_t) p * c.splits + s) * 128;
    gram128_store<WV>(G, mp, acc, macc, lane, isig);
}

__global__ __launch_bounds__(256, 2) void cma_gram128s(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y, s = blockIdx.x;
    if (pop_frozen(c, d.scal + p)) return;
    __shared__ double coef[4][2 * 2 * G128_CH];      // per wavefront: [buffer][v | w][row of chunk]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    switch (wave) {
    case 0: gram128_stream<0>(d, c, p, s, coef[0], lane); break;
    case 1: gram128_stream<1>(d, c, p, s, coef[1], lane); break;
    case 2: gram128_stream<2>(d, c, p, s, coef[2], lane); break;
    default: gram128_stream<3>(d, c, p, s, coef[3], lane); break;
    }
}

// ---------------------------------------------------------------------------
// paths: mean, ps, hsig, pc, sigma -- one workgroup of T threads per population (256; the lazy
// form of a big batch 1024: its two passes over B are a handful of dependent round trips to L2 /
// HBM per thread, and four times the threads make a quarter of the trips)
// ---------------------------------------------------------------------------
template<bool LAZY = false, int T = 256>   // LAZY: c.lazy_isc configurations (C^-1/2 dm from B and D)
__device__ __forceinline__ void paths_body(const CmaDev &d, const CmaConst &c, int p)
{
    CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    __shared__ double dm[512];
    __shared__ double red[T / 64];
    const int tid = threadIdx.x, ld = c.ld;
    double *xmean = d.xmean + (size_t) p * ld, *xold = d.xold + (size_t) p * ld;
    double *ps = d.ps + (size_t) p * ld, *pc = d.pc + (size_t) p * ld;
    const double sigma = sc->sigma;

    // weighted mean (active_cmaes.cpp:75-85 / cmaes.cpp:85-96)
    for (int j = tid; j < ld; j += T) {
        // (the slabs in their order, eight loads in flight at a time: summed straight off the loop
        // every addition waited for its own round trip -- with 32 slabs, 5 us of a single-population
        // generation here and 6 in cma_cov)
        double sum = 0.;
        {
            const double *mp = d.mean_part + (size_t) p * c.splits * ld + j;
            int s = 0;
            for (; s + 8 <= c.splits; s += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = mp[(size_t) (s + u) * ld];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++) sum += x[u];
            }
            for (; s < c.splits; s++) sum += mp[(size_t) s * ld];
        }
        const double xo = xmean[j];
        sum = xo + sigma * sum;   // the slabs hold sum_k w_k (x_k - xold) / sigma
        double xn = 0.;
        if (j < c.n) {
            xn = c.variant == 1 ? xo * (1. - c.cm) + sum * c.cm : sum;
            if (c.bound) xn = fmax(d.lower[j], fmin(xn, d.upper[j]));
        }
        xold[j] = xo;
        xmean[j] = xn;
        dm[j] = xn - xo;
    }
    __syncthreads();

    // ps (active_cmaes.cpp:88-95); C^-1/2 is symmetric, so column reads are row reads
    const double csc = sqrt(c.cs * (2. - c.cs) * c.mueff);
    const double den = c.variant == 1 ? c.cm * sigma : sigma;
    const double *isc = d.isc + (size_t) p * ld * ld;
    double ssq = 0.;
    if (LAZY && sc->basis_ok) {
        // C^-1/2 dm = B (D^-1 (B^T dm)) from the basis itself: 2 n^2 operations instead of the
        // n^3 of forming C^-1/2 after every decomposition (cma_post then only packs B D)
        constexpr int HN = T / 128;         // row groups of a column in the first pass
        constexpr int RS = T / 16, NU = 128 / RS;   // rows per sweep / sweeps of the second
        __shared__ double sv[256], cv[256], part[HN - 1][128];
        const double *B = d.B + (size_t) p * ld * ld, *D = d.D + (size_t) p * ld;
        // (both products with the loads of a thread independent of each other: the matrix comes
        // from L2 / HBM once, latency is what there is to hide)
        for (int jb = 0; jb < c.n; jb += 128) {     // (one pass for n <= 128, two up to 256)
            // t = B^T dm: column j by threads j, j + 128, ... (rows h, h + HN, ...), 8 rows in flight
            const int j = jb + (tid & 127), h = tid >> 7;
            double t0 = 0., t1 = 0., t2 = 0., t3 = 0.;
            if (j < c.n) {
                int i = h;
                for (; i + 7 * HN < c.n; i += 8 * HN) {
                    double b[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) b[u] = B[(size_t) (i + HN * u) * ld + j];
                    t0 += b[0] * dm[i] + b[4] * dm[i + 4 * HN];
                    t1 += b[1] * dm[i + HN] + b[5] * dm[i + 5 * HN];
                    t2 += b[2] * dm[i + 2 * HN] + b[6] * dm[i + 6 * HN];
                    t3 += b[3] * dm[i + 3 * HN] + b[7] * dm[i + 7 * HN];
                }
                for (; i < c.n; i += HN) t0 += B[(size_t) i * ld + j] * dm[i];
            }
            const double t = (t0 + t1) + (t2 + t3);
            if (h > 0) part[h - 1][tid & 127] = t;
            __syncthreads();
            if (h == 0) {
                double tt = t;
#pragma unroll
                for (int q = 0; q < HN - 1; q++) tt += part[q][tid & 127];
                sv[j] = j < c.n ? tt / D[j] : 0.;
            }
            __syncthreads();
        }
        for (int ib = 0; ib < c.n; ib += 128) {
            // cv = B sv: rows r, r + RS, ... by the 16 lanes of a DPP row (128-byte segments)
            const int g = tid & 15, r = ib + (tid >> 4);
            double a[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int i = r + RS * u;
                double tot = 0.;
                for (int kb = 0; kb < c.n; kb += 128) {
                    double x[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int j = kb + g + 16 * k;
                        x[k] = (i < c.n && j < c.n) ? B[(size_t) i * ld + j] * sv[j] : 0.;
                    }
                    const double part8 = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
                    tot = kb == 0 ? part8 : tot + part8;
                }
                a[u] = tot;
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const double v = group_sum<16>(a[u]);
                if (g == 0 && r + RS * u < c.n) cv[r + RS * u] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < ld; i += T) {
            const double v = i < c.n ? (1. - c.cs) * ps[i] + csc * cv[i] / den : 0.;
            ps[i] = v;
            ssq += v * v;
        }
    } else {
        for (int i = tid; i < ld; i += T) {
            double v = 0.;
            if (i < c.n) {
                double acc = 0.;
                for (int j = 0; j < c.n; j++) acc += isc[(size_t) j * ld + i] * dm[j];
                v = (1. - c.cs) * ps[i] + csc * acc / den;
            }
            ps[i] = v;
            ssq += v * v;
        }
    }
    ssq = wave_sum(ssq);
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
    __syncthreads();
    double rsum = 0.;
#pragma unroll
    for (int q = 0; q < T / 64; q++) rsum += red[q];
    const double pslen = sqrt(rsum);

    // hsig (active_cmaes.cpp:98-105); fev already counts this generation
    const double denom = 1. - pow(1. - c.cs, 2. * sc->fev / c.lambda);
    const int hsig = (pslen / sqrt(denom) / c.chi < 1.4 + 2. / (c.n + 1.)) ? 1 : 0;

    // pc (active_cmaes.cpp:108-112)
    const double ccc = sqrt(c.cc * (2. - c.cc) * c.mueff);
    for (int i = tid; i < ld; i += T) {
        double v = 0.;
        if (i < c.n) v = (1. - c.cc) * pc[i] + hsig * ccc * dm[i] / den;
        pc[i] = v;
    }

    // sigma (base_cmaes.cpp:176-189); uses the history of the PREVIOUS generations
    if (tid == 0) {
        const double *f = d.f + (size_t) p * c.lambda_pad;
        const int *order = d.order + (size_t) p * c.lambda_pad;
        double sg = sigma * exp(fmin(1., (c.cs / c.damps) * (pslen / c.chi - 1.)));
        if (f[order[0]] == f[order[c.ik]]) sg *= exp(0.2 + c.cs / c.damps);
        if (sc->it >= c.hlen && sc->fworst - sc->fbest == 0.) sg *= exp(0.2 + c.cs / c.damps);
        sc->sigma = sg;
        sc->hsig = hsig;
        sc->pslen = pslen;
    }
}

__global__ __launch_bounds__(256) void cma_paths(CmaDev d, CmaConst c)
{
    paths_body<false>(d, c, blockIdx.x);
}
__global__ __launch_bounds__(256) void cma_paths_lazy(CmaDev d, CmaConst c)
{
    paths_body<true>(d, c, blockIdx.x);
}
__global__ __launch_bounds__(1024) void cma_paths_lazy1k(CmaDev d, CmaConst c)
{
    paths_body<true, 1024>(d, c, blockIdx.x);
}

// ---------------------------------------------------------------------------
// cov: C <- decay * C + c1 (pc pc^T + c2 C) + sum of the Gram slabs   (lower half,
// mirrored).  grid (ceil(n*(n+1)/2 / 256), P), 256 threads.
// NB launched BEFORE cma_paths commits the new sigma? No: the slabs already hold
// y = (x - xold)/sigma_old, and this kernel reads only pc and hsig.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void cov_body(const CmaDev &d, const CmaConst &c, int p, int bx)
{
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const int q = bx * 256 + threadIdx.x;
    const int total = c.n * (c.n + 1) / 2;
    if (q >= total) return;
    int i, j;
    tri_tile(q, i, j);   // same triangular unranking, on elements
    const int ld = c.ld;
    double *C = d.C + (size_t) p * ld * ld;
    const double *pc = d.pc + (size_t) p * ld;
    const double cij = C[(size_t) i * ld + j];
    const double c2 = (1. - sc->hsig) * c.cc * (2. - c.cc);
    const double decay = c.variant == 1 ? (1. - c.c1 - c.cmu + c.cneg * c.alphaold)
                                        : (1. - c.c1 - c.cmu);
    double sum = decay * cij + c.c1 * (pc[i] * pc[j] + c2 * cij);
    const double *G = d.gram_part + (size_t) p * c.splits * ld * ld + (size_t) i * ld + j;
    double g = 0.;
    {
        int s = 0;
        for (; s + 8 <= c.splits; s += 8) {       // (as in paths_body: same order, eight loads in flight)
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = G[(size_t) (s + u) * ld * ld];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 8; u++) g += x[u];
        }
        for (; s < c.splits; s++) g += G[(size_t) s * ld * ld];
    }
    sum += g;
    C[(size_t) i * ld + j] = sum;
    C[(size_t) j * ld + i] = sum;
}

__global__ __launch_bounds__(256) void cma_cov(CmaDev d, CmaConst c)
{
    cov_body(d, c, blockIdx.y, blockIdx.x);
}

// ---------------------------------------------------------------------------
// post: C^-1/2 = B diag(1/D) B^T (cmaes.cpp:274-282) and the two packed MFMA operands
// grid (ld/16, ld/16, P), 256 threads (one 16x16 tile per workgroup)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cma_post(CmaDev d, CmaConst c, int mode)
{
    const int p = blockIdx.z;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    // mode 0: after cma_eigen, only if it decomposed; 1: always; 2: pack only (C^-1/2 is
    // taken as stored -- Cmaes::init sets it to I whatever B holds, cmaes.cpp:55-59)
    if (mode == 0 && !sc->eigen_done) return;
    // mode 3: C^-1/2 on demand (bbo_get "invsqrtC" under lazy_isc) -- only from a consistent basis
    if (mode == 3 && !sc->basis_ok) return;
    // lazy_isc (no box, ld <= 256): C^-1/2 is not formed after a decomposition -- cma_paths works
    // from B and D, the whitened norms come from the sampler -- this launch only packs B D
    const bool gemm = mode == 3 || (mode != 2 && !c.lazy_isc);
    const int ld = c.ld, n = c.n;
    const int tid = threadIdx.x;
    const int i = blockIdx.y * 16 + (tid >> 4), j = blockIdx.x * 16 + (tid & 15);
    const double *B = d.B + (size_t) p * ld * ld;
    const double *D = d.D + (size_t) p * ld;
    __shared__ double Bi[16][17], Bj[16][17], Dinv[16];
    double sum = 0.;
    if (gemm)
    for (int k0 = 0; k0 < ld; k0 += 16) {
        const int r = tid >> 4, k = k0 + (tid & 15);
        const int ri = blockIdx.y * 16 + r, rj = blockIdx.x * 16 + r;
        Bi[r][tid & 15] = (ri < n && k < n) ? B[(size_t) ri * ld + k] : 0.;
        Bj[r][tid & 15] = (rj < n && k < n) ? B[(size_t) rj * ld + k] : 0.;
        if (tid < 16) Dinv[tid] = (k0 + tid < n) ? D[k0 + tid] : 1.;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) sum += Bi[tid >> 4][kk] / Dinv[kk] * Bj[tid & 15][kk];
        __syncthreads();
    }
    const bool in = i < n && j < n;
    // packed B-operand element (row i -> column tile/lane, col j -> k index)
    const int KS = ld >> 2;
    const size_t pk = ((size_t) (i >> 4) * KS + (j >> 2)) * 64 + ((j & 3) << 4) + (i & 15);
    if (gemm || mode == 2) {
        double v = in ? sum : 0.;
        if (mode == 2) v = in ? d.isc[(size_t) p * ld * ld + (size_t) i * ld + j] : 0.;
        else d.isc[(size_t) p * ld * ld + (size_t) i * ld + j] = v;
        d.ISp[(size_t) p * ld * ld + pk] = v;
    }
    if (mode != 2 && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) d.scal[p].basis_ok = 1;
    d.BDp[(size_t) p * ld * ld + pk] = in ? B[(size_t) i * ld + j] * D[j] : 0.;
}

// ---------------------------------------------------------------------------
// post for ld <= 128 on the matrix cores: NBW (1 or 4: few populations) workgroups per
// population, each stages B in LDS and owns 8/NBW tile rows of C^-1/2 = (B diag(1/D)) B^T (A fragment = B[i][k] / D[k], the
// reference's own term order, cmaes.cpp:277; B fragment = the same rows of B unscaled) and a
// 1/NBW of the two packed operands.  grid (P, NBW), 256 threads, dynamic LDS ld*(ld+2)+ld doubles
// ---------------------------------------------------------------------------
template<int NBW>
__global__ __launch_bounds__(256) void cma_post_mfma(CmaDev d, CmaConst c, int mode)
{
    const int p = blockIdx.x;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    if (mode == 0 && !sc->eigen_done) return;
    if (mode == 3 && !sc->basis_ok) return;
    // modes: 0 after a decomposition, 1 after B or D were set, 2 pack the operands of the C^-1/2
    // that is there (init), 3 only C^-1/2 of the current basis (a reader asked for it while the
    // engine runs with lazy_isc: cma_paths then works from B and D and nothing else needs it)
    const bool gemm = mode == 3 || (mode != 2 && !c.lazy_isc);
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int ld = c.ld, n = c.n, ldp = ld + 2;
    double *Bs = lds, *Dv = lds + ld * ldp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *B = d.B + (size_t) p * ld * ld;
    double *isc = d.isc + (size_t) p * ld * ld;
    double *ISp = d.ISp + (size_t) p * ld * ld, *BDp = d.BDp + (size_t) p * ld * ld;
    // eight independent loads in flight per thread, then the LDS stores
    const int hp = ld >> 1, total = ld * hp;
    for (int q0 = tid; q0 < total; q0 += 8 * 256) {
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + 256 * u;
            v[u] = q < total ? reinterpret_cast<const double2*>(B)[q] : make_double2(0., 0.);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = q0 + 256 * u;
            if (q < total) {
                const int i = q / hp, j = (q - i * hp) * 2;
                if (!(i < n && j < n)) v[u].x = 0.;
                if (!(i < n && j + 1 < n)) v[u].y = 0.;
                *reinterpret_cast<double2*>(&Bs[i * ldp + j]) = v[u];
            }
        }
    }
    for (int j = tid; j < ld; j += 256) Dv[j] = j < n ? d.D[(size_t) p * ld + j] : 1.;
    __syncthreads();

    const int NT = ld >> 4, KS = ld >> 2;
    const int fr = lane & 15, fk = lane >> 4;
    const int wb = blockIdx.y;            // workgroup wb of NBW: tile rows wb, wb + NBW, ...
    constexpr int NR = 8 / NBW;           // tile rows per workgroup (NBW = 4: 2, NBW = 1: 8)
    if (gemm && NBW == 1) {
        // one workgroup per population: wavefront w owns tile ROWS w, w + 4 and sweeps all eight
        // column tiles, so the divisions B[i][k] / D[k] (the reference's term order, kept) are
        // made once per row and k-step, not once per wavefront
        d4_t acc[2][8];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int t = 0; t < 8; t++) acc[h][t] = d4_t { 0., 0., 0., 0. };
        if (wave < NT) {
            const bool r1 = wave + 4 < NT;
            for (int ks = 0; ks < KS; ks++) {
                const int k = 4 * ks + fk;
                const double dk = Dv[k];
                const double a0 = Bs[(wave * 16 + fr) * ldp + k] / dk;
                const double a1 = r1 ? Bs[((wave + 4) * 16 + fr) * ldp + k] / dk : 0.;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const double f = t < NT ? Bs[(t * 16 + fr) * ldp + k] : 0.;
                    acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, f, acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, f, acc[1][t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int ti = wave + 4 * h;
                if (ti < NT) {
#pragma unroll
                    for (int t = 0; t < 8; t++) {
                        if (t < NT) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int i = ti * 16 + fk + 4 * r, j = t * 16 + fr;
                                const double v = acc[h][t][r];   // 0 outside n: B is staged as 0 there
                                isc[(size_t) i * ld + j] = v;
                                ISp[((size_t) (i >> 4) * KS + (j >> 2)) * 64 + ((j & 3) << 4) + (i & 15)] = v;
                            }
                        }
                    }
                }
            }
        }
    }
    if (gemm && NBW != 1) {
        // wavefront w: column tiles w, w + 4 of this workgroup's tile rows
        d4_t acc[NR][2];
#pragma unroll
        for (int h = 0; h < NR; h++)
#pragma unroll
            for (int u = 0; u < 2; u++) acc[h][u] = d4_t { 0., 0., 0., 0. };
        const bool c1 = wave + 4 < NT;
        if (wb < NT && wave < NT) {
            for (int ks = 0; ks < KS; ks++) {
                const int k = 4 * ks + fk;
                const double dk = Dv[k];
                const double f0 = Bs[(wave * 16 + fr) * ldp + k];
                const double f1 = c1 ? Bs[((wave + 4) * 16 + fr) * ldp + k] : 0.;
#pragma unroll
                for (int h = 0; h < NR; h++) {
                    const int ti = wb + NBW * h;
                    const double a = ti < NT ? Bs[(ti * 16 + fr) * ldp + k] / dk : 0.;
                    acc[h][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f0, acc[h][0], 0, 0, 0);
                    acc[h][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f1, acc[h][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h = 0; h < NR; h++) {
                const int ti = wb + NBW * h;
                if (ti < NT) {
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int t = wave + 4 * u;
                        if (t < NT) {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int i = ti * 16 + fk + 4 * r, j = t * 16 + fr;
                                const double v = acc[h][u][r];   // 0 outside n: B is staged as 0 there
                                isc[(size_t) i * ld + j] = v;
                                ISp[((size_t) (i >> 4) * KS + (j >> 2)) * 64 + ((j & 3) << 4) + (i & 15)] = v;
                            }
                        }
                    }
                }
            }
        }
    }
    if (mode == 3) return;
    if (mode != 2 && tid == 0 && wb == 0) d.scal[p].basis_ok = 1;
    // packed operands: element (i, j) -> column tile i >> 4, k-step j >> 2, lane (j & 3, i & 15)
    const int q0 = wb * (ld * ld / NBW), q1 = q0 + ld * ld / NBW;   // this workgroup's share
#pragma unroll 4
    for (int q = q0 + tid; q < q1; q += 256) {
        const int t4 = q >> 6, l = q & 63;             // t4 = nt * KS + ks
        const int nt = t4 / KS, ks = t4 - nt * KS;
        const int i = nt * 16 + (l & 15), j = 4 * ks + (l >> 4);
        const bool in = i < n && j < n;
        BDp[q] = in ? Bs[i * ldp + j] * Dv[j] : 0.;
    }
    if (mode == 2) {
        for (int q = q0 + tid; q < q1; q += 256) {
            const int t4 = q >> 6, l = q & 63;
            const int nt = t4 / KS, ks = t4 - nt * KS;
            const int i = nt * 16 + (l & 15), j = 4 * ks + (l >> 4);
            ISp[q] = (i < n && j < n) ? isc[(size_t) i * ld + j] : 0.;
        }
    }
}

// ---------------------------------------------------------------------------
// history + stop tests (base_cmaes.cpp:191-209, :155 it++, cmaes.cpp:151-227)
// one workgroup of 64 threads per population
// ---------------------------------------------------------------------------
__device__ __forceinline__ void history_stop_body(const CmaDev &d, const CmaConst &c, int p, int lane)
{
    CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const int ld = c.ld, n = c.n;
    double *hb = d.hist_best + (size_t) p * c.hlen, *hk = d.hist_kth + (size_t) p * c.hlen;
    const double *f = d.f + (size_t) p * c.lambda_pad;
    const int *order = d.order + (size_t) p * c.lambda_pad;
    int it = sc->it;
    int head = sc->hist_head, len = sc->hist_len;
    double fbest = sc->fbest, fworst = sc->fworst;

    if (it < c.mit) {
        head = (head + 1) % c.hlen;
        if (lane == 0) {
            hb[head] = f[order[0]];
            hk[head] = f[order[c.ik]];
        }
        if (len < c.hlen) len++;
        cma_wave_sync();
        if (len == c.hlen) {
            double lo = BBO_INF, hi = -BBO_INF;
            for (int k = lane; k < c.hlen; k += 64) {
                lo = fmin(lo, hb[k]);
                hi = fmax(hi, hb[k]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo = fmin(lo, __shfl_xor(lo, off, 64));
                hi = fmax(hi, __shfl_xor(hi, off, 64));
            }
            fbest = lo;
            fworst = hi;
        }
    }
    it++;
    cma_wave_sync();

    const double sigma = sc->sigma;
    const double *pc = d.pc + (size_t) p * ld, *xm = d.xmean + (size_t) p * ld;
    const double *C = d.C + (size_t) p * ld * ld, *B = d.B + (size_t) p * ld * ld;
    const double *D = d.D + (size_t) p * ld;
    int flag = 0;
    const int off = c.stop_off;     // (extension: silenced tests; 0 = the reference's nine)
    if (it >= c.mit) {
        flag = 1;
    } else if (f[order[0]] <= c.ftarget) {
        flag = 10;                  // (extension: target value reached; ftarget = -inf by default)
    } else if (!(off & 4) && it >= c.hlen && fworst - fbest < c.tol) {
        flag = 2;
    } else {
        // EqualFunVals
        if (!(off & 8) && len >= n) {
            int eq = 0;
            for (int i = lane; i < n; i += 64) {
                const int idx = (c.hlen + head - i) % c.hlen;
                eq += hb[idx] == hk[idx];
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) eq += __shfl_xor(eq, off, 64);
            if (3 * eq >= n) flag = 3;
        }
        const bool sep = c.variant == 2;   // SepCmaes: the same tests on _diagd (sep_cmaes.cpp:166-205)
        if (!flag && !(off & 16)) {   // TolX
            int bad = 0;
            for (int i = lane; i < n; i += 64) {
                const double sd = sep ? D[i] : sqrt(C[(size_t) i * ld + i]);
                bad |= (fmax(pc[i], sd) * sigma / c.sigma0 >= c.tol);
            }
            if (!__any(bad)) flag = 4;
        }
        if (!flag && !(off & 32) && sigma / c.sigma0 > 1.0e20 * D[n - 1]) flag = 5;
        if (!flag && !(off & 128) && D[n - 1] > 1.0e7 * D[0]) flag = 7;
        if (!flag && !(off & 256)) {   // NoEffectAxis
            const int iaxis = n - 1 - ((it - 1) % n);
            int moved = 0;
            if (sep) {
                moved = xm[iaxis] != xm[iaxis] + 0.1 * sigma * D[iaxis];
            } else {
                for (int i = lane; i < n; i += 64)
                    moved |= (xm[i] != xm[i] + 0.1 * sigma * D[iaxis] * B[(size_t) iaxis * ld + i]);
            }
            if (!__any(moved)) flag = 8;
        }
        if (!flag && !(off & 512)) {   // NoEffectCoor
            int stuck = 0;
            for (int i = lane; i < n; i += 64) {
                const double sd = sep ? D[i] : sqrt(C[(size_t) i * ld + i]);
                stuck |= (xm[i] == xm[i] + 0.2 * sigma * sd);
            }
            if (__any(stuck)) flag = 9;
        }
    }
    if (lane == 0) {
        sc->it = it;
        sc->hist_head = head;
        sc->hist_len = len;
        sc->fbest = fbest;
        sc->fworst = fworst;
        sc->flag = flag;
        if (flag) sc->stop = 1;
        else if (sc->fev >= c.mfev) sc->stop = 2;
    }
}

__global__ __launch_bounds__(64) void cma_history_stop(CmaDev d, CmaConst c)
{
    history_stop_body(d, c, blockIdx.x, threadIdx.x);
}

// ---------------------------------------------------------------------------
// Small problems (n <= 16, lambda <= 64; the README example is n = 10, lambda = 20): `gens` whole
// generations in ONE launch, the population's whole state resident in LDS.
//
// A generation of such a problem is nine launches of a few microseconds of work each, and what
// each of them waits for is not the launch but its own dependent global-memory round trips
// (1-2 us each; measured: running the nine bodies inside one kernel on the HBM-resident state
// gained 7 %).  So one 256-thread workgroup per population copies the state into LDS, builds a
// LOCAL VIEW of it -- a CmaDev whose pointers are LDS addresses, laid out like population 0 --
// and runs the bodies of the kernels above on that view, back to back, a barrier where the
// kernel sequence has a kernel boundary: the same code, the same arithmetic, bit for bit (the
// bodies dereference generic pointers).  The state returns to HBM once, at the end of the launch.
// What is left of a generation is the serial chain of the 10 x 10 eigendecomposition.
// Used for up to SMALL_FUSED_MAXP populations (beyond, the kernel sequence fills the GPU better).
// grid (P), 256 threads, dynamic LDS = scratch of the largest body + small_state_doubles().
// ---------------------------------------------------------------------------
constexpr int SMALL_FUSED_MAXP = 512;

__host__ __device__ inline int small_state_doubles(const CmaConst &c)
{
    const int ld = c.ld, ld2 = ld * ld, lp = c.lambda_pad;
    return lp * ld                      // X
            + 2 * lp + lp               // f, zn2, rank + order (ints, two per double)
            + 5 * ld                    // xmean, xold, pc, ps, D
            + 5 * ld2                   // C, B, isc, BDp, ISp
            + c.mu_pad                  // S
            + c.splits * ld2 + c.splits * ld    // gram_part, mean_part
            + 2 * ((c.hlen + 1) & ~1)   // history rings (kept 16-byte aligned)
            + c.mu_pad                  // weights
            + 3 * ld                    // lower, upper, aux
            + 16;                       // CmaScal
}

__device__ inline void small_phase_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
}

// The phases as real (not inlined) functions: each gets its own register allocation.  Inlined
// into one kernel the nine bodies needed 256 VGPRs + 110 AGPRs and spilled ~500 SGPRs.
#define SMALL_NOINLINE __device__ __attribute__((noinline))
// (the scratch is named inside each function, not passed in: through a pointer argument the
// compiler loses the address space and turns every ds_read into a flat load -- the serial QL chain
// of the eigensolver ran 1.5x slower that way)
// The view's pointers are LDS addresses, but typed generic: as they stand every access through
// them is a FLAT load or store (and every wait a vmcnt(0) & lgkmcnt(0)).  Each phase therefore
// takes its own copy of the view and tells the compiler where the pointers point
// (llvm.assume(is.shared): InferAddressSpaces then rewrites their uses to ds_read / ds_write).
__device__ __forceinline__ CmaDev small_lds_view(const CmaDev &v)
{
    CmaDev w = v;
#if defined(__HIP_DEVICE_COMPILE__)
#define BBO_LDS(ptr) __builtin_assume(__builtin_amdgcn_is_shared((const void*) (w.ptr)))
#else
#define BBO_LDS(ptr) (void) 0      /* (the host pass only parses this function) */
#endif
    BBO_LDS(X); BBO_LDS(f); BBO_LDS(zn2); BBO_LDS(rank); BBO_LDS(order); BBO_LDS(xmean);
    BBO_LDS(xold); BBO_LDS(pc); BBO_LDS(ps); BBO_LDS(D); BBO_LDS(C); BBO_LDS(B); BBO_LDS(isc);
    BBO_LDS(BDp); BBO_LDS(ISp); BBO_LDS(S); BBO_LDS(gram_part); BBO_LDS(mean_part);
    BBO_LDS(hist_best); BBO_LDS(hist_kth); BBO_LDS(weights); BBO_LDS(lower); BBO_LDS(upper);
    BBO_LDS(aux); BBO_LDS(scal);
#undef BBO_LDS
    return w;
}
SMALL_NOINLINE void small_sample(const CmaDev &v, const CmaConst &c, int bx, int psub, bool first)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const CmaDev w = small_lds_view(v);
    sample_eval64_body<1, 8>(w, c, 0, bx, lds, psub, first);
}
SMALL_NOINLINE void small_rank(const CmaDev &v, const CmaConst &c, int lane)
{
    const CmaDev w = small_lds_view(v);
    rank_wave_body(w, c, 0, lane);
}
SMALL_NOINLINE void small_whiten(const CmaDev &v, const CmaConst &c, int mt)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const CmaDev w = small_lds_view(v);
    whiten_body<1>(w, c, 0, mt, lds);
}
SMALL_NOINLINE void small_gram(const CmaDev &v, const CmaConst &c, int s, int ldy)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const CmaDev w = small_lds_view(v);
    gram_body(w, c, 0, s, 0, ldy, lds);
}
SMALL_NOINLINE void small_paths(const CmaDev &v, const CmaConst &c)
{
    const CmaDev w = small_lds_view(v);
    paths_body(w, c, 0);
}
SMALL_NOINLINE void small_cov(const CmaDev &v, const CmaConst &c)
{
    const CmaDev w = small_lds_view(v);
    cov_body(w, c, 0, 0);
}
SMALL_NOINLINE void small_eigen(const CmaDev &v, const CmaConst &c, int lane)
{
    __shared__ __attribute__((aligned(16))) double eig_lds[EIGS_DOUBLES];
    const CmaDev w = small_lds_view(v);
    eigen_small_body(w, c, 0, lane, eig_lds, 0, 1);
}
SMALL_NOINLINE void small_stop(const CmaDev &v, const CmaConst &c, int lane)
{
    const CmaDev w = small_lds_view(v);
    history_stop_body(w, c, 0, lane);
}
#undef SMALL_NOINLINE

__device__ inline void small_copy(double *dst, const double *src, int count, int tid)
{
    for (int i = tid; i < count; i += 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void cma_small_generations(CmaDev d, CmaConst c_arg, int gens,
        int gram_ldy, int scratch_doubles)
{
    __shared__ CmaConst cs;                 // (by reference to the phase functions)
    if (threadIdx.x == 0) cs = c_arg;
    __syncthreads();
    const CmaConst &c = cs;
    const int p = blockIdx.x;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (pop_frozen(c, d.scal + p)) return;       // nothing to do: leave the state where it is
    const int ld = c.ld, ld2 = ld * ld, lp = c.lambda_pad;
    static_assert(sizeof(CmaScal) <= 16 * sizeof(double), "CmaScal outgrew its LDS slot");

    // ---- the local view: population p's state in LDS, addressed like population 0.  The view
    // itself lives in LDS too: thirty pointers held in scalar registers next to the kernel's own
    // arguments made the compiler spill ~500 of them (readlane / writelane on every use)
    __shared__ CmaDev vs;
    if (tid == 0) {
        CmaDev t = d;
        double *q = lds + scratch_doubles;
        t.X = q; q += lp * ld;
        t.f = q; q += lp;
        t.zn2 = q; q += lp;
        t.rank = reinterpret_cast<int*>(q);
        t.order = t.rank + lp; q += lp;
        t.xmean = q; q += ld;
        t.xold = q; q += ld;
        t.pc = q; q += ld;
        t.ps = q; q += ld;
        t.D = q; q += ld;
        t.C = q; q += ld2;
        t.B = q; q += ld2;
        t.isc = q; q += ld2;
        t.BDp = q; q += ld2;
        t.ISp = q; q += ld2;
        t.S = q; q += c.mu_pad;
        t.gram_part = q; q += c.splits * ld2;
        t.mean_part = q; q += c.splits * ld;
        t.hist_best = q; q += (c.hlen + 1) & ~1;
        t.hist_kth = q; q += (c.hlen + 1) & ~1;
        t.weights = q; q += c.mu_pad;
        t.lower = q; q += ld;
        t.upper = q; q += ld;
        t.aux = q; q += ld;
        t.scal = reinterpret_cast<CmaScal*>(q);
        if (d.zinject) t.zinject = d.zinject + (size_t) p * c.lambda * c.n;
        if (d.zrecord) t.zrecord = d.zrecord + (size_t) p * c.lambda * c.n;
        vs = t;
    }
    __syncthreads();
    const CmaDev &v = vs;
    double *wl = const_cast<double*>(v.weights), *lo = const_cast<double*>(v.lower);
    double *up = const_cast<double*>(v.upper), *ax = const_cast<double*>(v.aux);

    small_copy(v.xmean, d.xmean + (size_t) p * ld, ld, tid);
    small_copy(v.xold, d.xold + (size_t) p * ld, ld, tid);
    small_copy(v.pc, d.pc + (size_t) p * ld, ld, tid);
    small_copy(v.ps, d.ps + (size_t) p * ld, ld, tid);
    small_copy(v.D, d.D + (size_t) p * ld, ld, tid);
    small_copy(v.C, d.C + (size_t) p * ld2, ld2, tid);
    small_copy(v.B, d.B + (size_t) p * ld2, ld2, tid);
    small_copy(v.isc, d.isc + (size_t) p * ld2, ld2, tid);
    small_copy(v.BDp, d.BDp + (size_t) p * ld2, ld2, tid);
    small_copy(v.ISp, d.ISp + (size_t) p * ld2, ld2, tid);
    small_copy(v.hist_best, d.hist_best + (size_t) p * c.hlen, c.hlen, tid);
    small_copy(v.hist_kth, d.hist_kth + (size_t) p * c.hlen, c.hlen, tid);
    for (int i = tid; i < c.mu_pad; i += 256) wl[i] = i < c.mu ? d.weights[i] : 0.;
    small_copy(lo, d.lower, ld, tid);
    small_copy(up, d.upper, ld, tid);
    small_copy(ax, d.aux, ld, tid);
    // (the outputs of a generation that a LATER call may read before it is overwritten)
    small_copy(v.X, d.X + (size_t) p * lp * ld, lp * ld, tid);
    small_copy(v.f, d.f + (size_t) p * lp, lp, tid);
    small_copy(v.zn2, d.zn2 + (size_t) p * lp, lp, tid);
    small_copy(v.S, d.S + (size_t) p * c.mu_pad, c.mu_pad, tid);
    for (int i = tid; i < lp; i += 256) {
        v.rank[i] = d.rank[(size_t) p * lp + i];
        v.order[i] = d.order[(size_t) p * lp + i];
    }
    if (tid == 0) *v.scal = d.scal[p];
    small_phase_sync();

    const CmaScal *sc = v.scal;
    // (diagnostic: phase clocks of the last generation, bbo_set "eig_stamps"; 100 MHz)
#define SMALL_STAMP(slot) do { if (d.stamps && p == 0 && tid == 0) d.stamps[slot] = wall_clock64(); } while (0)
    for (int g = 0; g < gens; g++) {
        if (pop_frozen(c, sc)) break;       // (uniform: every thread reads the same LDS word)
        SMALL_STAMP(16);
        for (int bx = 0; bx * 64 < lp; bx++) {
            small_sample(v, c, bx, p, g == 0 && bx == 0);   // (the table stays for the launch)
            small_phase_sync();
        }
        SMALL_STAMP(17);
        if (wave == 0) small_rank(v, c, lane);
        small_phase_sync();
        SMALL_STAMP(18);
        if (c.variant == 1)
            for (int mt = 0; mt * 16 < c.mu_pad; mt++) {
                small_whiten(v, c, mt);
                small_phase_sync();
            }
        SMALL_STAMP(19);
        for (int s = 0; s < c.splits; s++) {
            small_gram(v, c, s, gram_ldy);
            small_phase_sync();
        }
        SMALL_STAMP(20);
        small_paths(v, c);
        small_phase_sync();
        SMALL_STAMP(21);
        small_cov(v, c);                    // n (n + 1) / 2 <= 136 entries: one pass
        small_phase_sync();
        SMALL_STAMP(22);
        if (wave == 0) small_eigen(v, c, lane);
        small_phase_sync();
        SMALL_STAMP(23);
        if (wave == 0) small_stop(v, c, lane);
        small_phase_sync();
        SMALL_STAMP(24);
    }
#undef SMALL_STAMP

    // ---- back to HBM ---------------------------------------------------------------------------
    small_copy(d.X + (size_t) p * lp * ld, v.X, lp * ld, tid);
    small_copy(d.f + (size_t) p * lp, v.f, lp, tid);
    small_copy(d.zn2 + (size_t) p * lp, v.zn2, lp, tid);
    for (int i = tid; i < lp; i += 256) {
        d.rank[(size_t) p * lp + i] = v.rank[i];
        d.order[(size_t) p * lp + i] = v.order[i];
    }
    small_copy(d.xmean + (size_t) p * ld, v.xmean, ld, tid);
    small_copy(d.xold + (size_t) p * ld, v.xold, ld, tid);
    small_copy(d.pc + (size_t) p * ld, v.pc, ld, tid);
    small_copy(d.ps + (size_t) p * ld, v.ps, ld, tid);
    small_copy(d.D + (size_t) p * ld, v.D, ld, tid);
    small_copy(d.C + (size_t) p * ld2, v.C, ld2, tid);
    small_copy(d.B + (size_t) p * ld2, v.B, ld2, tid);
    small_copy(d.isc + (size_t) p * ld2, v.isc, ld2, tid);
    small_copy(d.BDp + (size_t) p * ld2, v.BDp, ld2, tid);
    small_copy(d.ISp + (size_t) p * ld2, v.ISp, ld2, tid);
    small_copy(d.S + (size_t) p * c.mu_pad, v.S, c.mu_pad, tid);
    small_copy(d.gram_part + (size_t) p * c.splits * ld2, v.gram_part, c.splits * ld2, tid);
    small_copy(d.mean_part + (size_t) p * c.splits * ld, v.mean_part, c.splits * ld, tid);
    small_copy(d.hist_best + (size_t) p * c.hlen, v.hist_best, c.hlen, tid);
    small_copy(d.hist_kth + (size_t) p * c.hlen, v.hist_kth, c.hlen, tid);
    if (tid == 0) d.scal[p] = *v.scal;
}

} // namespace bbo
