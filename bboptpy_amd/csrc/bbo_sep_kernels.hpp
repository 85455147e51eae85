// bbo_sep_kernels.hpp -- separable CMA-ES (diagonal covariance) as gfx950 kernels.
//
// Reference: SepCmaes, src/multivariate/cma/sep_cmaes.cpp:41-206 (Ros & Hansen 2008).  With a
// diagonal C there is no matrix product left: every kernel is a streaming pass over the
// population, priced in HBM bytes.
//
//   kernel             reference lines                       bytes per candidate
//   sep_sample_eval<G> sep_cmaes.cpp:70-80 + base_cmaes.cpp:214-217   8n + 8 written
//   (cma_rank / cma_rank_sort: shared with the full-covariance engine)
//   sep_moments        :85-95 (weighted mean), :121-129 (rank-mu sum)  8n read per SELECTED row
//   sep_paths          :97-134 + base_cmaes.cpp:176-189        latency (1 workgroup / population)
//   sep_history_stop   base_cmaes.cpp:191-209, sep_cmaes.cpp:137-206   latency
//
// The engine state is CmaDev with D = _diagd and csep = _c; B, C, C^-1/2 do not exist.
#pragma once

#include "bbo_cma_kernels.hpp"

namespace bbo {

// ---------------------------------------------------------------------------
// sample + evaluate: x = m + sigma d z (sep_cmaes.cpp:73), G lanes per candidate (G | 64, so a
// row is drawn and evaluated by ONE wavefront: no workgroup barrier after the table is staged),
// the row kept in LDS for the objective (at row_swizzle(j): a lane writes the four columns of a
// Philox call 4 apart, which unswizzled lands 16 lanes on each LDS bank).  A workgroup takes chunks of T/G candidates, blockIdx.x,
// blockIdx.x + gridDim.x, ...: the 16 KB table of the normal generator is staged once per
// workgroup, so the host sizes the grid for a few thousand workgroups, not one per chunk.
// The draw is the two-step one (normal_quad_fast / normal_quad_settle, bbo_rng.hpp).
// grid (<= ceil(lambda_pad / (T/G)), P), T threads, dynamic LDS (T/G) * ld doubles; ld <= 64 G.
// Normals: the same (candidate, column) -> Philox mapping as the full-covariance sampler
// (cma_quad_col0), so the oracle's statement covers both.
// ---------------------------------------------------------------------------
// FULL: n == ld, no box, lambda == lambda_pad, no injected / recorded normals -- the guards of
// the general form (a third of its vector instructions) are gone at compile time.
template<int G, int T = 256, bool FULL = false>
__global__ __launch_bounds__(T) void sep_sample_eval(CmaDev d, CmaConst c)
{
    static_assert(64 % G == 0, "a candidate's lanes must sit in one wavefront");
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int R = T / G;
    const int tid = threadIdx.x, r = tid / G, g = tid % G;
    const int ld = c.ld;
    double *xr = lds + (size_t) r * ld;
    const int gen = sc->it;
    const double sigma = sc->sigma;
    const double *xm = d.xmean + (size_t) p * ld, *dd = d.D + (size_t) p * ld;
    const uint32_t sw = stream_word(STREAM_CMA_NORMAL, (uint32_t) p);
    __shared__ double2 ntab[NORMAL_TABLE_N];
    normal_table_fill(ntab, tid, T);
    __syncthreads();
    const int chunks = (c.lambda_pad + R - 1) / R;
    for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const int row = chunk * R + r;
        double *Xp = d.X + ((size_t) p * c.lambda_pad + row) * ld;
        if (row < c.lambda_pad) {
            uint64_t pend = 0;                       // 4 bits per call, <= 16 calls per lane
            int it = 0;
            for (int q = g; q < ld / 4; q += G, it++) {
                double z[4];
                if (FULL)
                    pend |= (uint64_t) normal_quad_fast(c.seed, (uint32_t) row, (uint32_t) q,
                            (uint32_t) gen, sw, ntab, z[0], z[1], z[2], z[3]) << (4 * it);
                else
                    pend |= (uint64_t) cma_draw_quad_fast(d, c, p, row, q, gen, sw, ntab, z)
                            << (4 * it);
                const int j0 = cma_quad_col0(q);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = j0 + 4 * i;
                    double v = 0.;
                    if (FULL) {
                        v = xm[j] + sigma * dd[j] * z[i];
                    } else if (j < c.n) {
                        v = xm[j] + sigma * dd[j] * z[i];
                        if (c.bound) v = fmax(d.lower[j], fmin(v, d.upper[j]));
                    }
                    xr[row_swizzle(j)] = v;
                    Xp[j] = v;
                }
            }
            while (pend) {                           // the draws the fast step left open
                const int b = __ffsll((unsigned long long) pend) - 1;
                pend &= pend - 1;
                const int q = g + G * (b >> 2), j = cma_quad_col0(q) + 4 * (b & 3);
                const double z = FULL ? normal_quad_settle(c.seed, (uint32_t) row, (uint32_t) q,
                                                (uint32_t) (b & 3), (uint32_t) gen, sw, ntab,
                                                zig_global_f())
                                      : cma_settle_draw(d, c, p, row, q, b & 3, gen, sw, ntab);
                double v = xm[j] + sigma * dd[j] * z;
                if (!FULL && c.bound) v = fmax(d.lower[j], fmin(v, d.upper[j]));
                xr[row_swizzle(j)] = v;
                Xp[j] = v;
            }
        }
        cma_wave_sync();
        if (c.obj >= 0 && row < c.lambda_pad) {
            double f = eval_row_group<G, true>(c.obj, c.n, xr, d.aux, g);
            if (g == 0) {
                if ((!FULL && !(row < c.lambda)) || f != f) f = BBO_INF;
                d.f[(size_t) p * c.lambda_pad + row] = f;
            }
        }
        cma_wave_sync();                     // the rows are reused by the next chunk
    }
}

// ---------------------------------------------------------------------------
// first and second weighted moments of the mu best candidates, slab s of the ranks:
//   mean_part[s][j] = sum_k w_k x_k[j]                      (sep_cmaes.cpp:88-93)
//   gram_part[s][j] = sum_k w_k ((x_k[j] - m_j) / sigma)^2  (:124-128)
// one thread per column, rows gathered through `order` (each row read = one coalesced sweep).
// grid (splits, ceil(ld/256), P), 256 threads
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sep_moments(CmaDev d, CmaConst c)
{
    const int p = blockIdx.z, s = blockIdx.x;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    const int j = blockIdx.y * 256 + threadIdx.x, ld = c.ld;
    if (j >= ld) return;
    const int per = (c.mu + c.splits - 1) / c.splits;
    const int k0 = s * per, k1 = min(c.mu, k0 + per);
    const double *Xp = d.X + (size_t) p * c.lambda_pad * ld;
    const int *order = d.order + (size_t) p * c.lambda_pad;
    const double xo = d.xmean[(size_t) p * ld + j];        // the mean has not moved yet
    const double isig = 1. / sc->sigma;
    double m1 = 0., m2 = 0.;
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = Xp[(size_t) order[k + u] * ld + j];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const double w = d.weights[k + u], di = (x[u] - xo) * isig;
            m1 += w * x[u];
            m2 += w * (di * di);
        }
    }
    for (; k < k1; k++) {
        const double x = Xp[(size_t) order[k] * ld + j], w = d.weights[k];
        const double di = (x - xo) * isig;
        m1 += w * x;
        m2 += w * (di * di);
    }
    d.mean_part[((size_t) p * c.splits + s) * ld + j] = j < c.n ? m1 : 0.;
    d.gram_part[((size_t) p * c.splits + s) * ld + j] = j < c.n ? m2 : 0.;
}

// ---------------------------------------------------------------------------
// mean, ps, hsig, pc, diagonal C, d = sqrt(C), sigma (sep_cmaes.cpp:85-134): one workgroup of
// 256 threads per population
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sep_paths(CmaDev d, CmaConst c)
{
    const int p = blockIdx.x;
    CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    __shared__ double red[4];
    const int tid = threadIdx.x, ld = c.ld;
    double *xmean = d.xmean + (size_t) p * ld, *xold = d.xold + (size_t) p * ld;
    double *ps = d.ps + (size_t) p * ld, *pc = d.pc + (size_t) p * ld;
    double *cs_ = d.csep + (size_t) p * ld, *dd = d.D + (size_t) p * ld;
    const double sigma = sc->sigma;
    const double csc = sqrt(c.cs * (2. - c.cs) * c.mueff);
    double ssq = 0.;
    for (int j = tid; j < ld; j += 256) {
        double sum = 0.;
        for (int s = 0; s < c.splits; s++) sum += d.mean_part[((size_t) p * c.splits + s) * ld + j];
        const double xo = xmean[j];
        double xn = 0., v = 0.;
        if (j < c.n) {
            xn = sum;
            if (c.bound) xn = fmax(d.lower[j], fmin(xn, d.upper[j]));
            // (the reference scales the step by _c[i], not by 1/_diagd[i]: kept, :100-101)
            v = ps[j] * (1. - c.cs);
            v += csc * cs_[j] * (xn - xo) / sigma;
        }
        xold[j] = xo;
        xmean[j] = xn;
        ps[j] = v;
        ssq += v * v;
    }
    ssq = wave_sum(ssq);
    if ((tid & 63) == 0) red[tid >> 6] = ssq;
    __syncthreads();
    const double pslen = sqrt(red[0] + red[1] + red[2] + red[3]);
    const double denom = 1. - pow(1. - c.cs, 2. * sc->fev / c.lambda);
    const int hsig = (pslen / sqrt(denom) / c.chi < 1.4 + 2. / (c.n + 1.)) ? 1 : 0;
    const double ccc = sqrt(c.cc * (2. - c.cc) * c.mueff);
    for (int j = tid; j < ld; j += 256) {
        if (j < c.n) {
            const double pcj = (1. - c.cc) * pc[j] + hsig * ccc * (xmean[j] - xold[j]) / sigma;
            pc[j] = pcj;
            double m2 = 0.;
            for (int s = 0; s < c.splits; s++)
                m2 += d.gram_part[((size_t) p * c.splits + s) * ld + j];
            const double cn = (1. - c.ccov) * cs_[j] + (c.ccov / c.mueff) * pcj * pcj
                    + c.ccov * (1. - 1. / c.mueff) * m2;
            cs_[j] = cn;
            dd[j] = sqrt(cn);
        } else {
            pc[j] = 0.;
            cs_[j] = 1.;
            dd[j] = 1.;
        }
    }
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

} // namespace bbo
