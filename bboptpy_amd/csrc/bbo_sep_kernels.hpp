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

// sum of `count` values `stride` apart in their order, eight loads in flight at a time (summed
// straight off a loop every addition waits for its own round trip; the same additions in the same
// order)
__device__ inline double slab_sum(const double *v, int count, size_t stride)
{
    double sum = 0.;
    int s = 0;
    for (; s + 8 <= count; s += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = v[(size_t) (s + u) * stride];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; u++) sum += x[u];
    }
    for (; s < count; s++) sum += v[(size_t) s * stride];
    return sum;
}


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
// sample + evaluate for the objectives that are a SUM OF PER-COORDINATE TERMS (sphere, ellipsoid,
// Rastrigin, cigar, discus, different powers), lean case only (n == ld, no box, lambda ==
// lambda_pad, nothing injected or recorded): a lane adds the term of every coordinate it draws to
// its own partial sum the moment the draw is settled, the 64 lanes of the row meet in one
// butterfly -- the row never goes through LDS.  sep_sample_eval keeps a row in LDS only because a
// lane draws the columns of its Philox calls (cma_quad_col0) and the objective walks the row in
// another order; 8 rows x 8 KB + the generator's 16 KB table filled half the LDS of a CU and held
// the kernel to 16 wavefronts per CU, and it is latency the kernel is short of (with 8 wavefronts
// per CU it ran at half the speed: round 4, two rows per wavefront tried for a dense settle).
// The terms are added in the lane's draw order, not in the strided order of eval_row_group: f
// agrees with it and with the oracle to rounding (1e-13 relative), X is bit-identical.
// grid (<= lambda_pad / (SEP_K T / 64), P), T threads, no dynamic LDS
// ---------------------------------------------------------------------------
__host__ __device__ inline bool sep_sum_objective(int obj)
{
    return obj == OBJ_SPHERE || obj == OBJ_ELLIPSOID || obj == OBJ_RASTRIGIN || obj == OBJ_CIGAR
            || obj == OBJ_DISCUS || obj == OBJ_DIFFPOW;
}

template<int OBJ>
__device__ __forceinline__ void sep_term(int j, double v, const double *aux, double &a, double &b)
{
    if (OBJ == OBJ_SPHERE) a += v * v;
    else if (OBJ == OBJ_ELLIPSOID) a += aux[j] * (v * v);
    else if (OBJ == OBJ_RASTRIGIN) a += v * v - 10. * cos_2pi(v);
    else if (OBJ == OBJ_DIFFPOW) a += pow(fabs(v), aux[j]);
    else {                                   // cigar / discus: coordinate 0 apart
        if (j > 0) a += v * v;
        else b = v * v;
    }
}

// sum of v over the wavefront, the same total in every lane (DPP row rotations + one readlane per
// 16-lane row: a fixed order)
__device__ inline double sep_wave_sum(double v)
{
    v = row16_sum(v);
    auto rl = [](double x, int l) {
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l),
                __builtin_amdgcn_readlane(__double2loint(x), l));
    };
    return ((rl(v, 0) + rl(v, 16)) + rl(v, 32)) + rl(v, 48);
}

// A wavefront takes SEP_K consecutive rows at a time: the fast step of all of them first, then
// ONE dense settle for what they left open.  A settle round costs the wavefront ~370 instructions
// whether one lane or sixty-four need it; settled row by row (sep_sample_eval) a round runs with
// four or five lanes active, 1.2 rounds per row -- 105 of the 235 instructions per Philox call.
// Here the open draws of the batch (0.43 % of 4 K n: ~18 at n = 1024) are gathered into a list
// (ballot + prefix count per pass, as many passes as the busiest lane has open draws) and settled
// by CONSECUTIVE lanes, one round for up to 64 of them; the settled value goes to HBM and its
// term to the row's sum through one wavefront reduction per row -- any lane can settle any draw
// because nothing has to come back to a register of the lane that drew it.
constexpr int SEP_K = 4;
constexpr int SEP_PLIST = 128;

// NC > 0: ld = 256 NC exactly and the lane keeps the mean and sigma d of ITS 4 NC columns in
// registers for all rows (the columns a lane draws depend on the lane and the call only).  Read
// next to their use -- load, wait, multiply, store, four times per call, each wait behind the
// store in front of it on the one in-order memory counter -- they were what the kernel waited for
// (round 4, from the ISA: the settle work had gone and the time had not).  NC = 0: any ld, loads.
template<int OBJ, int T, int NC>
__device__ __forceinline__ void sep_sample_sum_body(const CmaDev &d, const CmaConst &c,
        const double2 *ntab, int *plist)
{
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    constexpr int R = T / 64;
    const int tid = threadIdx.x, r = tid >> 6, g = tid & 63;
    const int ld = c.ld;
    const int gen = sc->it;
    const double sigma = sc->sigma;
    const double *xm = d.xmean + (size_t) p * ld, *dd = d.D + (size_t) p * ld;
    const uint32_t sw = stream_word(STREAM_CMA_NORMAL, (uint32_t) p);
    int *pl = plist + r * SEP_PLIST;
    const int chunks = c.lambda_pad / (R * SEP_K);
    constexpr int NCR = NC > 0 ? NC : 1;
    double mreg[NCR][4], sreg[NCR][4];
    if (NC > 0) {
#pragma unroll
        for (int it = 0; it < NCR; it++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = cma_quad_col0(g + 64 * it) + 4 * i;
                mreg[it][i] = xm[j];
                sreg[it][i] = sigma * dd[j];
            }
    }
    for (int chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const int row0 = (chunk * R + r) * SEP_K;
        double *X0 = d.X + ((size_t) p * c.lambda_pad + row0) * ld;
        double a[SEP_K], b[SEP_K];
        uint32_t pend[SEP_K];                        // 4 bits per call, <= 8 calls per lane and row
#pragma unroll
        for (int k = 0; k < SEP_K; k++) {
            a[k] = b[k] = 0.;
            pend[k] = 0;
            double *Xp = X0 + (size_t) k * ld;
            auto one_call = [&](int q, int it, const double (&mm)[4], const double (&ss)[4]) {
                double z[4];
                const uint32_t open = normal_quad_fast(c.seed, (uint32_t) (row0 + k), (uint32_t) q,
                        (uint32_t) gen, sw, ntab, z[0], z[1], z[2], z[3]);
                pend[k] |= open << (4 * it);
                const int j0 = cma_quad_col0(q);
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = j0 + 4 * i;
                    const double v = mm[i] + ss[i] * z[i];
                    Xp[j] = v;
                    // (an open draw's term is added when it is settled: nothing to take back)
                    double ta = 0., tb = b[k];
                    sep_term<OBJ>(j, v, d.aux, ta, tb);
                    const bool ok = !((open >> i) & 1u);
                    a[k] += ok ? ta : 0.;
                    b[k] = ok ? tb : b[k];
                }
                // (one call at a time: left alone the scheduler interleaves the sixteen calls of a
                // batch and needs 512 registers for it -- one wavefront per SIMD)
                __builtin_amdgcn_sched_barrier(0);
            };
            if (NC > 0) {
#pragma unroll
                for (int it = 0; it < NCR; it++) one_call(g + 64 * it, it, mreg[it], sreg[it]);
            } else {
                int it = 0;
                for (int q = g; q < ld / 4; q += 64, it++) {
                    double mm[4], ss[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int j = cma_quad_col0(q) + 4 * i;
                        mm[i] = xm[j];
                        ss[i] = sigma * dd[j];
                    }
                    one_call(q, it, mm, ss);
                }
            }
        }
        // ---- the open draws of the batch, densely ---------------------------------------------
        for (;;) {
            int cnt = 0;
            for (;;) {                               // gather up to SEP_PLIST of them
                int k = -1;
#pragma unroll
                for (int kk = SEP_K - 1; kk >= 0; kk--) k = pend[kk] ? kk : k;
                const bool has = k >= 0;
                const unsigned long long mk = __ballot(has);
                if (!mk) break;
                const int slot = cnt + __popcll(mk & ((1ull << g) - 1ull));
                if (has && slot < SEP_PLIST) {
                    uint32_t pk = 0;
#pragma unroll
                    for (int kk = 0; kk < SEP_K; kk++) pk = kk == k ? pend[kk] : pk;
                    const int bit = __ffs(pk) - 1;
                    pl[slot] = (g << 16) | (k << 8) | bit;
#pragma unroll
                    for (int kk = 0; kk < SEP_K; kk++) pend[kk] = kk == k ? (pk & (pk - 1)) : pend[kk];
                }
                cnt += __popcll(mk);
                if (cnt >= SEP_PLIST) break;
            }
            if (cnt == 0) break;
            cnt = min(cnt, SEP_PLIST);
            cma_wave_sync();
            for (int i0 = 0; i0 < cnt; i0 += 64) {
                double ta = 0., tb = 0.;
                int ek = -1, ej = -1;
                if (i0 + g < cnt) {
                    const int e = pl[i0 + g];
                    const int sl = e >> 16, bit = e & 255;
                    ek = (e >> 8) & 255;
                    const int q = sl + 64 * (bit >> 2);
                    ej = cma_quad_col0(q) + 4 * (bit & 3);
                    const double z = normal_quad_settle(c.seed, (uint32_t) (row0 + ek), (uint32_t) q,
                            (uint32_t) (bit & 3), (uint32_t) gen, sw, ntab, zig_global_f());
                    const double v = xm[ej] + sigma * dd[ej] * z;
                    X0[(size_t) ek * ld + ej] = v;
                    sep_term<OBJ>(ej, v, d.aux, ta, tb);
                }
#pragma unroll
                for (int kk = 0; kk < SEP_K; kk++) {
                    const double sa = sep_wave_sum(ek == kk ? ta : 0.);
                    if (g == 0) a[kk] += sa;
                    if (OBJ == OBJ_CIGAR || OBJ == OBJ_DISCUS) {
                        const double sb = sep_wave_sum((ek == kk && ej == 0) ? tb : 0.);
                        if (g == 0) b[kk] += sb;        // (lane 0 draws column 0: its b was left 0)
                    }
                }
            }
            cma_wave_sync();
        }
#pragma unroll
        for (int k = 0; k < SEP_K; k++) {
            const double sa = group_sum<64>(a[k]);
            double f = sa;
            if (OBJ == OBJ_RASTRIGIN) f = 10. * c.n + sa;
            if (OBJ == OBJ_CIGAR || OBJ == OBJ_DISCUS) {
                const double sb = group_sum<64>(b[k]);   // (one lane holds x_0^2, the others 0)
                f = OBJ == OBJ_CIGAR ? sb + 1.0e6 * sa : 1.0e6 * sb + sa;
            }
            if (g == 0) d.f[(size_t) p * c.lambda_pad + row0 + k] = f != f ? BBO_INF : f;
        }
    }
}

// (a kernel per objective: in one kernel behind a switch every objective paid the register
// allocation of the hungriest -- pow and the cosine, inlined per coordinate)
template<int T, int NC, int OBJ>
__global__ __launch_bounds__(T) void sep_sample_sum(CmaDev d, CmaConst c)
{
    if (pop_frozen(c, d.scal + blockIdx.y)) return;
    __shared__ double2 ntab[NORMAL_TABLE_N];
    __shared__ int plist[(T / 64) * SEP_PLIST];
    normal_table_fill(ntab, threadIdx.x, T);
    __syncthreads();
    sep_sample_sum_body<OBJ, T, NC>(d, c, ntab, plist);
}

// ---------------------------------------------------------------------------
// first and second weighted moments of the mu best candidates, slab s of the ranks:
//   mean_part[s][j] = sum_k w_k x_k[j]                      (sep_cmaes.cpp:88-93)
//   gram_part[s][j] = sum_k w_k ((x_k[j] - m_j) / sigma)^2  (:124-128)
// two columns per thread, rows gathered through `order` (each row read = one coalesced sweep).
// grid (splits, ceil(ld/512), P), 256 threads
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sep_moments(CmaDev d, CmaConst c)
{
    const int p = blockIdx.z, s = blockIdx.x;
    const CmaScal *sc = d.scal + p;
    if (pop_frozen(c, sc)) return;
    // (round 5: two columns per thread as one 16-byte load, eight rows in flight -- 64 KB per
    // workgroup instead of 8; every column's sums run over the rows in the same order as before.
    // It bought 2 %: the kernel reads its 1.07 GB of gathered 8 KB rows at 4 TB/s whatever the
    // workgroup size (64 ... 512 threads measured) or the depth of the loads)
    const int j = 2 * (blockIdx.y * blockDim.x + threadIdx.x), ld = c.ld;
    if (j >= ld) return;
    const int per = (c.mu + c.splits - 1) / c.splits;
    const int k0 = s * per, k1 = min(c.mu, k0 + per);
    const double *Xp = d.X + (size_t) p * c.lambda_pad * ld;
    const int *order = d.order + (size_t) p * c.lambda_pad;
    const double2 xo = *reinterpret_cast<const double2*>(&d.xmean[(size_t) p * ld + j]);   // the mean has not moved yet
    const double isig = 1. / sc->sigma;
    double2 m1 = make_double2(0., 0.), m2 = make_double2(0., 0.);
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        double2 x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = *reinterpret_cast<const double2*>(&Xp[(size_t) order[k + u] * ld + j]);
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const double w = d.weights[k + u];
            const double d0 = (x[u].x - xo.x) * isig, d1 = (x[u].y - xo.y) * isig;
            m1.x += w * x[u].x;
            m1.y += w * x[u].y;
            m2.x += w * (d0 * d0);
            m2.y += w * (d1 * d1);
        }
    }
    for (; k < k1; k++) {
        const double2 x = *reinterpret_cast<const double2*>(&Xp[(size_t) order[k] * ld + j]);
        const double w = d.weights[k];
        const double d0 = (x.x - xo.x) * isig, d1 = (x.y - xo.y) * isig;
        m1.x += w * x.x;
        m1.y += w * x.y;
        m2.x += w * (d0 * d0);
        m2.y += w * (d1 * d1);
    }
    double *mp = d.mean_part + ((size_t) p * c.splits + s) * ld + j;
    double *gp = d.gram_part + ((size_t) p * c.splits + s) * ld + j;
    *reinterpret_cast<double2*>(mp) = make_double2(j < c.n ? m1.x : 0., j + 1 < c.n ? m1.y : 0.);
    *reinterpret_cast<double2*>(gp) = make_double2(j < c.n ? m2.x : 0., j + 1 < c.n ? m2.y : 0.);
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
        const double sum = slab_sum(d.mean_part + (size_t) p * c.splits * ld + j, c.splits, ld);
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
            const double m2 = slab_sum(d.gram_part + (size_t) p * c.splits * ld + j, c.splits, ld);
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
