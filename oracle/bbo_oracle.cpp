/*
 * oracle/bbo_oracle.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement (flat arrays, one translation unit, C ABI for ctypes) of the
 * reference's per-generation hot path -- the CHECKER the HIP path is compared
 * with.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product (bboptpy_amd/) never does.
 *
 * Parity status: PINNED.  With its mt19937 replay generator this file is
 * checked bit-for-bit against the real reference compiled into
 * oracle/_ref/libbbo_ref.so (tests/test_oracle_vs_reference.py, development
 * container) and against the fixtures that build wrote to tests/golden/
 * (tests/test_oracle_golden.py, everywhere).
 *
 * What follows which reference lines:
 *   rng:       libstdc++-11 <random> behaviour as the reference consumes it,
 *              SURVEY.md Appendix C (src/random.hpp:311-337,639-642,677-680)
 *   cma_*:     src/multivariate/cma/base_cmaes.cpp:54-238, cmaes.cpp:44-478,
 *              active_cmaes.cpp:42-168
 *   blas bits: src/blas.cpp:52-73 (daxpym), :106-128 (dscalm), :154-181 (dnrm2)
 *   de_*:      src/multivariate/de/shade.cpp:56-298, jade.cpp:64-294
 *   pso_*:     src/multivariate/pso/apso.cpp:48-452
 *   restart_*: src/multivariate/cma/bipop_cmaes.cpp:61-267, ipop_cmaes.cpp:65-189
 *
 * Two execution modes where the GPU cannot be sequential like the reference:
 *   async = reference-faithful (in-place replacement inside the i-loop),
 *   sync  = generation-synchronous (what the HIP kernels compute), see
 *           DESIGN.md "sync semantics".
 * Three random sources: RNG_MT (replay of the reference's global mt19937),
 * RNG_PHILOX (the device generator, oracle/philox.h), RNG_INJECT (normals
 * supplied by the caller, CMA only).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <stdexcept>
#include <functional>
#include <vector>

#include "objectives.h"
#include "philox.h"

namespace {

/* ======================================================================== */
/* mt19937 + the four libstdc++-11 samplers the reference uses               */
/* ======================================================================== */
struct Mt19937 {
    uint32_t s[624];
    int pos;
    void seed(uint32_t v)
    {
        s[0] = v;
        for (int i = 1; i < 624; i++)
            s[i] = 1812433253u * (s[i - 1] ^ (s[i - 1] >> 30)) + (uint32_t) i;
        pos = 624;
    }
    void refill()
    {
        for (int i = 0; i < 624; i++) {
            const uint32_t y = (s[i] & 0x80000000u) | (s[(i + 1) % 624] & 0x7fffffffu);
            s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        pos = 0;
    }
    uint32_t next()
    {
        if (pos >= 624) refill();
        uint32_t y = s[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

Mt19937 g_mt;   /* process-global, like random.hpp:677-680 */
bool g_mt_seeded = false;

Mt19937& engine()
{
    if (!g_mt_seeded) {
        g_mt.seed(5489u);
        g_mt_seeded = true;
    }
    return g_mt;
}

/* generate_canonical<double,53> over a 32-bit engine: two draws, low word first */
double mt_canonical()
{
    Mt19937 &e = engine();
    const double lo = (double) e.next();
    const double hi = (double) e.next();
    const double sum = lo + hi * 4294967296.0;
    double r = sum / 18446744073709551616.0;
    if (r >= 1.) r = std::nextafter(1., 0.);
    return r;
}

/* Random::get(a, b) on doubles, random.hpp:329-337 */
double mt_uniform(double a, double b)
{
    if (!(a < b)) std::swap(a, b);
    return mt_canonical() * (b - a) + a;
}

/* Random::get(i, j) on ints, random.hpp:311-319 -> uniform_int_distribution
 * (Lemire multiply-shift with rejection, libstdc++-11 uniform_int_dist.h) */
int mt_int(int a, int b)
{
    if (!(a < b)) std::swap(a, b);
    const uint32_t urange = (uint32_t) b - (uint32_t) a;
    Mt19937 &e = engine();
    if (urange == 0xffffffffu) return (int) ((uint32_t) a + e.next());
    const uint32_t range = urange + 1u;
    uint64_t prod = (uint64_t) e.next() * (uint64_t) range;
    uint32_t low = (uint32_t) prod;
    if (low < range) {
        const uint32_t thresh = (0u - range) % range;
        while (low < thresh) {
            prod = (uint64_t) e.next() * (uint64_t) range;
            low = (uint32_t) prod;
        }
    }
    return (int) ((uint32_t) a + (uint32_t) (prod >> 32));
}

/* std::normal_distribution<>: Marsaglia polar, returns y*m and caches x*m */
struct NormalCache {
    bool have = false;
    double saved = 0.;
    double draw()
    {
        if (have) {
            have = false;
            return saved;
        }
        double x, y, r2;
        do {
            x = 2. * mt_canonical() - 1.;
            y = 2. * mt_canonical() - 1.;
            r2 = x * x + y * y;
        } while (r2 > 1. || r2 == 0.);
        const double m = std::sqrt(-2. * std::log(r2) / r2);
        saved = x * m;
        have = true;
        return y * m;
    }
};

NormalCache g_test_normal;

enum { RNG_MT = 0, RNG_PHILOX = 1, RNG_INJECT = 2 };

/* ======================================================================== */
/* level-1 helpers with the reference's rounding order                       */
/* ======================================================================== */
/* scaled sum of squares, blas.cpp:154-181 */
double nrm2(int n, const double *x)
{
    if (n < 1) return 0.;
    if (n == 1) return std::fabs(x[0]);
    double scale = 0., ssq = 1.;
    for (int i = 0; i < n; i++) {
        if (x[i] != 0.) {
            const double a = std::fabs(x[i]);
            if (scale < a) {
                ssq = 1. + ssq * (scale / a) * (scale / a);
                scale = a;
            } else {
                ssq = ssq + (a / scale) * (a / scale);
            }
        }
    }
    return scale * std::sqrt(ssq);
}

/* element-wise, so the reference's unrolling (blas.cpp:106-128, :52-73) does
 * not change any rounding: plain loops are bit-identical */
void scal(int n, double a, double *x)
{
    for (int i = 0; i < n; i++) x[i] *= a;
}

void axpy(int n, double a, const double *x, double *y)
{
    if (n <= 0 || a == 0.) return;
    for (int i = 0; i < n; i++) y[i] += a * x[i];
}

/* std::inner_product(first, last, first2, 0.) */
double dot(int n, const double *a, const double *b)
{
    double s = 0.;
    for (int i = 0; i < n; i++) s = s + a[i] * b[i];
    return s;
}

/* math_utils.tpp:27-29 */
double sign_of(double x, double y)
{
    return y >= 0. ? std::fabs(x) : -std::fabs(x);
}

/* math_utils.tpp:39-51.  NB: inside cmaes.cpp the unqualified call
 * `hypot(p, 1.)` resolves to the C library ::hypot (a non-template beats the
 * template in overload resolution), so tql2 below uses ::hypot; this template
 * form is kept for the variants that call it explicitly. */
double hypot_tpl(double a, double b)
{
    const double absa = std::fabs(a), absb = std::fabs(b);
    double r = 0.;
    if (absa > absb) {
        r = b / a;
        r = absa * std::sqrt(1. + r * r);
    } else if (b != 0.) {
        r = a / b;
        r = absb * std::sqrt(1. + r * r);
    }
    return r;
}

int g_hypot_mode = 0;   /* 0 = ::hypot (what the compiled reference does), 1 = template */

double ref_hypot(double a, double b)
{
    return g_hypot_mode == 0 ? ::hypot(a, b) : hypot_tpl(a, b);
}

/* ======================================================================== */
/* CMA-ES (plain and active)                                                 */
/* ======================================================================== */
struct Ring {
    int cap = 0, head = -1, len = 0;
    std::vector<double> v;
    void reset(int c)
    {
        cap = c;
        head = -1;
        len = 0;
        v.assign(c, 0.);
    }
    void add(double x)
    {
        head = (head + 1) % cap;
        v[head] = x;
        if (len < cap) len++;
    }
    double back(int i) const { return v[(cap + head - i) % cap]; }
};

struct Ranked {
    int index;
    double value;
};

struct Cma {
    /* constructor parameters */
    int variant = 1;          /* 0 = Cmaes, 1 = ActiveCmaes, 2 = SepCmaes */
    int mfev = 0, lambda = 0;
    double tol = 0., sigma0 = 2., alphacov = 2., eigenrate = 0.25;
    bool bound = false, adaptpop = false, adaptit = false;
    /* problem */
    int obj = 0, n = 0;
    std::vector<double> lower, upper, aux;
    /* strategy constants and counters */
    int mu = 0, mit = 0, it = 0, hlen = 0, ik = 0, fev = 0;
    double mueff = 0., cc = 0., cs = 0., c1 = 0., cmu = 0., damps = 0., chi = 0.,
            sigma = 0., fbest = 0., fworst = 0.;
    Ring best, kth;
    int ibw[4] = { 0, 0, 0, 0 };
    double ybw[4] = { 0., 0., 0., 0. };
    std::vector<double> xmean, xold, weights, tmp, pc, ps, arx;
    std::vector<Ranked> fit;
    /* full-covariance state */
    int flag = 0, eigenlastev = 0;
    double eigenfreq = 0.;
    std::vector<double> D, B, C, invsqrtC;
    /* active-CMA additions */
    double cm = 1., cneg = 0., alphaold = 0.5;
    std::vector<double> ycoeff;
    /* sep-CMA additions (sep_cmaes.h:36-41): diagonal covariance; D doubles as _diagd */
    bool adjustlr = false;
    double ccov = 0.;
    std::vector<double> csep;
    /* randomness */
    int rng_mode = RNG_MT;
    uint64_t seed = 0;
    NormalCache Z;
    std::vector<double> zinject, zlast;
    /* set by eigen(): whether the last iterate() re-decomposed */
    int eigen_done = 0;

    double& b(int i, int j) { return B[(size_t) i * n + j]; }
    double& c(int i, int j) { return C[(size_t) i * n + j]; }
    double& isc(int i, int j) { return invsqrtC[(size_t) i * n + j]; }
    double* x(int k) { return &arx[(size_t) k * n]; }

    /* set by a caller that optimizes something else than a built-in objective through this
     * object (CCPSO's local search on the swarm weights, ccpso.cpp:400-412) */
    std::function<double(const double*)> fhook;
    double evaluate(const double *p)
    {
        return fhook ? fhook(p) : bbo_objective_eval(obj, n, p, aux.data());
    }

    /* base_cmaes.cpp:54-134 */
    void init_base(const double *guess)
    {
        if (adaptpop) lambda = 4 + (int) (3. * std::log(n));
        mu = lambda / 2;
        if (adaptit) {
            mit = (int) (100 + 50 * (n + 3) * (n + 3) / std::sqrt(1. * lambda));
            mfev = mit * lambda;
        } else {
            mit = mfev / lambda;
        }
        arx.assign((size_t) lambda * n, 0.);
        for (int i = 0; i < 4; i++) { ibw[i] = 0; ybw[i] = 0.; }
        fit.assign(lambda, Ranked { 0, 0. });

        weights.assign(mu, 0.);
        double sum = 0.;
        for (int i = 0; i < mu; i++) {
            weights[i] = std::log(0.5 * (lambda + 1.)) - std::log(i + 1.);
            sum += weights[i];
        }
        scal(mu, 1. / sum, weights.data());
        const double lenw = dot(mu, weights.data(), weights.data());
        mueff = 1. / lenw;

        chi = std::sqrt(n) * (1. - 1. / (4. * n) + 1. / (21. * n * n));
        sigma = sigma0;
        cc = (4. + mueff / n) / (n + 4. + 2. * mueff / n);
        cs = (mueff + 2.) / (5. + n + mueff);
        c1 = 2. / ((1.3 + n) * (1.3 + n) + mueff);
        cmu = std::min(1. - c1,
                2. * (mueff - 2. + 1. / mueff) / ((2. + n) * (2. + n) + mueff));
        damps = 1. + cs + 2. * std::max(0., std::sqrt((mueff - 1.) / (n + 1.)) - 1.);

        pc.assign(n, 0.);
        ps.assign(n, 0.);
        tmp.assign(n, 0.);
        xold.assign(n, 0.);
        xmean.assign(guess, guess + n);
        it = fev = 0;

        hlen = 10 + (int) std::ceil((30. * n) / lambda);
        ik = (int) std::ceil(0.1 + lambda / 4.);
        best.reset(hlen);
        kth.reset(hlen);
        fbest = -std::numeric_limits<double>::infinity();
        fworst = std::numeric_limits<double>::infinity();
    }

    /* cmaes.cpp:44-63 then active_cmaes.cpp:42-69 */
    void init(int obj_, int n_, const double *lo, const double *up, const double *guess)
    {
        obj = obj_;
        n = n_;
        lower.assign(lo, lo + n);
        upper.assign(up, up + n);
        aux.assign(n, 0.);
        if (!fhook) bbo_objective_aux(obj, n, aux.data());
        init_base(guess);

        if (variant == 2) {
            /* sep_cmaes.cpp:41-68 */
            cc = 4. / (n + 4.);
            cs = (mueff + 2.) / (3. + n + mueff);
            damps = 1. + cs + 2. * std::max(0., std::sqrt((mueff - 1.) / (n + 1.)) - 1.);
            ccov = 2. / ((n + std::sqrt(2.)) * (n + std::sqrt(2.)) * mueff);
            ccov += std::min(1., (2. * mueff - 1.) / ((n + 2.) * (n + 2.) + mueff))
                    * (1. - 1. / mueff);
            if (adjustlr) ccov *= ((n + 2.) / 3.);
            D.assign(n, 1.);
            csep.assign(n, 1.);
            flag = 0;
            return;
        }

        eigenfreq = eigenrate * lambda / (c1 + cmu) / n;
        eigenlastev = 0;
        D.assign(n, 1.);
        /* the reference resize()s _b and _c (values survive a re-init of the same
         * object) but clear()s _invsqrtc; then sets the three diagonals to 1 */
        if ((int) B.size() != n * n) B.assign((size_t) n * n, 0.);
        if ((int) C.size() != n * n) C.assign((size_t) n * n, 0.);
        invsqrtC.assign((size_t) n * n, 0.);
        for (int d = 0; d < n; d++) c(d, d) = isc(d, d) = b(d, d) = 1.;
        flag = 0;

        if (variant == 1) {
            cm = 1.;
            alphaold = 0.5;
            cc = (4. + 0. * mueff / n) / (n + 4. + 0. * 2. * mueff / n);
            cs = (mueff + 2.) / (3. + n + mueff);
            c1 = alphacov * std::min(1., lambda / 6.) / ((n + 1.3) * (n + 1.3) + mueff);
            cmu = 1. - c1;
            cmu = std::min(cmu,
                    alphacov * (mueff - 2. + 1. / mueff)
                            / ((2. + n) * (2. + n) + alphacov * mueff / 2.));
            cneg = (1. - cmu) * (alphacov / 8.) * mueff
                    / (std::pow(n + 2., 1.5) + 2. * mueff);
            damps = 1. + cs
                    + 2. * std::max(0., std::sqrt((mueff - 1.) / (n + 1.)) - 1.);
            eigenfreq = eigenrate * (1. / (c1 + cmu + cneg)) / n;
            eigenlastev = 0;
            ycoeff.assign(mu, 0.);
        }
    }

    /* base_cmaes.cpp:136-148 */
    void set_params(int np, double sig, int mfev_)
    {
        lambda = np;
        sigma0 = sig;
        mfev = mfev_;
        adaptpop = adaptit = false;
        if (bound) bound = false;
    }

    double next_normal(int k, int j)
    {
        if (rng_mode == RNG_MT) return Z.draw();
        if (rng_mode == RNG_INJECT) return zinject[(size_t) k * n + j];
        /* RNG_PHILOX: column j of candidate k in generation `it` (layout: philox.h) */
        double z[4];
        bbo_normal_quad(seed, (uint32_t) k, (uint32_t) bbo_cma_quad_of_column(j), (uint32_t) it,
                bbo_stream(BBO_STREAM_CMA_NORMAL, 0), z);
        return z[bbo_cma_slot_of_column(j)];
    }

    /* cmaes.cpp:65-80 */
    void sample()
    {
        zlast.assign((size_t) lambda * n, 0.);
        if (variant == 2) {
            /* sep_cmaes.cpp:70-80 */
            for (int k = 0; k < lambda; k++)
                for (int i = 0; i < n; i++) {
                    const double z = next_normal(k, i);
                    zlast[(size_t) k * n + i] = z;
                    double v = xmean[i] + sigma * D[i] * z;
                    if (bound) v = std::max(lower[i], std::min(v, upper[i]));
                    x(k)[i] = v;
                }
            return;
        }
        for (int k = 0; k < lambda; k++) {
            for (int i = 0; i < n; i++) {
                const double z = next_normal(k, i);
                zlast[(size_t) k * n + i] = z;
                tmp[i] = D[i] * z;
            }
            for (int i = 0; i < n; i++) {
                const double s = dot(n, &B[(size_t) i * n], tmp.data());
                double v = xmean[i] + sigma * s;
                if (bound) v = std::max(lower[i], std::min(v, upper[i]));
                x(k)[i] = v;
            }
        }
    }

    /* base_cmaes.cpp:211-230 */
    void evaluate_sort()
    {
        for (int i = 0; i < lambda; i++) {
            fit[i].index = i;
            fit[i].value = evaluate(x(i));
        }
        fev += lambda;
        std::sort(fit.begin(), fit.end(),
                [](const Ranked &a, const Ranked &b) { return a.value < b.value; });
        ibw[0] = fit[0].index;
        ibw[1] = fit[1].index;
        ibw[2] = fit[lambda - 2].index;
        ibw[3] = fit[lambda - 1].index;
        ybw[0] = fit[0].value;
        ybw[1] = fit[1].value;
        ybw[2] = fit[lambda - 2].value;
        ybw[3] = fit[lambda - 1].value;
    }

    /* base_cmaes.cpp:176-189 */
    void update_sigma()
    {
        const double pslen = nrm2(n, ps.data());
        sigma *= std::exp(std::min(1., (cs / damps) * (pslen / chi - 1.)));
        if (fit[0].value == fit[ik].value) sigma *= std::exp(0.2 + cs / damps);
        if (it >= hlen && fworst - fbest == 0.) sigma *= std::exp(0.2 + cs / damps);
    }

    /* cmaes.cpp:82-149 (variant 0) and active_cmaes.cpp:71-168 (variant 1) */
    /* sep_cmaes.cpp:82-135 */
    void update_distribution_sep()
    {
        std::copy(xmean.begin(), xmean.end(), xold.begin());
        for (int i = 0; i < n; i++) {
            double sum = 0.;
            for (int k = 0; k < mu; k++) sum += weights[k] * x(fit[k].index)[i];
            xmean[i] = sum;
            if (bound) xmean[i] = std::max(lower[i], std::min(xmean[i], upper[i]));
        }
        const double csc = std::sqrt(cs * (2. - cs) * mueff);
        for (int i = 0; i < n; i++) {
            ps[i] *= (1. - cs);
            /* (the reference scales by _c[i], not by 1/_diagd[i]: kept) */
            ps[i] += csc * csep[i] * (xmean[i] - xold[i]) / sigma;
        }
        const double pslen = nrm2(n, ps.data());
        const double denom = 1. - std::pow(1. - cs, 2. * fev / lambda);
        const int hsig = pslen / std::sqrt(denom) / chi < 1.4 + 2. / (n + 1.) ? 1 : 0;
        const double ccc = std::sqrt(cc * (2. - cc) * mueff);
        for (int i = 0; i < n; i++)
            pc[i] = (1. - cc) * pc[i] + hsig * ccc * (xmean[i] - xold[i]) / sigma;
        for (int i = 0; i < n; i++) {
            double sum = (1. - ccov) * csep[i] + (ccov / mueff) * pc[i] * pc[i];
            for (int k = 0; k < mu; k++) {
                const double di = (x(fit[k].index)[i] - xold[i]) / sigma;
                sum += ccov * (1. - 1. / mueff) * weights[k] * di * di;
            }
            csep[i] = sum;
            D[i] = std::sqrt(csep[i]);
        }
        update_sigma();
    }

    /* sep_cmaes.cpp:137-206; returns the stop flag (0 = continue) */
    int converged_sep()
    {
        if (it >= mit) return flag = 1;
        if (it >= hlen && fworst - fbest < tol) return flag = 2;
        if (best.len >= n && kth.len >= n) {
            int eq = 0;
            for (int i = 0; i < n; i++) {
                if (best.back(i) == kth.back(i)) {
                    eq++;
                    if (3 * eq >= n) return flag = 3;
                }
            }
        }
        bool all = true;
        for (int i = 0; i < n; i++) {
            if (std::max(pc[i], D[i]) * sigma / sigma0 >= tol) {
                all = false;
                break;
            }
        }
        if (all) return flag = 4;
        if (sigma / sigma0 > 1.0e20 * D[n - 1]) return flag = 5;   /* unsorted _diagd: kept */
        if (D[n - 1] > 1.0e7 * D[0]) return flag = 7;
        const int iaxis = n - 1 - ((it - 1) % n);
        if (xmean[iaxis] == xmean[iaxis] + 0.1 * sigma * D[iaxis]) return flag = 8;
        for (int i = 0; i < n; i++) {
            if (xmean[i] == xmean[i] + 0.2 * sigma * D[i]) return flag = 9;
        }
        return 0;
    }

    void update_distribution()
    {
        if (variant == 2) {
            update_distribution_sep();
            return;
        }
        const bool active = (variant == 1);
        std::copy(xmean.begin(), xmean.end(), xold.begin());
        for (int i = 0; i < n; i++) {
            double sum = 0.;
            for (int k = 0; k < mu; k++) sum += weights[k] * x(fit[k].index)[i];
            xmean[i] = active ? xold[i] * (1. - cm) + sum * cm : sum;
            if (bound) xmean[i] = std::max(lower[i], std::min(xmean[i], upper[i]));
        }

        const double csc = std::sqrt(cs * (2. - cs) * mueff);
        for (int i = 0; i < n; i++) {
            ps[i] *= (1. - cs);
            for (int j = 0; j < n; j++) {
                if (active)
                    ps[i] += csc * isc(i, j) * (xmean[j] - xold[j]) / (cm * sigma);
                else
                    ps[i] += csc * isc(i, j) * (xmean[j] - xold[j]) / sigma;
            }
        }

        const double pslen = nrm2(n, ps.data());
        const double denom = 1. - std::pow(1. - cs, 2. * fev / lambda);
        const int hsig = (pslen / std::sqrt(denom) / chi < 1.4 + 2. / (n + 1.)) ? 1 : 0;

        const double ccc = std::sqrt(cc * (2. - cc) * mueff);
        for (int i = 0; i < n; i++) {
            if (active)
                pc[i] = (1. - cc) * pc[i] + hsig * ccc * (xmean[i] - xold[i]) / (cm * sigma);
            else
                pc[i] = (1. - cc) * pc[i] + hsig * ccc * (xmean[i] - xold[i]) / sigma;
        }

        if (active) {
            for (int i = 0; i < mu; i++) {
                const double *xt = x(fit[lambda - mu + 1 + i - 1].index);
                const double *xb = x(fit[lambda - i - 1].index);
                double ssqtop = 0., ssqbot = 0.;
                for (int j = 0; j < n; j++) {
                    double tt = 0., tb = 0.;
                    for (int l = 0; l < n; l++) {
                        tt += isc(j, l) * (xt[l] - xold[l]);
                        tb += isc(j, l) * (xb[l] - xold[l]);
                    }
                    ssqtop += tt * tt;
                    ssqbot += tb * tb;
                }
                ssqbot = std::max(ssqbot, 1e-8);
                ycoeff[i] = ssqtop / ssqbot;
            }
        }

        const double c2 = (1. - hsig) * cc * (2. - cc);
        const double cmu1 = active ? cmu + cneg * (1. - alphaold) : cmu;
        for (int i = 0; i < n; i++) {
            for (int j = 0; j <= i; j++) {
                double sum;
                if (active)
                    sum = (1. - c1 - cmu + cneg * alphaold) * c(i, j)
                            + c1 * (pc[i] * pc[j] + c2 * c(i, j));
                else
                    sum = (1. - c1 - cmu) * c(i, j) + c1 * (pc[i] * pc[j] + c2 * c(i, j));
                for (int k = 0; k < mu; k++) {
                    const double *xm = x(fit[k].index);
                    const double di = (xm[i] - xold[i]) / sigma;
                    const double dj = (xm[j] - xold[j]) / sigma;
                    sum += cmu1 * weights[k] * di * dj;
                }
                if (active) {
                    for (int k = 0; k < mu; k++) {
                        const double *xm = x(fit[lambda - k - 1].index);
                        const double di = (xm[i] - xold[i]) / sigma;
                        const double dj = (xm[j] - xold[j]) / sigma;
                        sum -= cneg * weights[k] * ycoeff[k] * di * dj;
                    }
                }
                c(i, j) = sum;
            }
        }

        update_sigma();
        eigen(false);
    }

    /* base_cmaes.cpp:191-209 */
    void update_history()
    {
        if (it >= mit) return;
        best.add(fit[0].value);
        kth.add(fit[ik].value);
        if (best.len == best.cap) {
            fbest = std::numeric_limits<double>::infinity();
            fworst = -std::numeric_limits<double>::infinity();
            for (double fx : best.v) {
                fbest = std::min(fx, fbest);
                fworst = std::max(fx, fworst);
            }
        }
    }

    /* base_cmaes.cpp:150-156 */
    void iterate()
    {
        sample();
        evaluate_sort();
        update_distribution();
        update_history();
        it++;
    }

    /* cmaes.cpp:151-227; returns the stop flag (0 = continue) */
    int converged()
    {
        if (variant == 2) return converged_sep();
        if (it >= mit) return flag = 1;
        if (it >= hlen && fworst - fbest < tol) return flag = 2;
        if (best.len >= n && kth.len >= n) {
            int eq = 0;
            for (int i = 0; i < n; i++) {
                if (best.back(i) == kth.back(i)) {
                    eq++;
                    if (3 * eq >= n) return flag = 3;
                }
            }
        }
        bool all = true;
        for (int i = 0; i < n; i++) {
            if (std::max(pc[i], std::sqrt(c(i, i))) * sigma / sigma0 >= tol) {
                all = false;
                break;
            }
        }
        if (all) return flag = 4;
        if (sigma / sigma0 > 1.0e20 * D[n - 1]) return flag = 5;
        if (D[n - 1] > 1.0e7 * D[0]) return flag = 7;
        const int iaxis = n - 1 - ((it - 1) % n);
        all = true;
        for (int i = 0; i < n; i++) {
            if (xmean[i] != xmean[i] + 0.1 * sigma * D[iaxis] * b(iaxis, i)) {
                all = false;
                break;
            }
        }
        if (all) return flag = 8;
        for (int i = 0; i < n; i++) {
            if (xmean[i] == xmean[i] + 0.2 * sigma * std::sqrt(c(i, i))) return flag = 9;
        }
        return 0;
    }

    /* base_cmaes.cpp:232-238 */
    const double* best_solution()
    {
        return it <= 0 ? xmean.data() : x(ibw[0]);
    }

    /* base_cmaes.cpp:162-174 */
    bool optimize(int obj_, int n_, const double *lo, const double *up, const double *guess)
    {
        init(obj_, n_, lo, up, guess);
        while (fev < mfev) {
            iterate();
            if (converged()) return true;
        }
        return false;
    }

    /* cmaes.cpp:229-283 */
    void eigen(bool force)
    {
        eigen_done = 0;
        if (!force && fev - eigenlastev <= eigenfreq) return;
        eigen_done = 1;
        for (int i = 0; i < n; i++)
            for (int j = 0; j <= i; j++) b(i, j) = b(j, i) = c(i, j);
        eigenlastev = fev;
        householder_tridiag();
        ql_implicit();

        if (D[0] <= 0.) {
            for (int i = 0; i < n; i++) D[i] = std::max(D[i], 0.);
            const double shift = D[n - 1] / 1e14;
            for (int i = 0; i < n; i++) {
                c(i, i) += shift;
                D[i] += shift;
            }
        }
        if (D[n - 1] > 1e14 * D[0]) {
            const double shift = D[n - 1] / 1e14 - D[0];
            for (int i = 0; i < n; i++) {
                c(i, i) += shift;
                D[i] += shift;
            }
        }
        for (int i = 0; i < n; i++) D[i] = std::sqrt(D[i]);
        for (int i = 0; i < n; i++) {
            for (int j = 0; j <= i; j++) {
                double sum = 0.;
                for (int k = 0; k < n; k++) sum += b(i, k) / D[k] * b(j, k);
                isc(i, j) = isc(j, i) = sum;
            }
        }
    }

    /* Householder reduction of the symmetric matrix held in B to tridiagonal
     * form (EISPACK tred2 as restated in cmaes.cpp:285-381): on exit D holds
     * the diagonal, tmp[1..n-1] the sub-diagonal, B the accumulated
     * orthogonal transform.  d = D, e = tmp. */
    void householder_tridiag()
    {
        double *d = D.data(), *e = tmp.data();
        for (int j = 0; j < n; j++) d[j] = b(n - 1, j);

        for (int i = n - 1; i > 0; i--) {
            double scale = 0., h = 0.;
            for (int k = 0; k < i; k++) scale += std::fabs(d[k]);
            if (scale == 0.) {
                e[i] = d[i - 1];
                for (int j = 0; j < i; j++) {
                    d[j] = b(i - 1, j);
                    b(i, j) = b(j, i) = 0.;
                }
            } else {
                for (int k = 0; k < i; k++) {
                    d[k] /= scale;
                    h += d[k] * d[k];
                }
                double f = d[i - 1];
                double g = std::sqrt(h);
                if (f > 0) g = -g;
                e[i] = scale * g;
                h = h - f * g;
                d[i - 1] = f - g;
                for (int j = 0; j < i; j++) e[j] = 0.;

                for (int j = 0; j < i; j++) {
                    f = d[j];
                    b(j, i) = f;
                    g = e[j] + b(j, j) * f;
                    for (int k = j + 1; k <= i - 1; k++) {
                        g += b(k, j) * d[k];
                        e[k] += b(k, j) * f;
                    }
                    e[j] = g;
                }
                scal(i, 1. / h, e);
                f = dot(i, e, d);
                const double hh = f / (h + h);
                axpy(i, -hh, d, e);
                for (int j = 0; j < i; j++) {
                    f = d[j];
                    g = e[j];
                    for (int k = j; k <= i - 1; k++) b(k, j) -= (f * e[k] + g * d[k]);
                    d[j] = b(i - 1, j);
                    b(i, j) = 0.;
                }
            }
            d[i] = h;
        }

        for (int i = 0; i < n - 1; i++) {
            b(n - 1, i) = b(i, i);
            b(i, i) = 1.;
            const double h = d[i + 1];
            if (h != 0.) {
                for (int k = 0; k <= i; k++) d[k] = b(k, i + 1) / h;
                for (int j = 0; j <= i; j++) {
                    double g = 0.;
                    for (int k = 0; k <= i; k++) g += b(k, i + 1) * b(k, j);
                    for (int k = 0; k <= i; k++) b(k, j) -= g * d[k];
                }
            }
            for (int k = 0; k <= i; k++) b(k, i + 1) = 0.;
        }
        for (int j = 0; j < n; j++) {
            d[j] = b(n - 1, j);
            b(n - 1, j) = 0.;
        }
        b(n - 1, n - 1) = 1.;
        e[0] = 0.;
    }

    /* implicit-shift QL on the tridiagonal (EISPACK tql2 as restated in
     * cmaes.cpp:383-478), eigenvectors accumulated into B, then ascending
     * selection sort of the pairs. */
    void ql_implicit()
    {
        double *d = D.data(), *e = tmp.data();
        for (int i = 1; i < n; i++) e[i - 1] = e[i];
        e[n - 1] = 0.;
        double f = 0., tst1 = 0.;
        const double eps = std::pow(2., -52.);
        for (int l = 0; l < n; l++) {
            tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
            int m = l;
            for (m = l; m < n; m++)
                if (std::fabs(e[m]) <= eps * tst1) break;
            if (m >= n) break;

            if (m > l) {
                do {
                    double g = d[l];
                    double p = (d[l + 1] - g) / (2. * e[l]);
                    double r = ref_hypot(p, 1.);
                    r = sign_of(r, p);
                    d[l] = e[l] / (p + r);
                    d[l + 1] = e[l] * (p + r);
                    const double dl1 = d[l + 1];
                    double h = g - d[l];
                    for (int i = l + 2; i < n; i++) d[i] -= h;
                    f += h;

                    p = d[m];
                    double cth = 1., c2 = cth, c3 = cth;
                    const double el1 = e[l + 1];
                    double s = 0., s2 = 0.;
                    for (int i = m - 1; i >= l; i--) {
                        c3 = c2;
                        c2 = cth;
                        s2 = s;
                        g = cth * e[i];
                        h = cth * p;
                        r = ref_hypot(p, e[i]);
                        e[i + 1] = s * r;
                        s = e[i] / r;
                        cth = p / r;
                        p = cth * d[i] - s * g;
                        d[i + 1] = h + s * (cth * g + s * d[i]);
                        for (int k = 0; k < n; k++) {
                            h = b(k, i + 1);
                            b(k, i + 1) = s * b(k, i) + cth * h;
                            b(k, i) = cth * b(k, i) - s * h;
                        }
                    }
                    p = -s * s2 * c3 * el1 * e[l] / dl1;
                    e[l] = s * p;
                    d[l] = cth * p;
                } while (std::fabs(e[l]) > eps * tst1);
            }
            d[l] += f;
            e[l] = 0.;
        }

        for (int i = 0; i < n - 1; i++) {
            int k = i;
            double p = d[i];
            for (int j = i + 1; j < n; j++) {
                if (d[j] < p) {
                    k = j;
                    p = d[j];
                }
            }
            if (k != i) {
                d[k] = d[i];
                d[i] = p;
                for (int j = 0; j < n; j++) std::swap(b(j, i), b(j, k));
            }
        }
    }

    /* ---- named state access for the tests -------------------------------- */
    static int put(const std::vector<double> &v, double *out, int cap)
    {
        for (size_t i = 0; i < v.size() && (int) i < cap; i++) out[i] = v[i];
        return (int) v.size();
    }
    static int put1(double v, double *out, int cap)
    {
        if (cap > 0) out[0] = v;
        return 1;
    }

    int get(const std::string &k, double *out, int cap)
    {
        if (k == "xmean") return put(xmean, out, cap);
        if (k == "xold") return put(xold, out, cap);
        if (k == "pc") return put(pc, out, cap);
        if (k == "ps") return put(ps, out, cap);
        if (k == "weights") return put(weights, out, cap);
        if (k == "D") return put(D, out, cap);
        if (k == "B") return put(B, out, cap);
        if (k == "C") return put(C, out, cap);
        if (k == "invsqrtC") return put(invsqrtC, out, cap);
        if (k == "arx") return put(arx, out, cap);
        if (k == "zlast") return put(zlast, out, cap);
        if (k == "ycoeff") return put(ycoeff, out, cap);
        if (k == "fit_val" || k == "fit_idx") {
            for (int i = 0; i < (int) fit.size() && i < cap; i++)
                out[i] = (k == "fit_val") ? fit[i].value : (double) fit[i].index;
            return (int) fit.size();
        }
        if (k == "best_hist") return put(best.v, out, cap);
        if (k == "kth_hist") return put(kth.v, out, cap);
        if (k == "sigma") return put1(sigma, out, cap);
        if (k == "sigma0") return put1(sigma0, out, cap);
        if (k == "n") return put1(n, out, cap);
        if (k == "lambda") return put1(lambda, out, cap);
        if (k == "mu") return put1(mu, out, cap);
        if (k == "mueff") return put1(mueff, out, cap);
        if (k == "cc") return put1(cc, out, cap);
        if (k == "cs") return put1(cs, out, cap);
        if (k == "c1") return put1(c1, out, cap);
        if (k == "cmu") return put1(cmu, out, cap);
        if (k == "damps") return put1(damps, out, cap);
        if (k == "chi") return put1(chi, out, cap);
        if (k == "eigenfreq") return put1(eigenfreq, out, cap);
        if (k == "eigenlastev") return put1(eigenlastev, out, cap);
        if (k == "eigen_done") return put1(eigen_done, out, cap);
        if (k == "hlen") return put1(hlen, out, cap);
        if (k == "ik") return put1(ik, out, cap);
        if (k == "mit") return put1(mit, out, cap);
        if (k == "mfev") return put1(mfev, out, cap);
        if (k == "it") return put1(it, out, cap);
        if (k == "fev") return put1(fev, out, cap);
        if (k == "flag") return put1(flag, out, cap);
        if (k == "fbest") return put1(fbest, out, cap);
        if (k == "fworst") return put1(fworst, out, cap);
        if (k == "best_len") return put1(best.len, out, cap);
        if (k == "best_buffer") return put1(best.head, out, cap);
        if (k == "cneg") return put1(cneg, out, cap);
        if (k == "alphaold") return put1(alphaold, out, cap);
        if (k == "cm") return put1(cm, out, cap);
        if (k == "ccov") return put1(ccov, out, cap);
        if (k == "csep") return put(csep, out, cap);
        return -1;
    }

    int set(const std::string &k, const double *in, int count)
    {
        auto take = [&](std::vector<double> &v) {
            if ((int) v.size() != count) return -2;
            std::copy(in, in + count, v.begin());
            return count;
        };
        if (k == "xmean") return take(xmean);
        if (k == "xold") return take(xold);
        if (k == "pc") return take(pc);
        if (k == "ps") return take(ps);
        if (k == "D") return take(D);
        if (k == "B") return take(B);
        if (k == "C") return take(C);
        if (k == "invsqrtC") return take(invsqrtC);
        if (k == "csep") return take(csep);
        if (k == "arx") return take(arx);
        if (k == "best_hist") return take(best.v);
        if (k == "kth_hist") return take(kth.v);
        if (count != 1) return -2;
        if (k == "sigma") { sigma = in[0]; return 1; }
        if (k == "it") { it = (int) in[0]; return 1; }
        if (k == "fev") { fev = (int) in[0]; return 1; }
        if (k == "eigenlastev") { eigenlastev = (int) in[0]; return 1; }
        if (k == "fbest") { fbest = in[0]; return 1; }
        if (k == "fworst") { fworst = in[0]; return 1; }
        if (k == "best_len") { best.len = kth.len = (int) in[0]; return 1; }
        if (k == "best_buffer") { best.head = kth.head = (int) in[0]; return 1; }
        return -1;
    }
};

} // namespace

/* the DE / PSO / restart-driver restatements live in their own include so the
 * file stays navigable; they share the helpers above */
#include "bbo_oracle_pop.inc"

extern "C" {

/* ---- generators --------------------------------------------------------- */
void orc_seed(uint32_t s)
{
    g_mt.seed(s);
    g_mt_seeded = true;
}
double orc_draw_uniform(double a, double b) { return mt_uniform(a, b); }
int orc_draw_int(int a, int b) { return mt_int(a, b); }
double orc_draw_normal(void) { return g_test_normal.draw(); }
void orc_reset_test_normal(void) { g_test_normal = NormalCache(); }
uint32_t orc_draw_raw(void) { return engine().next(); }
void orc_set_hypot_mode(int m) { g_hypot_mode = m; }

void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
        uint32_t *out)
{
    bbo_philox(seed, c0, c1, c2, c3, out);
}
double orc_log_unit(double u) { return bbo_log_unit(u); }
void orc_sincos_turn(double t, double *s, double *c) { bbo_sincos_turn(t, s, c); }
double orc_exp_neg(double s) { return bbo_exp_neg(s); }
void orc_normal_quad(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, double *z)
{
    bbo_normal_quad(seed, c0, c1, c2, c3, z);
}
/* strip i of the ziggurat: x_i = W 2^22, K, f(x_i); n = BBO_ZIG_N on i < 0 */
int orc_zig_strip(int i, double *w, uint32_t *k, double *f)
{
    if (i < 0 || i > BBO_ZIG_N) return BBO_ZIG_N;
    *w = i < BBO_ZIG_N ? bbo_zig_w[i] : 0.;
    *k = i < BBO_ZIG_N ? bbo_zig_k[i] : 0u;
    *f = bbo_zig_f[i];
    return BBO_ZIG_N;
}
void orc_philox_normals(uint64_t seed, int gen, int rows, int n, double *out)
{
    for (int k = 0; k < rows; k++)
        for (int j = 0; j < n; j++) {
            double z[4];
            bbo_normal_quad(seed, (uint32_t) k, (uint32_t) bbo_cma_quad_of_column(j),
                    (uint32_t) gen, bbo_stream(BBO_STREAM_CMA_NORMAL, 0), z);
            out[(size_t) k * n + j] = z[bbo_cma_slot_of_column(j)];
        }
}

double orc_objective(int obj, int n, const double *x)
{
    std::vector<double> aux(n > 0 ? n : 1);
    bbo_objective_aux(obj, n, aux.data());
    return bbo_objective_eval(obj, n, x, aux.data());
}

/* ---- CMA-ES -------------------------------------------------------------- */
void* orc_cma_create(int variant, int mfev, double tol, int np, double sigma0,
        int bound, double alphacov, double eigenrate)
{
    Cma *h = new Cma();
    h->variant = variant;
    h->mfev = mfev;
    h->tol = tol;
    h->lambda = np;
    h->sigma0 = sigma0;
    h->bound = bound != 0;
    h->alphacov = alphacov;
    h->eigenrate = eigenrate;
    if (variant == 2) h->adjustlr = alphacov != 0.;   /* SepCmaes: the slot carries adjustlr */
    return h;
}
void orc_cma_destroy(void *p) { delete static_cast<Cma*>(p); }
void orc_cma_set_rng(void *p, int mode, uint64_t seed)
{
    Cma *h = static_cast<Cma*>(p);
    h->rng_mode = mode;
    h->seed = seed;
}
void orc_cma_inject_z(void *p, const double *z, int count)
{
    static_cast<Cma*>(p)->zinject.assign(z, z + count);
}
void orc_cma_init(void *p, int obj, int n, const double *lower, const double *upper,
        const double *guess)
{
    static_cast<Cma*>(p)->init(obj, n, lower, upper, guess);
}
void orc_cma_set_params(void *p, int np, double sigma, int mfev)
{
    static_cast<Cma*>(p)->set_params(np, sigma, mfev);
}
void orc_cma_iterate(void *p) { static_cast<Cma*>(p)->iterate(); }
void orc_cma_sample(void *p) { static_cast<Cma*>(p)->sample(); }
void orc_cma_evaluate_sort(void *p) { static_cast<Cma*>(p)->evaluate_sort(); }
void orc_cma_update_distribution(void *p) { static_cast<Cma*>(p)->update_distribution(); }
void orc_cma_update_history(void *p)
{
    Cma *h = static_cast<Cma*>(p);
    h->update_history();
    h->it++;
}
void orc_cma_eigen(void *p, int force) { static_cast<Cma*>(p)->eigen(force != 0); }
int orc_cma_converged(void *p) { return static_cast<Cma*>(p)->converged(); }
int orc_cma_get(void *p, const char *key, double *out, int cap)
{
    return static_cast<Cma*>(p)->get(key, out, cap);
}
int orc_cma_set(void *p, const char *key, const double *in, int count)
{
    return static_cast<Cma*>(p)->set(key, in, count);
}
int orc_cma_optimize(void *p, int obj, int n, const double *lower, const double *upper,
        const double *guess, double *x_out, int *fev, int *converged)
{
    Cma *h = static_cast<Cma*>(p);
    const bool c = h->optimize(obj, n, lower, upper, guess);
    const double *xb = h->best_solution();
    for (int i = 0; i < n; i++) x_out[i] = xb[i];
    *fev = h->fev;
    *converged = c ? 1 : 0;
    return 0;
}
void orc_cma_solution(void *p, double *x_out, int *fev, int *converged)
{
    Cma *h = static_cast<Cma*>(p);
    const double *xb = h->best_solution();
    for (int i = 0; i < h->n; i++) x_out[i] = xb[i];
    *fev = h->fev;
    *converged = h->converged() ? 1 : 0;
}

} // extern "C"
