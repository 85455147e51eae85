/*
 * oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
 *
 * A C-ABI driver around the REAL reference (mike-gimelfarb/bboptpy), compiled
 * from the sources where they lie under /root/reference by oracle/Makefile
 * (target `ref`), output oracle/_ref/libbbo_ref.so.  Nothing from the
 * reference is copied: this file only #includes its headers and subclasses its
 * optimizers to read their protected state (SURVEY.md Appendix B "State
 * probe").  It exists only in the development container; the GPU box sees the
 * prebuilt .so (if any) and the committed fixtures under tests/golden/.
 *
 * Used by: tests/ (to pin oracle/bbo_oracle.cpp bit-for-bit), by
 * oracle/gen_golden.py (to write tests/golden/...), and optionally by
 * bench.py's cpu_baseline leg (kind "reference").
 *
 * Reference interface driven here:
 *   MultivariateOptimizer   src/multivariate/multivariate.h:132-146
 *   Cmaes / ActiveCmaes     src/multivariate/cma/cmaes.h:40, active_cmaes.h:40
 *   BiPopCmaes / IPopCmaes  src/multivariate/cma/bipop_cmaes.h:49, ipop_cmaes.h:55
 *   ShadeSearch / JadeSearch src/multivariate/de/shade.h:42, jade.h:49
 *   APSOSearch              src/multivariate/pso/apso.h:38
 *   random_static::seed     src/random.hpp:238-241
 */
#include <cstdint>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "random.hpp"
#include "multivariate/cma/active_cmaes.h"
#include "multivariate/cma/bipop_cmaes.h"
#include "multivariate/cma/ipop_cmaes.h"
#include "multivariate/cma/sep_cmaes.h"
#include "multivariate/de/shade.h"
#include "multivariate/de/jade.h"
#include "multivariate/de/sansde.h"
#include "multivariate/pso/apso.h"
#include "multivariate/pso/cso.h"
#include "multivariate/pso/ccpso.h"

#include "objectives.h"

using Random = effolkronium::random_static;

namespace {

struct ObjCtx {
    int obj = 0, n = 0;
    long calls = 0;
    std::vector<double> aux;
};

multivariate make_objective(const std::shared_ptr<ObjCtx> &ctx)
{
    return [ctx](const double *x) -> double {
        ctx->calls++;
        return bbo_objective_eval(ctx->obj, ctx->n, x, ctx->aux.data());
    };
}

std::shared_ptr<ObjCtx> make_ctx(int obj, int n)
{
    auto ctx = std::make_shared<ObjCtx>();
    ctx->obj = obj;
    ctx->n = n;
    ctx->aux.resize(n > 0 ? n : 1);
    bbo_objective_aux(obj, n, ctx->aux.data());
    return ctx;
}

int put(const std::vector<double> &v, double *out, int cap)
{
    const int m = (int) v.size();
    for (int i = 0; i < m && i < cap; i++) out[i] = v[i];
    return m;
}

int put_mat(const std::vector<std::vector<double>> &a, double *out, int cap)
{
    int k = 0;
    for (const auto &row : a)
        for (double v : row) {
            if (k < cap) out[k] = v;
            k++;
        }
    return k;
}

int put1(double v, double *out, int cap)
{
    if (cap > 0) out[0] = v;
    return 1;
}

/* ---- CMA probes -------------------------------------------------------- */
template<class Base, bool ACTIVE>
struct CmaProbe: Base {
    using Base::Base;

    int get(const std::string &k, double *out, int cap)
    {
        if (k == "xmean") return put(this->_xmean, out, cap);
        if (k == "xold") return put(this->_xold, out, cap);
        if (k == "pc") return put(this->_pc, out, cap);
        if (k == "ps") return put(this->_ps, out, cap);
        if (k == "weights") return put(this->_weights, out, cap);
        if (k == "D") return put(this->_diagd, out, cap);
        if (k == "B") return put_mat(this->_b, out, cap);
        if (k == "C") return put_mat(this->_c, out, cap);
        if (k == "invsqrtC") return put_mat(this->_invsqrtc, out, cap);
        if (k == "arx") return put_mat(this->_arx, out, cap);
        if (k == "fit_val" || k == "fit_idx") {
            int m = 0;
            for (const auto &f : this->_fitness) {
                if (m < cap) out[m] = (k == "fit_val") ? f._value : (double) f._index;
                m++;
            }
            return m;
        }
        if (k == "best_hist") return put(this->_best._hist, out, cap);
        if (k == "kth_hist") return put(this->_kth._hist, out, cap);
        if (k == "sigma") return put1(this->_sigma, out, cap);
        if (k == "sigma0") return put1(this->_sigma0, out, cap);
        if (k == "n") return put1(this->_n, out, cap);
        if (k == "lambda") return put1(this->_lambda, out, cap);
        if (k == "mu") return put1(this->_mu, out, cap);
        if (k == "mueff") return put1(this->_mueff, out, cap);
        if (k == "cc") return put1(this->_cc, out, cap);
        if (k == "cs") return put1(this->_cs, out, cap);
        if (k == "c1") return put1(this->_c1, out, cap);
        if (k == "cmu") return put1(this->_cmu, out, cap);
        if (k == "damps") return put1(this->_damps, out, cap);
        if (k == "chi") return put1(this->_chi, out, cap);
        if (k == "eigenfreq") return put1(this->_eigenfreq, out, cap);
        if (k == "eigenlastev") return put1(this->_eigenlastev, out, cap);
        if (k == "hlen") return put1(this->_hlen, out, cap);
        if (k == "ik") return put1(this->_ik, out, cap);
        if (k == "mit") return put1(this->_mit, out, cap);
        if (k == "mfev") return put1(this->_mfev, out, cap);
        if (k == "it") return put1(this->_it, out, cap);
        if (k == "fev") return put1(this->_fev, out, cap);
        if (k == "flag") return put1(this->_flag, out, cap);
        if (k == "fbest") return put1(this->_fbest, out, cap);
        if (k == "fworst") return put1(this->_fworst, out, cap);
        if (k == "best_len") return put1(this->_best._len, out, cap);
        if (k == "best_buffer") return put1(this->_best._buffer, out, cap);
        if constexpr (ACTIVE) {
            if (k == "cneg") return put1(this->_cneg, out, cap);
            if (k == "alphaold") return put1(this->_alphaold, out, cap);
            if (k == "cm") return put1(this->_cm, out, cap);
            if (k == "ycoeff") return put(this->_ycoeff, out, cap);
        }
        return -1;
    }
};

using PlainProbe = CmaProbe<Cmaes, false>;
using ActiveProbe = CmaProbe<ActiveCmaes, true>;

/* SepCmaes keeps a diagonal: _c and _diagd are vectors, there is no B / C^-1/2 */
struct SepProbe: SepCmaes {
    using SepCmaes::SepCmaes;

    int get(const std::string &k, double *out, int cap)
    {
        if (k == "xmean") return put(_xmean, out, cap);
        if (k == "xold") return put(_xold, out, cap);
        if (k == "pc") return put(_pc, out, cap);
        if (k == "ps") return put(_ps, out, cap);
        if (k == "weights") return put(_weights, out, cap);
        if (k == "D") return put(_diagd, out, cap);
        if (k == "csep") return put(_c, out, cap);
        if (k == "arx") return put_mat(_arx, out, cap);
        if (k == "fit_val" || k == "fit_idx") {
            int m = 0;
            for (const auto &f : _fitness) {
                if (m < cap) out[m] = (k == "fit_val") ? f._value : (double) f._index;
                m++;
            }
            return m;
        }
        if (k == "best_hist") return put(_best._hist, out, cap);
        if (k == "kth_hist") return put(_kth._hist, out, cap);
        if (k == "sigma") return put1(_sigma, out, cap);
        if (k == "sigma0") return put1(_sigma0, out, cap);
        if (k == "n") return put1(_n, out, cap);
        if (k == "lambda") return put1(_lambda, out, cap);
        if (k == "mu") return put1(_mu, out, cap);
        if (k == "mueff") return put1(_mueff, out, cap);
        if (k == "cc") return put1(_cc, out, cap);
        if (k == "cs") return put1(_cs, out, cap);
        if (k == "ccov") return put1(_ccov, out, cap);
        if (k == "damps") return put1(_damps, out, cap);
        if (k == "chi") return put1(_chi, out, cap);
        if (k == "hlen") return put1(_hlen, out, cap);
        if (k == "ik") return put1(_ik, out, cap);
        if (k == "mit") return put1(_mit, out, cap);
        if (k == "mfev") return put1(_mfev, out, cap);
        if (k == "it") return put1(_it, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "flag") return put1(_flag, out, cap);
        if (k == "fbest") return put1(_fbest, out, cap);
        if (k == "fworst") return put1(_fworst, out, cap);
        if (k == "best_len") return put1(_best._len, out, cap);
        if (k == "best_buffer") return put1(_best._buffer, out, cap);
        return -1;
    }
};

struct RefCma {
    int variant = 1;
    std::unique_ptr<BaseCmaes> alg;
    std::shared_ptr<ObjCtx> ctx;
    std::vector<double> lower, upper, guess;

    int get(const char *key, double *out, int cap)
    {
        if (variant == 0) return static_cast<PlainProbe*>(alg.get())->get(key, out, cap);
        if (variant == 2) return static_cast<SepProbe*>(alg.get())->get(key, out, cap);
        return static_cast<ActiveProbe*>(alg.get())->get(key, out, cap);
    }
    Cmaes* cma() { return static_cast<Cmaes*>(alg.get()); }
};

/* ---- DE / PSO probes --------------------------------------------------- */
struct ShadeProbe: ShadeSearch {
    using ShadeSearch::ShadeSearch;
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "x" || k == "f") {
            int m = 0;
            for (const auto &p : _swarm) {
                if (k == "f") {
                    if (m < cap) out[m] = p._f;
                    m++;
                } else
                    for (double v : p._x) {
                        if (m < cap) out[m] = v;
                        m++;
                    }
            }
            return m;
        }
        if (k == "arch") return put_mat(_arch, out, cap);
        if (k == "MCR") return put(_MCR, out, cap);
        if (k == "MF") return put(_MF, out, cap);
        if (k == "SCR") return put(_SCR, out, cap);
        if (k == "SF") return put(_SF, out, cap);
        if (k == "w") return put(_w, out, cap);
        if (k == "k") return put1(_k, out, cap);
        if (k == "np") return put1(_np, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "larch") return put1((double) _arch.size(), out, cap);
        return -1;
    }
};

struct JadeProbe: JadeSearch {
    using JadeSearch::JadeSearch;
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "x" || k == "f") {
            int m = 0;
            for (const auto &p : _swarm) {
                if (k == "f") {
                    if (m < cap) out[m] = p._f;
                    m++;
                } else
                    for (double v : p._x) {
                        if (m < cap) out[m] = v;
                        m++;
                    }
            }
            return m;
        }
        if (k == "arch") return put_mat(_arch, out, cap);
        if (k == "mucr") return put1(_mucr, out, cap);
        if (k == "muf") return put1(_muf, out, cap);
        if (k == "np") return put1(_np, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "larch") return put1((double) _arch.size(), out, cap);
        return -1;
    }
};

struct SansdeProbe: SaNSDESearch {
    using SaNSDESearch::SaNSDESearch;
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "x" || k == "f" || k == "cr") {
            int m = 0;
            for (const auto &q : _swarm) {
                if (k == "x") {
                    for (double v : q._x) {
                        if (m < cap) out[m] = v;
                        m++;
                    }
                } else {
                    if (m < cap) out[m] = k == "f" ? q._f : q._cr;
                    m++;
                }
            }
            return m;
        }
        if (k == "np") return put1(_np, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "it") return put1(_it, out, cap);
        if (k == "p") return put1(_p, out, cap);
        if (k == "fp") return put1(_fp, out, cap);
        if (k == "crm") return put1(_crm, out, cap);
        if (k == "crrec") return put1(_crrec, out, cap);
        if (k == "crdeltaf") return put1(_crdeltaf, out, cap);
        if (k == "pns" || k == "pnf") {
            const auto &a = k == "pns" ? _pns : _pnf;
            if (cap >= 2) { out[0] = a[0]; out[1] = a[1]; }
            return 2;
        }
        if (k == "fpns" || k == "fpnf") return put(k == "fpns" ? _fpns : _fpnf, out, cap);
        return -1;
    }
};

struct CsoProbe: CSOSearch {
    using CSOSearch::CSOSearch;
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "x" || k == "v" || k == "pmean" || k == "f" || k == "home") {
            int m = 0;
            for (const auto &q : _swarm) {
                if (k == "f") {
                    if (m < cap) out[m] = q._f;
                    m++;
                } else if (k == "home") {
                    /* birth slot, recovered from the stored neighbour pointer */
                    const int right = _ring ? (int) (q._right - &_swarm[0]) : 0;
                    if (m < cap) out[m] = _ring ? (double) ((right - 1 + _np) % _np) : -1.;
                    m++;
                } else {
                    const auto &src = k == "x" ? q._x : k == "v" ? q._v : q._mean;
                    for (double v : src) {
                        if (m < cap) out[m] = v;
                        m++;
                    }
                }
            }
            return m;
        }
        if (k == "mean") return put(_mean, out, cap);
        if (k == "meanw") return put(_meanw, out, cap);
        if (k == "phil") return put(_phil, out, cap);
        if (k == "phih") return put(_phih, out, cap);
        if (k == "xbest") return put(_best->_x, out, cap);
        if (k == "fbest") return put1(_best->_f, out, cap);
        if (k == "np") return put1(_np, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        return -1;
    }
};

struct CcpsoProbe: CCPSOSearch {
    using CCPSOSearch::CCPSOSearch;
    void set_local(MultivariateOptimizer *l, int freq)
    {
        _local = l;
        _localfreq = freq;
    }
    int get(const std::string &k, double *out, int cap)
    {
        auto flat = [&](const std::vector<std::vector<double>> &m) {
            int q = 0;
            for (const auto &row : m)
                for (double v : row) {
                    if (q < cap) out[q] = v;
                    q++;
                }
            return q;
        };
        auto flati = [&](const std::vector<std::vector<int>> &m) {
            int q = 0;
            for (const auto &row : m)
                for (int v : row) {
                    if (q < cap) out[q] = v;
                    q++;
                }
            return q;
        };
        if (k == "x") return flat(_X);
        if (k == "y") return flat(_Y);
        if (k == "yhat") return put(_yhat, out, cap);
        if (k == "fx") return flat(_fX);
        if (k == "fy") return flat(_fY);
        if (k == "k") return flati(_k);
        if (k == "ibest") return flati(_ibest);
        if (k == "strat") return flati(_strat);
        if (k == "fyhat") return put1(_fyhat, out, cap);
        if (k == "phat") return put1(_phat, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "it") return put1(_gen, out, cap);
        if (k == "is") return put1(_is, out, cap);
        if (k == "nswarm") return put1(_nswarm, out, cap);
        if (k == "cpswarm") return put1(_cpswarm, out, cap);
        if (k == "improved") return put1(_improved ? 1 : 0, out, cap);
        if (k == "np") return put1(_np, out, cap);
        return -1;
    }
};

struct ApsoProbe: APSOSearch {
    using APSOSearch::APSOSearch;
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "x" || k == "v" || k == "xb") {
            int m = 0;
            for (const auto &p : _swarm) {
                const auto &src = (k == "x") ? p._x : (k == "v") ? p._v : p._xb;
                for (double v : src) {
                    if (m < cap) out[m] = v;
                    m++;
                }
            }
            return m;
        }
        if (k == "f" || k == "fb") {
            int m = 0;
            for (const auto &p : _swarm) {
                if (m < cap) out[m] = (k == "f") ? p._f : p._fb;
                m++;
            }
            return m;
        }
        if (k == "xbest") return put(_xbest, out, cap);
        if (k == "ws") return put(_ws, out, cap);
        if (k == "fbest") return put1(_fbest, out, cap);
        if (k == "w") return put1(_w, out, cap);
        if (k == "c1") return put1(_c1, out, cap);
        if (k == "c2") return put1(_c2, out, cap);
        if (k == "state") return put1(_state, out, cap);
        if (k == "it") return put1(_it, out, cap);
        if (k == "maxit") return put1(_maxit, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "np") return put1(_np, out, cap);
        return -1;
    }
};

struct BiPopProbe: BiPopCmaes {
    using BiPopCmaes::BiPopCmaes;
    void set_print(bool on) { _print = on; }   /* the Tabular rows, bipop_cmaes.cpp:100-106,153-162 */
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "xbest") return put(_xbest, out, cap);
        if (k == "x0") return put(_x0, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "it") return put1(_it, out, cap);
        if (k == "lambdadef") return put1(_lambdadef, out, cap);
        if (k == "largelambda") return put1(_largelambda, out, cap);
        if (k == "smalllambda") return put1(_smalllambda, out, cap);
        if (k == "largebudget") return put1(_largebudget, out, cap);
        if (k == "smallbudget") return put1(_smallbudget, out, cap);
        if (k == "largerestarts") return put1(_largerestarts, out, cap);
        if (k == "smallrestarts") return put1(_smallrestarts, out, cap);
        if (k == "bestregime") return put1(_bestregime, out, cap);
        if (k == "fx") return put1(_fx, out, cap);
        if (k == "fxbest") return put1(_fxbest, out, cap);
        if (k == "largesigma") return put1(_largesigma, out, cap);
        if (k == "smallsigma") return put1(_smallsigma, out, cap);
        return -1;
    }
};

struct IPopProbe: IPopCmaes {
    using IPopCmaes::IPopCmaes;
    void set_print(bool on) { _print = on; }   /* ipop_cmaes.cpp:104-109,158-160 */
    int get(const std::string &k, double *out, int cap)
    {
        if (k == "xbest") return put(_xbest, out, cap);
        if (k == "x0") return put(_x0, out, cap);
        if (k == "fev") return put1(_fev, out, cap);
        if (k == "it") return put1(_it, out, cap);
        if (k == "lambda") return put1(_lambda, out, cap);
        if (k == "lambdadef") return put1(_lambdadef, out, cap);
        if (k == "sigma") return put1(_sigma, out, cap);
        if (k == "fx") return put1(_fx, out, cap);
        if (k == "fbest") return put1(_fbest, out, cap);
        return -1;
    }
};

template<class T>
struct RefPop {
    std::unique_ptr<T> alg;
    std::shared_ptr<ObjCtx> ctx;
    std::vector<double> lower, upper, guess;
    std::unique_ptr<RefCma> base;   /* restart drivers own their inner CMA */
};

template<class H>
void bind_problem(H *h, int obj, int n, const double *lower, const double *upper,
        const double *guess)
{
    h->ctx = make_ctx(obj, n);
    h->lower.assign(lower, lower + n);
    h->upper.assign(upper, upper + n);
    h->guess.assign(guess, guess + n);
}

std::normal_distribution<> g_testZ { 0., 1. };

} // namespace

extern "C" {

/* ---- global RNG -------------------------------------------------------- */
void ref_seed(uint32_t s)
{
    Random::seed(s);
}

/* raw distribution draws, to pin the oracle's own mt19937/distribution code
 *  */
double ref_draw_uniform(double a, double b) { return Random::get(a, b); }
int ref_draw_int(int a, int b) { return Random::get(a, b); }
double ref_draw_normal(void) { return Random::get(g_testZ); }
void ref_reset_test_normal(void) { g_testZ.reset(); }
uint32_t ref_draw_raw(void) { return Random::engine()(); }

/* ---- CMA-ES ------------------------------------------------------------ */
void* ref_cma_create(int variant, int mfev, double tol, int np, double sigma0,
        int bound, double alphacov, double eigenrate)
{
    auto *h = new RefCma();
    h->variant = variant;
    if (variant == 0)
        h->alg.reset(new PlainProbe(mfev, tol, np, sigma0, bound != 0, eigenrate));
    else if (variant == 2)   /* SepCmaes: the alphacov slot carries adjustlr */
        h->alg.reset(new SepProbe(mfev, tol, np, sigma0, bound != 0, alphacov != 0.));
    else
        h->alg.reset(new ActiveProbe(mfev, tol, np, sigma0, bound != 0, alphacov,
                eigenrate));
    return h;
}

void ref_cma_destroy(void *p) { delete static_cast<RefCma*>(p); }

void ref_cma_init(void *p, int obj, int n, const double *lower,
        const double *upper, const double *guess)
{
    auto *h = static_cast<RefCma*>(p);
    bind_problem(h, obj, n, lower, upper, guess);
    multivariate_problem prob { make_objective(h->ctx), n, h->lower.data(),
            h->upper.data() };
    h->alg->init(prob, h->guess.data());
}

void ref_cma_iterate(void *p) { static_cast<RefCma*>(p)->alg->iterate(); }

/* runs the reference's own stop tests; returns its _flag (0 = keep going) */
int ref_cma_converged(void *p)
{
    auto *h = static_cast<RefCma*>(p);
    const bool c = h->alg->converged();
    double f = 0.;
    h->get("flag", &f, 1);
    return c ? (int) f : 0;
}

int ref_cma_get(void *p, const char *key, double *out, int cap)
{
    return static_cast<RefCma*>(p)->get(key, out, cap);
}

/* the next `count` normals samplePopulation() WILL draw (engine and the
 * optimizer's own cached-spare distribution are copied, nothing is consumed) */
void ref_cma_peek_normals(void *p, double *out, int count)
{
    auto *h = static_cast<RefCma*>(p);
    auto eng = Random::get_engine();
    auto dist = h->variant == 2 ? static_cast<SepCmaes*>(h->alg.get())->_Z : h->cma()->_Z;
    for (int i = 0; i < count; i++) out[i] = dist(eng);
}

int ref_cma_optimize(void *p, int obj, int n, const double *lower,
        const double *upper, const double *guess, double *x_out, int *fev,
        int *converged)
{
    auto *h = static_cast<RefCma*>(p);
    bind_problem(h, obj, n, lower, upper, guess);
    multivariate_problem prob { make_objective(h->ctx), n, h->lower.data(),
            h->upper.data() };
    const auto sol = h->alg->optimize(prob, h->guess.data());
    for (int i = 0; i < n; i++) x_out[i] = sol._sol[i];
    *fev = sol._fev;
    *converged = sol._converged ? 1 : 0;
    return 0;
}

void ref_cma_solution(void *p, double *x_out, int *fev, int *converged)
{
    auto *h = static_cast<RefCma*>(p);
    const auto sol = h->alg->solution();
    for (size_t i = 0; i < sol._sol.size(); i++) x_out[i] = sol._sol[i];
    *fev = sol._fev;
    *converged = sol._converged ? 1 : 0;
}

/* ---- generic population optimizers (SHADE=0, JADE=1, APSO=2) ----------- */
#define POP_API(NAME, PROBE, CTOR_ARGS_DECL, CTOR_ARGS)                          \
void* ref_##NAME##_create CTOR_ARGS_DECL                                          \
{                                                                                 \
    auto *hp = new RefPop<PROBE>();                                                \
    hp->alg.reset(new PROBE CTOR_ARGS);                                            \
    return hp;                                                                     \
}                                                                                 \
void ref_##NAME##_destroy(void *p) { delete static_cast<RefPop<PROBE>*>(p); }    \
void ref_##NAME##_init(void *p, int obj, int n, const double *lower,             \
        const double *upper, const double *guess)                                 \
{                                                                                 \
    auto *hp = static_cast<RefPop<PROBE>*>(p);                                     \
    bind_problem(hp, obj, n, lower, upper, guess);                                 \
    multivariate_problem prob { make_objective(hp->ctx), n, hp->lower.data(),      \
            hp->upper.data() };                                                    \
    hp->alg->init(prob, hp->guess.data());                                          \
}                                                                                 \
void ref_##NAME##_iterate(void *p)                                                \
{                                                                                 \
    static_cast<RefPop<PROBE>*>(p)->alg->iterate();                               \
}                                                                                 \
int ref_##NAME##_get(void *p, const char *key, double *out, int cap)             \
{                                                                                 \
    return static_cast<RefPop<PROBE>*>(p)->alg->get(key, out, cap);               \
}                                                                                 \
void ref_##NAME##_solution(void *p, double *x_out, int *fev, int *converged)     \
{                                                                                 \
    auto *hp = static_cast<RefPop<PROBE>*>(p);                                     \
    const auto sol = hp->alg->solution();                                          \
    for (size_t i = 0; i < sol._sol.size(); i++) x_out[i] = sol._sol[i];          \
    *fev = sol._fev;                                                              \
    *converged = sol._converged ? 1 : 0;                                          \
}                                                                                 \
int ref_##NAME##_optimize(void *p, int obj, int n, const double *lower,          \
        const double *upper, const double *guess, double *x_out, int *fev,       \
        int *converged)                                                           \
{                                                                                 \
    auto *hp = static_cast<RefPop<PROBE>*>(p);                                     \
    bind_problem(hp, obj, n, lower, upper, guess);                                 \
    multivariate_problem prob { make_objective(hp->ctx), n, hp->lower.data(),      \
            hp->upper.data() };                                                    \
    const auto sol = hp->alg->optimize(prob, hp->guess.data());                     \
    for (int i = 0; i < n; i++) x_out[i] = sol._sol[i];                           \
    *fev = sol._fev;                                                              \
    *converged = sol._converged ? 1 : 0;                                          \
    return 0;                                                                     \
}

POP_API(shade, ShadeProbe,
        (int mfev, int npinit, double tol, int archive, int repaircr, int h, int npmin),
        (mfev, npinit, tol, archive != 0, repaircr != 0, h, npmin))

POP_API(jade, JadeProbe,
        (int mfev, int np, double tol, int archive, int repaircr, double pelite,
                double cdamp, double sigma),
        (mfev, np, tol, archive != 0, repaircr != 0, pelite, cdamp, sigma))

POP_API(sansde, SansdeProbe,
        (int mfev, int np, double tol, int repaircr, int crref, int pupdate, int crupdate),
        (mfev, np, tol, repaircr != 0, crref, pupdate, crupdate))

POP_API(cso, CsoProbe,
        (int mfev, double stol, int np, int pcompete, int ring, int correct, double vmax),
        (mfev, stol, np, pcompete, ring != 0, correct != 0, vmax))

POP_API(ccpso, CcpsoProbe,
        (int mfev, double stol, int np, const int *pps, int npps, int correct, double pcauchy),
        (mfev, stol, np, const_cast<int*>(pps), npps, correct != 0, pcauchy, nullptr, 10))

/* CCPSO with its local optimizer: takes ownership of `base` (a ref_cma_create handle) */
void ref_ccpso_set_local(void *p, void *base, int localfreq, unsigned long long)
{
    auto *hp = static_cast<RefPop<CcpsoProbe>*>(p);
    hp->base.reset(static_cast<RefCma*>(base));
    hp->alg->set_local(hp->base->alg.get(), localfreq);
}

POP_API(apso, ApsoProbe,
        (int mfev, double tol, int np, int correct),
        (mfev, tol, np, correct != 0))

/* ---- restart drivers ---------------------------------------------------- */
void* ref_bipop_create(void *base, int mfev, double sigma0, int maxlargeruns,
        int nbipop, double ksigmadec, double kbudget)
{
    auto *h = new RefPop<BiPopProbe>();
    h->base.reset(static_cast<RefCma*>(base));   /* takes ownership */
    h->alg.reset(new BiPopProbe(h->base->alg.get(), mfev, false, sigma0,
            maxlargeruns, nbipop != 0, ksigmadec, kbudget));
    return h;
}
void ref_bipop_destroy(void *p) { delete static_cast<RefPop<BiPopProbe>*>(p); }
void ref_bipop_set_print(void *p, int on)
{
    static_cast<RefPop<BiPopProbe>*>(p)->alg->set_print(on != 0);
}
void ref_bipop_init(void *p, int obj, int n, const double *lower,
        const double *upper, const double *guess)
{
    auto *h = static_cast<RefPop<BiPopProbe>*>(p);
    bind_problem(h, obj, n, lower, upper, guess);
    multivariate_problem prob { make_objective(h->ctx), n, h->lower.data(),
            h->upper.data() };
    h->alg->init(prob, h->guess.data());
}
void ref_bipop_iterate(void *p) { static_cast<RefPop<BiPopProbe>*>(p)->alg->iterate(); }
int ref_bipop_get(void *p, const char *key, double *out, int cap)
{
    return static_cast<RefPop<BiPopProbe>*>(p)->alg->get(key, out, cap);
}
int ref_bipop_inner_get(void *p, const char *key, double *out, int cap)
{
    return static_cast<RefPop<BiPopProbe>*>(p)->base->get(key, out, cap);
}
int ref_bipop_optimize(void *p, int obj, int n, const double *lower,
        const double *upper, const double *guess, double *x_out, int *fev,
        int *converged)
{
    auto *h = static_cast<RefPop<BiPopProbe>*>(p);
    bind_problem(h, obj, n, lower, upper, guess);
    multivariate_problem prob { make_objective(h->ctx), n, h->lower.data(),
            h->upper.data() };
    const auto sol = h->alg->optimize(prob, h->guess.data());
    for (int i = 0; i < n; i++) x_out[i] = sol._sol[i];
    *fev = sol._fev;
    *converged = sol._converged ? 1 : 0;
    return 0;
}

void* ref_ipop_create(void *base, int mfev, double sigma0, int nipop,
        double ksigmadec, int boundlambda)
{
    auto *h = new RefPop<IPopProbe>();
    h->base.reset(static_cast<RefCma*>(base));
    h->alg.reset(new IPopProbe(h->base->alg.get(), mfev, false, sigma0,
            nipop != 0, ksigmadec, boundlambda != 0));
    return h;
}
void ref_ipop_destroy(void *p) { delete static_cast<RefPop<IPopProbe>*>(p); }
void ref_ipop_set_print(void *p, int on)
{
    static_cast<RefPop<IPopProbe>*>(p)->alg->set_print(on != 0);
}
void ref_ipop_init(void *p, int obj, int n, const double *lower,
        const double *upper, const double *guess)
{
    auto *h = static_cast<RefPop<IPopProbe>*>(p);
    bind_problem(h, obj, n, lower, upper, guess);
    multivariate_problem prob { make_objective(h->ctx), n, h->lower.data(),
            h->upper.data() };
    h->alg->init(prob, h->guess.data());
}
void ref_ipop_iterate(void *p) { static_cast<RefPop<IPopProbe>*>(p)->alg->iterate(); }
int ref_ipop_get(void *p, const char *key, double *out, int cap)
{
    return static_cast<RefPop<IPopProbe>*>(p)->alg->get(key, out, cap);
}
int ref_ipop_optimize(void *p, int obj, int n, const double *lower,
        const double *upper, const double *guess, double *x_out, int *fev,
        int *converged)
{
    auto *h = static_cast<RefPop<IPopProbe>*>(p);
    bind_problem(h, obj, n, lower, upper, guess);
    multivariate_problem prob { make_objective(h->ctx), n, h->lower.data(),
            h->upper.data() };
    const auto sol = h->alg->optimize(prob, h->guess.data());
    for (int i = 0; i < n; i++) x_out[i] = sol._sol[i];
    *fev = sol._fev;
    *converged = sol._converged ? 1 : 0;
    return 0;
}

double ref_objective(int obj, int n, const double *x)
{
    auto ctx = make_ctx(obj, n);
    return bbo_objective_eval(obj, n, x, ctx->aux.data());
}

} // extern "C"
