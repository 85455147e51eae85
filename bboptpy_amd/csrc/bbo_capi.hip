// bbo_capi.hip -- the extern "C" surface declared in include/bbopt_hip.h.
// Each entry point catches bbo::Error / std::exception and turns it into a status code
// plus a message retrievable with bbo_last_error(), so no C++ exception crosses the ABI.
#include "bbo_cma.hpp"
#include "bbo_ccpso.hpp"

#include <memory>
#include <mutex>

namespace bbo {
Optimizer* make_de_engine(const bbo_params &p);       // bbo_de.hip
Optimizer* make_pso_engine(const bbo_params &p);      // bbo_pso.hip
Optimizer* make_cso_engine(const bbo_params &p);      // bbo_cso.hip
Optimizer* make_ccpso_engine(const bbo_params &p);    // bbo_ccpso.hip
Optimizer* make_restart_driver(const bbo_params &p, Optimizer *base);   // bbo_restart.hip
}

struct bbo_handle_s {
    std::unique_ptr<bbo::Optimizer> opt;
    std::string error;
    int algo = 0;
};

namespace {

std::string g_create_error;
std::mutex g_create_mutex;

template<class F>
int guarded(bbo_handle h, F fn)
{
    if (!h || !h->opt) return BBO_ERR_ARG;
    try {
        fn();
        return BBO_OK;
    } catch (const bbo::Error &e) {
        h->error = e.what();
        return e.status;
    } catch (const std::exception &e) {
        h->error = e.what();
        return BBO_ERR_HIP;
    }
}

bbo::ObjectiveSpec to_spec(const bbo_objective *o)
{
    if (!o) throw bbo::Error(BBO_ERR_ARG, "objective must not be NULL");
    bbo::ObjectiveSpec s;
    s.kind = o->kind;
    s.builtin = o->builtin;
    s.scalar = o->scalar;
    s.batch = o->batch;
    s.user = o->user;
    if (s.kind == BBO_OBJECTIVE_BUILTIN) {
        if (s.builtin < 0 || s.builtin > BBO_OBJ_SCHWEFEL12)
            throw bbo::Error(BBO_ERR_ARG, "unknown builtin objective id");
    } else if (s.kind == BBO_OBJECTIVE_SCALAR_CALLBACK) {
        if (!s.scalar) throw bbo::Error(BBO_ERR_ARG, "scalar callback is NULL");
    } else if (s.kind == BBO_OBJECTIVE_BATCH_CALLBACK) {
        if (!s.batch) throw bbo::Error(BBO_ERR_ARG, "batch callback is NULL");
    } else {
        throw bbo::Error(BBO_ERR_ARG, "unknown objective kind");
    }
    return s;
}

} // namespace

#include <dlfcn.h>
namespace bbo {
const Roctx &Roctx::get()
{
    static const Roctx r = [] {
        Roctx x;
        for (const char *lib : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4",
                "libroctx64.so" }) {
            void *h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            x.push = reinterpret_cast<Roctx::push_fn>(dlsym(h, "roctxRangePushA"));
            x.pop = reinterpret_cast<Roctx::pop_fn>(dlsym(h, "roctxRangePop"));
            if (x.push && x.pop) break;
            x.push = nullptr;
            x.pop = nullptr;
        }
        return x;
    }();
    return r;
}
} // namespace bbo

extern "C" {

void bbo_params_default(bbo_params *p, int algo)
{
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->algo = algo;
    // defaults of py/multivariate_py.cpp:103-171,265-269
    p->sigma0 = 2.;
    p->bound = 0;
    p->alphacov = 2.;
    p->eigenrate = 0.25;
    p->archive = 1;
    p->repaircr = 1;
    p->pelite = 0.05;
    p->cdamp = 0.1;
    p->jade_sigma = 0.07;
    p->h = 100;
    p->npmin = 4;
    p->correct = 1;
    p->print = 0;
    p->nipop = 1;
    p->ksigmadec = 1.6;
    p->boundlambda = 1;
    p->maxlargeruns = 9;
    p->kbudget = 2.;
    p->seed = 0x9E3779B97F4A7C15ull;
    p->device = 0;
    p->populations = 1;
    p->poll_every = 8;
    p->adjustlr = 1;   /* py/multivariate_py.cpp:131-135: "adjustlr"_a = true */
    p->crref = 5;
    p->pupdate = 50;
    p->crupdate = 25;
    p->pcompete = 3;
    p->ring = 0;
    p->vmax = 0.2;
    p->npps = 0;
    p->pcauchy = -1.;
}

int bbo_create(const bbo_params *params, bbo_handle *out)
{
    std::lock_guard<std::mutex> lock(g_create_mutex);
    if (!params || !out) {
        g_create_error = "bbo_create: NULL argument";
        return BBO_ERR_ARG;
    }
    *out = nullptr;
    try {
        std::unique_ptr<bbo_handle_s> h(new bbo_handle_s());
        h->algo = params->algo;
        switch (params->algo) {
        case BBO_ALGO_CMAES:
        case BBO_ALGO_ACTIVE_CMAES:
        case BBO_ALGO_SEP_CMAES:
            h->opt.reset(new bbo::CmaEngine(*params));
            break;
        case BBO_ALGO_SHADE:
        case BBO_ALGO_JADE:
        case BBO_ALGO_SANSDE:
            h->opt.reset(bbo::make_de_engine(*params));
            break;
        case BBO_ALGO_APSO:
            h->opt.reset(bbo::make_pso_engine(*params));
            break;
        case BBO_ALGO_CSO:
            h->opt.reset(bbo::make_cso_engine(*params));
            break;
        case BBO_ALGO_CCPSO:
            h->opt.reset(bbo::make_ccpso_engine(*params));
            break;
        default:
            throw bbo::Error(BBO_ERR_ARG,
                    "bbo_create: unknown algo (restart drivers use bbo_create_restart)");
        }
        *out = h.release();
        return BBO_OK;
    } catch (const bbo::Error &e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception &e) {
        g_create_error = e.what();
        return BBO_ERR_HIP;
    }
}

int bbo_create_restart(const bbo_params *params, bbo_handle base, bbo_handle *out)
{
    std::lock_guard<std::mutex> lock(g_create_mutex);
    if (!params || !out || !base || !base->opt) {
        g_create_error = "bbo_create_restart: NULL argument";
        return BBO_ERR_ARG;
    }
    *out = nullptr;
    try {
        if (params->algo != BBO_ALGO_IPOP_CMAES && params->algo != BBO_ALGO_BIPOP_CMAES)
            throw bbo::Error(BBO_ERR_ARG, "bbo_create_restart: algo must be IPOP or BIPOP");
        if (base->algo != BBO_ALGO_CMAES && base->algo != BBO_ALGO_ACTIVE_CMAES
                && base->algo != BBO_ALGO_SEP_CMAES)
            throw bbo::Error(BBO_ERR_ARG, "bbo_create_restart: base must be a CMA-ES handle");
        std::unique_ptr<bbo_handle_s> h(new bbo_handle_s());
        h->algo = params->algo;
        h->opt.reset(bbo::make_restart_driver(*params, base->opt.get()));
        *out = h.release();
        return BBO_OK;
    } catch (const bbo::Error &e) {
        g_create_error = e.what();
        return e.status;
    } catch (const std::exception &e) {
        g_create_error = e.what();
        return BBO_ERR_HIP;
    }
}

int bbo_destroy(bbo_handle h)
{
    delete h;
    return BBO_OK;
}

int bbo_init(bbo_handle h, int n, const double *lower, const double *upper,
        const double *guess, const bbo_objective *objective)
{
    return guarded(h, [&] {
        if (!lower || !upper || !guess) throw bbo::Error(BBO_ERR_ARG, "NULL bound/guess");
        h->opt->init(n, lower, upper, guess, to_spec(objective));
    });
}

int bbo_iterate(bbo_handle h)
{
    return guarded(h, [&] { h->opt->iterate(); });
}

int bbo_solution(bbo_handle h, double *x_out, int *n_evals, int *converged)
{
    return bbo_solution_of(h, 0, x_out, n_evals, converged);
}

int bbo_solution_of(bbo_handle h, int population, double *x_out, int *n_evals,
        int *converged)
{
    return guarded(h, [&] {
        if (!x_out || !n_evals || !converged) throw bbo::Error(BBO_ERR_ARG, "NULL output");
        h->opt->solution(population, x_out, n_evals, converged);
    });
}

int bbo_optimize(bbo_handle h, int n, const double *lower, const double *upper,
        const double *guess, const bbo_objective *objective, double *x_out, int *n_evals,
        int *converged)
{
    return guarded(h, [&] {
        if (!lower || !upper || !guess || !x_out || !n_evals || !converged)
            throw bbo::Error(BBO_ERR_ARG, "NULL argument");
        h->opt->optimize(n, lower, upper, guess, to_spec(objective), x_out, n_evals,
                converged);
    });
}

int bbo_run(bbo_handle h, int max_generations, int *generations_done)
{
    return guarded(h, [&] {
        const int g = h->opt->run(max_generations);
        if (generations_done) *generations_done = g;
    });
}

int bbo_get(bbo_handle h, const char *key, int population, double *out, int cap)
{
    if (!h || !h->opt || !key) return BBO_ERR_ARG;
    try {
        return h->opt->get(key, population, out, cap);
    } catch (const bbo::Error &e) {
        h->error = e.what();
        return e.status;
    } catch (const std::exception &e) {
        h->error = e.what();
        return BBO_ERR_HIP;
    }
}

int bbo_set(bbo_handle h, const char *key, int population, const double *in, int count)
{
    if (!h || !h->opt || !key || !in) return BBO_ERR_ARG;
    try {
        return h->opt->set(key, population, in, count);
    } catch (const bbo::Error &e) {
        h->error = e.what();
        return e.status;
    } catch (const std::exception &e) {
        h->error = e.what();
        return BBO_ERR_HIP;
    }
}

int bbo_cma_phase_run(bbo_handle h, int phase)
{
    return guarded(h, [&] {
        auto *cma = dynamic_cast<bbo::CmaEngine*>(h->opt.get());
        if (!cma) throw bbo::Error(BBO_ERR_ARG, "not a CMA-ES handle");
        cma->phase(phase);
    });
}

int bbo_cma_inject_normals(bbo_handle h, const double *z, int count)
{
    return guarded(h, [&] {
        auto *cma = dynamic_cast<bbo::CmaEngine*>(h->opt.get());
        if (!cma) throw bbo::Error(BBO_ERR_ARG, "not a CMA-ES handle");
        cma->inject_normals(z, count);
    });
}

int bbo_cma_set_params(bbo_handle h, int np, double sigma0, int mfev)
{
    return guarded(h, [&] {
        auto *cma = dynamic_cast<bbo::CmaEngine*>(h->opt.get());
        if (!cma) throw bbo::Error(BBO_ERR_ARG, "not a CMA-ES handle");
        if (np < 4) throw bbo::Error(BBO_ERR_ARG, "CMA-ES needs np >= 4");
        cma->set_params(np, sigma0, mfev);
    });
}

int bbo_cma_set_seed(bbo_handle h, uint64_t seed)
{
    return guarded(h, [&] {
        auto *cma = dynamic_cast<bbo::CmaEngine*>(h->opt.get());
        if (!cma) throw bbo::Error(BBO_ERR_ARG, "not a CMA-ES handle");
        cma->set_seed(seed);
    });
}

int bbo_cma_evaluate(bbo_handle h, const double *x, double *f_out)
{
    return guarded(h, [&] {
        auto *cma = dynamic_cast<bbo::CmaEngine*>(h->opt.get());
        if (!cma) throw bbo::Error(BBO_ERR_ARG, "not a CMA-ES handle");
        if (!x || !f_out) throw bbo::Error(BBO_ERR_ARG, "NULL argument");
        if (cma->dimension() <= 0) throw bbo::Error(BBO_ERR_STATE, "evaluate before initialize()");
        *f_out = cma->evaluate_point(x);
    });
}

namespace {
bbo::CcpsoEngine* as_ccpso(bbo_handle h)
{
    auto *e = dynamic_cast<bbo::CcpsoEngine*>(h->opt.get());
    if (!e) throw bbo::Error(BBO_ERR_ARG, "not a CCPSO handle");
    return e;
}
}

int bbo_ccpso_set_shard(bbo_handle h, int rank, int world)
{
    return guarded(h, [&] { as_ccpso(h)->set_shard(rank, world); });
}

int bbo_ccpso_set_local(bbo_handle h, bbo_handle local, int localfreq)
{
    return guarded(h, [&] {
        if (local && !local->opt) throw bbo::Error(BBO_ERR_ARG, "bbo_ccpso_set_local: dead local handle");
        if (local == h) throw bbo::Error(BBO_ERR_ARG, "bbo_ccpso_set_local: local must be another optimizer");
        as_ccpso(h)->set_local(local ? local->opt.get() : nullptr, localfreq);
    });
}

int bbo_ccpso_phase(bbo_handle h, int phase)
{
    return guarded(h, [&] { as_ccpso(h)->phase(phase); });
}

int bbo_ccpso_table_record(bbo_handle h)
{
    if (!h || !h->opt) return BBO_ERR_ARG;
    try {
        return as_ccpso(h)->table_record();
    } catch (const bbo::Error &e) {
        h->error = e.what();
        return e.status;
    }
}

int bbo_ccpso_export_tables(bbo_handle h, double *dst, int device_memory)
{
    return guarded(h, [&] {
        if (!dst) throw bbo::Error(BBO_ERR_ARG, "NULL destination");
        as_ccpso(h)->export_tables(dst, device_memory != 0);
    });
}

int bbo_ccpso_merge_tables(bbo_handle h, const double *gathered, int world, int device_memory)
{
    return guarded(h, [&] {
        if (!gathered) throw bbo::Error(BBO_ERR_ARG, "NULL source");
        as_ccpso(h)->merge_tables(gathered, world, device_memory != 0);
    });
}

const char* bbo_last_error(bbo_handle h)
{
    if (!h) return g_create_error.c_str();
    return h->error.c_str();
}

const char* bbo_version(void)
{
    return "bbopt_hip 0.1 (gfx950)";
}

int bbo_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // extern "C"
