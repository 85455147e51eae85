// bbo_common.hpp -- host-side plumbing shared by the optimizer engines:
// error propagation to the C ABI, owned device buffers, the optimizer base class
// that mirrors MultivariateOptimizer (/root/reference/src/multivariate/multivariate.h:132-146).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/bbopt_hip.h"

namespace bbo {

struct Error: std::runtime_error {
    int status;
    Error(int st, const std::string &msg) :
            std::runtime_error(msg), status(st) {
    }
};

#define BBO_HIP(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess)                                                      \
            throw ::bbo::Error(BBO_ERR_HIP, std::string(#expr) + ": "              \
                    + hipGetErrorString(e_));                                      \
    } while (0)

#define BBO_REQUIRE(cond, msg)                                                     \
    do {                                                                           \
        if (!(cond)) throw ::bbo::Error(BBO_ERR_ARG, msg);                         \
    } while (0)

// Kernels that stage one row of ld doubles per 16-lane group (de_init, de_generation,
// sansde_generation, pso_init, pso_update, cso_init, cso_compete): rows per workgroup so that
// the stage stays within 64 KiB of LDS (16 rows up to ld = 512, 8 up to 1024, 4 up to 2048);
// the kernels take their row count from blockDim.x >> 4.
inline int rows_per_wg16(int ld)
{
    return ld <= 512 ? 16 : ld <= 1024 ? 8 : 4;
}

inline int round_up(int v, int m)
{
    return (v + m - 1) / m * m;
}

// an owned, zero-initialised device allocation
template<class T>
struct DevBuf {
    T *p = nullptr;
    size_t count = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) (void) hipFree(p);
        p = nullptr;
        count = 0;
    }
    void alloc(size_t n)
    {
        release();
        count = n;
        if (n == 0) return;
        BBO_HIP(hipMalloc((void**) &p, n * sizeof(T)));
        BBO_HIP(hipMemset(p, 0, n * sizeof(T)));
    }
    void upload(const T *src, size_t n, size_t offset = 0)
    {
        BBO_HIP(hipMemcpy(p + offset, src, n * sizeof(T), hipMemcpyHostToDevice));
    }
    void download(T *dst, size_t n, size_t offset = 0) const
    {
        BBO_HIP(hipMemcpy(dst, p + offset, n * sizeof(T), hipMemcpyDeviceToHost));
    }
};

// Per-kernel device timing with HIP events on the engine's own stream (bench.py's
// roofline leg): disabled by default; when enabled every launch is bracketed by an event
// pair, collect() (after a stream sync) folds the elapsed times into per-kernel totals.
// roctx ranges: with the `profile` switch on, every phase of a generation is also
// a named range on the host thread -- rocprofv3 --marker-trace shows "bbo:cma_eigen" over the
// launches it brackets.  The marker library is looked up at run time (libroctx64 / the
// rocprofiler-sdk one, whichever the process finds): the product links against nothing it may
// not find on a box, and without the library the ranges are simply not emitted.
struct Roctx {
    using push_fn = int (*)(const char*);
    using pop_fn = int (*)();
    push_fn push = nullptr;
    pop_fn pop = nullptr;
    static const Roctx &get();
};

class KernelTimer {
public:
    ~KernelTimer()
    {
        for (auto &e : pool_) {
            (void) hipEventDestroy(e.a);
            (void) hipEventDestroy(e.b);
        }
    }
    // names: one static string per slot ("bbo:<phase>") for the roctx ranges, or null
    void enable(bool on, int nslots, const char *const *names = nullptr)
    {
        on_ = on;
        names_ = names;
        ms_.assign(nslots, 0.);
        calls_.assign(nslots, 0);
        used_ = 0;
    }
    bool on() const { return on_; }
    void begin(hipStream_t st, int slot)
    {
        if (!on_) return;
        if (names_ && Roctx::get().push) {
            Roctx::get().push(names_[slot]);
            open_ = true;
        }
        if (used_ == pool_.size()) {
            Pair p;
            BBO_HIP(hipEventCreate(&p.a));
            BBO_HIP(hipEventCreate(&p.b));
            pool_.push_back(p);
        }
        pool_[used_].slot = slot;
        BBO_HIP(hipEventRecord(pool_[used_].a, st));
    }
    void end(hipStream_t st)
    {
        if (!on_) return;
        BBO_HIP(hipEventRecord(pool_[used_].b, st));
        used_++;
        if (open_) {
            Roctx::get().pop();
            open_ = false;
        }
    }
    void collect()   // call after the stream has been synchronised
    {
        for (size_t i = 0; i < used_; i++) {
            float ms = 0.f;
            BBO_HIP(hipEventElapsedTime(&ms, pool_[i].a, pool_[i].b));
            ms_[pool_[i].slot] += ms;
            calls_[pool_[i].slot]++;
        }
        used_ = 0;
    }
    int report(double *out, int cap) const   // [ms0, calls0, ms1, calls1, ...]
    {
        const int cnt = 2 * (int) ms_.size();
        if (out && cap >= cnt)
            for (size_t i = 0; i < ms_.size(); i++) {
                out[2 * i] = ms_[i];
                out[2 * i + 1] = calls_[i];
            }
        return cnt;
    }
private:
    struct Pair { hipEvent_t a, b; int slot; };
    std::vector<Pair> pool_;
    std::vector<double> ms_;
    std::vector<long> calls_;
    size_t used_ = 0;
    bool on_ = false;
    bool open_ = false;
    const char *const *names_ = nullptr;
};

// Raises a kernel's dynamic-LDS ceiling (hipFuncAttributeMaxDynamicSharedMemorySize), once
// per (device, kernel): the attribute belongs to the device the call is made on, and a process
// may hold handles on several devices.
inline void allow_lds(const void *fn, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    BBO_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({ dev, fn })) return;
    BBO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({ dev, fn });
}

// how the population's fitness is obtained
struct ObjectiveSpec {
    int kind = BBO_OBJECTIVE_BUILTIN;
    int builtin = BBO_OBJ_ROSENBROCK;
    bbo_scalar_fn scalar = nullptr;
    bbo_batch_fn batch = nullptr;
    void *user = nullptr;

    bool on_device() const { return kind == BBO_OBJECTIVE_BUILTIN; }
    // evaluates `rows` host-resident candidates through the callbacks
    void eval_host(const double *X, int rows, int n, int ld, double *f_out) const
    {
        if (kind == BBO_OBJECTIVE_BATCH_CALLBACK) {
            if (batch(X, rows, n, ld, f_out, user) != 0)
                throw Error(BBO_ERR_CALLBACK, "objective batch callback failed");
            return;
        }
        for (int r = 0; r < rows; r++) {
            int failed = 0;
            f_out[r] = scalar(X + (size_t) r * ld, n, user, &failed);
            if (failed) throw Error(BBO_ERR_CALLBACK, "objective callback failed");
        }
    }
};

// per-coordinate objective constants; must equal bbo_objective_aux() of
// oracle/objectives.h so the checker and the kernels see the same table
inline void fill_objective_aux(int obj, int n, double *aux)
{
    for (int i = 0; i < n; i++) {
        const double t = (n > 1) ? ((double) i) / (double) (n - 1) : 0.;
        switch (obj) {
        case BBO_OBJ_ELLIPSOID: aux[i] = pow(10., 6. * t); break;
        case BBO_OBJ_DIFFPOW:   aux[i] = 2. + 4. * t; break;
        case BBO_OBJ_GRIEWANK:  aux[i] = 1. / sqrt((double) (i + 1)); break;
        default:                aux[i] = 0.; break;
        }
    }
}

// a built-in objective at ONE point on the host: the same definition and the same per-coordinate
// table as the device kernels (bbo_objectives.hpp).  For the few places that evaluate a single
// point between device generations: the restart drivers' extra evaluation (bipop_cmaes.cpp:86),
// CCPSO's local search (ccpso.cpp:371-435).
inline double builtin_objective_host(int obj, int n, const double *x, const double *aux)
{
    const double two_pi = 6.283185307179586476925286766559, euler_e = 2.718281828459045235360287471352;
    double s = 0.;
    switch (obj) {
    case BBO_OBJ_SPHERE:
        for (int i = 0; i < n; i++) s += x[i] * x[i];
        return s;
    case BBO_OBJ_ROSENBROCK:
        for (int i = 0; i + 1 < n; i++) {
            const double a = x[i + 1] - x[i] * x[i], b = 1. - x[i];
            s += 100. * (a * a) + b * b;
        }
        return s;
    case BBO_OBJ_RASTRIGIN:
        for (int i = 0; i < n; i++) s += x[i] * x[i] - 10. * std::cos(two_pi * x[i]);
        return 10. * n + s;
    case BBO_OBJ_ELLIPSOID:
        for (int i = 0; i < n; i++) s += aux[i] * (x[i] * x[i]);
        return s;
    case BBO_OBJ_ACKLEY: {
        double cs = 0.;
        for (int i = 0; i < n; i++) {
            s += x[i] * x[i];
            cs += std::cos(two_pi * x[i]);
        }
        return -20. * std::exp(-0.2 * std::sqrt(s / n)) - std::exp(cs / n) + 20. + euler_e;
    }
    case BBO_OBJ_GRIEWANK: {
        double pr = 1.;
        for (int i = 0; i < n; i++) {
            s += x[i] * x[i];
            pr *= std::cos(x[i] * aux[i]);
        }
        return 1. + s / 4000. - pr;
    }
    case BBO_OBJ_CIGAR:
        for (int i = 1; i < n; i++) s += x[i] * x[i];
        return x[0] * x[0] + 1.0e6 * s;
    case BBO_OBJ_DISCUS:
        for (int i = 1; i < n; i++) s += x[i] * x[i];
        return 1.0e6 * (x[0] * x[0]) + s;
    case BBO_OBJ_DIFFPOW:
        for (int i = 0; i < n; i++) s += std::pow(std::fabs(x[i]), aux[i]);
        return s;
    case BBO_OBJ_SCHWEFEL12: {
        double run = 0.;
        for (int i = 0; i < n; i++) {
            run += x[i];
            s += run * run;
        }
        return s;
    }
    default:
        throw Error(BBO_ERR_ARG, "unknown builtin objective");
    }
}

// The C++ statement of MultivariateOptimizer (multivariate.h:132-146) that every
// engine implements and the C ABI dispatches to.
class Optimizer {
public:
    virtual ~Optimizer() {}
    virtual void init(int n, const double *lower, const double *upper, const double *guess,
            const ObjectiveSpec &obj) = 0;
    virtual void iterate() = 0;
    virtual void solution(int population, double *x_out, int *n_evals, int *converged) = 0;
    // init + loop until the algorithm's own stop rule or the evaluation budget
    virtual void optimize(int n, const double *lower, const double *upper,
            const double *guess, const ObjectiveSpec &obj, double *x_out, int *n_evals,
            int *converged) = 0;
    virtual int run(int max_generations) = 0;
    virtual int get(const std::string &key, int population, double *out, int cap) = 0;
    virtual int set(const std::string &key, int population, const double *in, int count) = 0;
    virtual int dimension() const = 0;

    std::string last_error;
};

} // namespace bbo
