/* integration/hip_optimizer.h -- the adapter a maintainer of the reference would add as
 * src/multivariate/hip/hip_optimizer.h: a MultivariateOptimizer
 * (src/multivariate/multivariate.h:132-146) whose four virtuals forward to the C ABI of
 * include/bbopt_hip.h.  It includes the REFERENCE's header, so it only compiles inside the
 * reference tree (or with -I<reference>/src/multivariate); tests/test_integration_adapter.py
 * compiles and runs it in the build container, where /root/reference exists.
 */
#ifndef BBOPT_HIP_OPTIMIZER_ADAPTER_H_
#define BBOPT_HIP_OPTIMIZER_ADAPTER_H_

#include <stdexcept>
#include <string>
#include <vector>

#include "multivariate.h"                    /* the reference's: multivariate.h:25-146 */
#include "bbopt_hip.h"                       /* this repository's include/             */

class HipOptimizer : public MultivariateOptimizer {
    bbo_handle _h = nullptr;
    multivariate _f;                         /* std::function<double(const double*)>, :31 */
    int _n = 0;

    static double tramp(const double *x, int, void *self, int *) {
        return static_cast<HipOptimizer*>(self)->_f(x);
    }
    bbo_objective objective(const multivariate_problem &p) {
        _f = p._f;
        _n = p._n;
        bbo_objective o{};
        o.kind = BBO_OBJECTIVE_SCALAR_CALLBACK;
        o.scalar = &tramp;
        o.user = this;
        return o;
    }
    [[noreturn]] void fail(bbo_handle h) const {
        const char *m = bbo_last_error(h);
        throw std::invalid_argument(m ? m : "libbbopt_hip error");   /* the reference's error type */
    }

public:
    explicit HipOptimizer(const bbo_params &prm) {
        if (bbo_create(&prm, &_h) != BBO_OK) fail(nullptr);
    }
    /* the optimizers that take another optimizer, as the reference's constructors do
     * (IPopCmaes / BiPopCmaes(base, ...): ipop_cmaes.cpp:41, bipop_cmaes.cpp:41;
     *  CCPSOSearch(..., local, localfreq): ccpso.cpp:51-70): the other object is borrowed and must
     * outlive this one, exactly like the reference's raw pointer */
    HipOptimizer(const bbo_params &prm, HipOptimizer &base) {
        if (bbo_create_restart(&prm, base._h, &_h) != BBO_OK) fail(nullptr);
    }
    void setLocal(HipOptimizer *local, int localfreq) {       /* CCPSO only */
        if (bbo_ccpso_set_local(_h, local ? local->_h : nullptr, localfreq) < 0) fail(_h);
    }
    HipOptimizer(const HipOptimizer&) = delete;
    HipOptimizer &operator=(const HipOptimizer&) = delete;
    ~HipOptimizer() override { bbo_destroy(_h); }

    void init(const multivariate_problem &p, const double *guess) override {
        bbo_objective o = objective(p);
        if (bbo_init(_h, p._n, p._lower, p._upper, guess, &o) != BBO_OK) fail(_h);
    }
    void iterate() override {
        if (bbo_iterate(_h) < 0) fail(_h);
    }
    multivariate_solution solution() override {
        std::vector<double> x(_n);
        int fev = 0, conv = 0;
        if (bbo_solution(_h, x.data(), &fev, &conv) < 0) fail(_h);
        return multivariate_solution(x, fev, conv != 0);
    }
    multivariate_solution optimize(const multivariate_problem &p, const double *guess) override {
        bbo_objective o = objective(p);
        std::vector<double> x(p._n);
        int fev = 0, conv = 0;
        if (bbo_optimize(_h, p._n, p._lower, p._upper, guess, &o, x.data(), &fev, &conv) < 0)
            fail(_h);
        return multivariate_solution(x, fev, conv != 0);
    }
};

#endif
