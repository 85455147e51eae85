/* integration/adapter_check.cpp -- compiled by tests/test_integration_adapter.py against the
 * reference's own src/multivariate/multivariate.h and this repository's include/bbopt_hip.h,
 * linked with bboptpy_amd/libbbopt_hip.so.  Uses HipOptimizer exactly as the reference uses any
 * MultivariateOptimizer (README example: Rosenbrock, n = 10, box [-10, 10]).  Without a GPU the
 * constructor must throw the reference's error type with the library's "no device" message
 * (exit 3); with one it optimizes and prints the solution (exit 0).
 */
#include <cstdio>
#include <cstring>
#include <memory>

#include "hip_optimizer.h"

static double rosenbrock10(const double *x) {
    double s = 0.;
    for (int i = 0; i < 9; i++) {
        const double a = x[i + 1] - x[i] * x[i], b = 1. - x[i];
        s += 100. * a * a + b * b;
    }
    return s;
}

int main() {
    bbo_params p;
    std::memset(&p, 0, sizeof p);
    bbo_params_default(&p, BBO_ALGO_ACTIVE_CMAES);
    p.mfev = 20000;
    p.tol = 1e-6;
    p.np = 20;
    std::unique_ptr<MultivariateOptimizer> opt;
    try {
        opt.reset(new HipOptimizer(p));
    } catch (const std::invalid_argument &e) {
        std::printf("invalid_argument: %s\n", e.what());
        return 3;
    }
    double lower[10], upper[10], guess[10];
    for (int i = 0; i < 10; i++) {
        lower[i] = -10.;
        upper[i] = 10.;
        guess[i] = -1.5 + 0.3 * i;
    }
    multivariate_problem prob(rosenbrock10, 10, lower, upper);      /* multivariate.h:61-65 */
    multivariate_solution sol = opt->optimize(prob, guess);
    std::printf("%s\n", sol.toString().c_str());
    /* and the stepwise interface */
    opt->init(prob, guess);
    for (int g = 0; g < 5; g++) opt->iterate();
    multivariate_solution s2 = opt->solution();
    std::printf("after 5 generations: %d evaluations\n", s2._fev);
    return rosenbrock10(sol._sol.data()) < 1e-6 ? 0 : 1;
}
