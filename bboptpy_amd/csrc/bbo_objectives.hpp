// bbo_objectives.hpp -- built-in objective functions, evaluated on the device.
//
// The reference evaluates a std::function one candidate at a time
// (/root/reference/src/multivariate/multivariate.h:32; call sites
// base_cmaes.cpp:214-217, shade.cpp:159, apso.cpp:188) and ships no objective of
// its own; its README objective is the Rosenbrock sum (README.md:111-112).  Here a
// whole population is evaluated inside the kernel that produced it: G lanes
// (a power of two, <= 64, contiguous in one wavefront) share one candidate row,
// each lane sums the coordinates j = g, g+G, ..., and the partial sums meet in a
// butterfly of wavefront shuffles.  Ids match bbo_objective_id in
// include/bbopt_hip.h and oracle/objectives.h (the CPU checker).
#pragma once

#include <hip/hip_runtime.h>

namespace bbo {

enum Objective : int {
    OBJ_HOST = -1,   // host callback: the kernel only produces X
    OBJ_SPHERE = 0,
    OBJ_ROSENBROCK = 1,
    OBJ_RASTRIGIN = 2,
    OBJ_ELLIPSOID = 3,
    OBJ_ACKLEY = 4,
    OBJ_GRIEWANK = 5,
    OBJ_CIGAR = 6,
    OBJ_DISCUS = 7,
    OBJ_DIFFPOW = 8,
    OBJ_SCHWEFEL12 = 9,
    OBJ_COUNT = 10
};

constexpr double TWO_PI = 6.283185307179586476925286766559;
constexpr double EULER_E = 2.718281828459045235360287471352;

template<int G>
__device__ inline double group_sum(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

template<int G>
__device__ inline double group_prod(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v *= __shfl_xor(v, off, G);
    return v;
}

// x: the candidate (LDS or global), n coordinates; aux: per-coordinate constants
// (see bbo_objective_aux in oracle/objectives.h -- the host fills the same table);
// g: this lane's index inside its group.  Every lane of the group returns f.
template<int G>
__device__ inline double eval_row_group(int obj, int n, const double *x, const double *aux,
        int g)
{
    double a = 0., b = 0.;
    switch (obj) {
    case OBJ_SPHERE:
        for (int j = g; j < n; j += G) a += x[j] * x[j];
        return group_sum<G>(a);
    case OBJ_ROSENBROCK:
        for (int j = g; j + 1 < n; j += G) {
            const double xj = x[j];
            const double t = x[j + 1] - xj * xj;
            const double u = 1. - xj;
            a += 100. * (t * t) + u * u;
        }
        return group_sum<G>(a);
    case OBJ_RASTRIGIN:
        for (int j = g; j < n; j += G) a += x[j] * x[j] - 10. * cos(TWO_PI * x[j]);
        return 10. * n + group_sum<G>(a);
    case OBJ_ELLIPSOID:
        for (int j = g; j < n; j += G) a += aux[j] * (x[j] * x[j]);
        return group_sum<G>(a);
    case OBJ_ACKLEY:
        for (int j = g; j < n; j += G) {
            a += x[j] * x[j];
            b += cos(TWO_PI * x[j]);
        }
        a = group_sum<G>(a);
        b = group_sum<G>(b);
        return -20. * exp(-0.2 * sqrt(a / n)) - exp(b / n) + 20. + EULER_E;
    case OBJ_GRIEWANK:
        b = 1.;
        for (int j = g; j < n; j += G) {
            a += x[j] * x[j];
            b *= cos(x[j] * aux[j]);
        }
        a = group_sum<G>(a);
        b = group_prod<G>(b);
        return 1. + a / 4000. - b;
    case OBJ_CIGAR:
        for (int j = g; j < n; j += G)
            if (j > 0) a += x[j] * x[j];
        return x[0] * x[0] + 1.0e6 * group_sum<G>(a);
    case OBJ_DISCUS:
        for (int j = g; j < n; j += G)
            if (j > 0) a += x[j] * x[j];
        return 1.0e6 * (x[0] * x[0]) + group_sum<G>(a);
    case OBJ_DIFFPOW:
        for (int j = g; j < n; j += G) a += pow(fabs(x[j]), aux[j]);
        return group_sum<G>(a);
    case OBJ_SCHWEFEL12: {
        // prefix sums are sequential: lane 0 of the group walks the row
        if (g == 0) {
            double run = 0.;
            for (int j = 0; j < n; j++) {
                run += x[j];
                a += run * run;
            }
        }
        return group_sum<G>(a);
    }
    default:
        return 0.;
    }
}

} // namespace bbo
