/*
 * oracle/objectives.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Synthetic BBOB-style objective functions evaluated one candidate at a time,
 * serial left-to-right summation, no FMA contraction (build with
 * -ffp-contract=off).  The reference ships NO objective functions; the only one
 * it documents is the README Rosenbrock `fx` (/root/reference/README.md:111-112,
 * examples/iterative.py:6-10), which BBO_OBJ_ROSENBROCK restates.  The same ids
 * are used by include/bbopt_hip.h (bbo_objective_id) and by the HIP kernels.
 *
 * Included by oracle/bbo_oracle.cpp (the CPU restatement) and by
 * oracle/ref_harness.cpp (the harness that drives the real reference), so both
 * see bit-identical objective values.
 */
#ifndef BBO_ORACLE_OBJECTIVES_H_
#define BBO_ORACLE_OBJECTIVES_H_

#include <math.h>

enum {
    BBO_OBJ_SPHERE = 0,
    BBO_OBJ_ROSENBROCK = 1,
    BBO_OBJ_RASTRIGIN = 2,
    BBO_OBJ_ELLIPSOID = 3,
    BBO_OBJ_ACKLEY = 4,
    BBO_OBJ_GRIEWANK = 5,
    BBO_OBJ_CIGAR = 6,
    BBO_OBJ_DISCUS = 7,
    BBO_OBJ_DIFFPOW = 8,
    BBO_OBJ_SCHWEFEL12 = 9,
    BBO_OBJ_COUNT = 10
};

#define BBO_TWO_PI 6.283185307179586476925286766559

/* cos(2 pi x): exact reduction of x to [-1/2, 1/2] and to an octant, then the fdlibm kernel
 * polynomials -- the same arithmetic, operation for operation, as cos_2pi in
 * bboptpy_amd/csrc/bbo_objectives.hpp, so the device's Rastrigin / Ackley values differ from
 * these only by the order of the final sum.  |error| <= 2 ulp of 1 against libm. */
static inline double bbo_cos_2pi(double x)
{
    const double t = fabs(x - rint(x));
    const double v = t * 8.;
    int k = (int) v;
    k = k > 3 ? 3 : k;
    double f = v - (double) k;
    if (k & 1) f = 1. - f;
    const double y = f * 0x1.921fb54442d18p-1;
    const double z = y * y;
    const int q = (k + 1) >> 1;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sy = fma(y * z, ps, y);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cy = fma(z * z, pc, fma(z, -0.5, 1.));
    const double c1 = k == 1 ? sy : -sy;
    return q == 0 ? cy : (q == 1 ? c1 : -cy);
}

/* per-coordinate constants some objectives need; aux must hold n doubles.
 * ELLIPSOID: 10^(6 i/(n-1)); DIFFPOW: exponent 2+4 i/(n-1); GRIEWANK: 1/sqrt(i+1). */
static inline void bbo_objective_aux(int obj, int n, double *aux)
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

static inline double bbo_objective_eval(int obj, int n, const double *x,
        const double *aux)
{
    double s = 0.;
    switch (obj) {
    case BBO_OBJ_SPHERE:
        for (int i = 0; i < n; i++) s += x[i] * x[i];
        return s;
    case BBO_OBJ_ROSENBROCK:
        /* README.md:111-112: sum 100 (x2 - x1^2)^2 + (1 - x1)^2 */
        for (int i = 0; i + 1 < n; i++) {
            const double a = x[i + 1] - x[i] * x[i];
            const double b = 1. - x[i];
            s += 100. * (a * a) + b * b;
        }
        return s;
    case BBO_OBJ_RASTRIGIN:
        for (int i = 0; i < n; i++)
            s += x[i] * x[i] - 10. * bbo_cos_2pi(x[i]);
        return 10. * n + s;
    case BBO_OBJ_ELLIPSOID:
        for (int i = 0; i < n; i++) s += aux[i] * (x[i] * x[i]);
        return s;
    case BBO_OBJ_ACKLEY: {
        double c = 0.;
        for (int i = 0; i < n; i++) {
            s += x[i] * x[i];
            c += bbo_cos_2pi(x[i]);
        }
        return -20. * exp(-0.2 * sqrt(s / n)) - exp(c / n) + 20.
                + 2.718281828459045235360287471352;
    }
    case BBO_OBJ_GRIEWANK: {
        double p = 1.;
        for (int i = 0; i < n; i++) {
            s += x[i] * x[i];
            p *= cos(x[i] * aux[i]);
        }
        return 1. + s / 4000. - p;
    }
    case BBO_OBJ_CIGAR:
        for (int i = 1; i < n; i++) s += x[i] * x[i];
        return x[0] * x[0] + 1.0e6 * s;
    case BBO_OBJ_DISCUS:
        for (int i = 1; i < n; i++) s += x[i] * x[i];
        return 1.0e6 * (x[0] * x[0]) + s;
    case BBO_OBJ_DIFFPOW:
        for (int i = 0; i < n; i++) s += pow(fabs(x[i]), aux[i]);
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
        return NAN;
    }
}

#endif /* BBO_ORACLE_OBJECTIVES_H_ */
