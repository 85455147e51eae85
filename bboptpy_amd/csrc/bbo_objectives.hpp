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

// cos(2 pi x) for any finite x: x is reduced to t = x - rint(x) in [-1/2, 1/2] EXACTLY, then to
// an octant, then the fdlibm kernel polynomials on [0, pi/4] -- no Payne-Hanek path, about a
// third of ocml's cos(2 pi x) (which first rounds 2 pi x).  |error| <= 2 ulp of 1.
__device__ inline double cos_2pi(double x)
{
    double t = fabs(x - rint(x));            // [0, 1/2], exact
    const double v = t * 8.;                 // [0, 4]
    int k = (int) v;                         // octant 0..4 (4 only at t = 1/2)
    k = k > 3 ? 3 : k;
    double f = v - (double) k;               // [0, 1]
    const bool odd = (k & 1) != 0;
    f = odd ? 1. - f : f;
    const double y = f * 0x1.921fb54442d18p-1;    // * pi/4
    const double z = y * y;
    // angle = q pi/2 +- y with q = (k + 1) >> 1 in {0, 1, 2}: cos = {cos y, -+sin y, -cos y}
    const int q = (k + 1) >> 1;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sy = __builtin_fma(y * z, ps, y);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cy = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.));
    // q = 0: cos(y) (k = 0) ; q = 1: cos(pi/2 -+ y) = +-sin(y): k = 1 -> angle = pi/2 - y -> sin y,
    // k = 2 -> pi/2 + y -> -sin y ; q = 2 (k = 3): cos(pi - y) = -cos y
    const double c1 = k == 1 ? sy : -sy;
    return q == 0 ? cy : (q == 1 ? c1 : -cy);
}

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
// SWZ: the row is stored at row_swizzle(j) instead of j (an LDS row written four columns apart
// per lane, see sep_sample_eval).
__host__ __device__ inline int row_swizzle(int j) { return j ^ (((j >> 4) & 3) << 2); }

template<int G, bool SWZ = false, int UNR = 1>
__device__ inline double eval_row_group(int obj, int n, const double *xrow, const double *aux,
        int g)
{
    struct Row {
        const double *p;
        __device__ double operator[](int j) const { return p[SWZ ? row_swizzle(j) : j]; }
    };
    const Row x { xrow };
    // (UNR > 1 unrolls the strided loops: the LDS reads and products of UNR terms are in flight
    // together, the terms are still ADDED in the loop's order -- same bits, a fraction of the
    // latency.  For the kernels whose evaluation IS the latency chain (CCPSO's context
    // evaluations); elsewhere the extra registers cost occupancy)
    double a = 0., b = 0.;
    switch (obj) {
    case OBJ_SPHERE:
#pragma unroll UNR
        for (int j = g; j < n; j += G) a += x[j] * x[j];
        return group_sum<G>(a);
    case OBJ_ROSENBROCK:
#pragma unroll UNR
        for (int j = g; j + 1 < n; j += G) {
            const double xj = x[j];
            const double t = x[j + 1] - xj * xj;
            const double u = 1. - xj;
            a += 100. * (t * t) + u * u;
        }
        return group_sum<G>(a);
    case OBJ_RASTRIGIN:
#pragma unroll UNR
        for (int j = g; j < n; j += G) a += x[j] * x[j] - 10. * cos_2pi(x[j]);
        return 10. * n + group_sum<G>(a);
    case OBJ_ELLIPSOID:
#pragma unroll UNR
        for (int j = g; j < n; j += G) a += aux[j] * (x[j] * x[j]);
        return group_sum<G>(a);
    case OBJ_ACKLEY:
#pragma unroll UNR
        for (int j = g; j < n; j += G) {
            a += x[j] * x[j];
            b += cos_2pi(x[j]);
        }
        a = group_sum<G>(a);
        b = group_sum<G>(b);
        return -20. * exp(-0.2 * sqrt(a / n)) - exp(b / n) + 20. + EULER_E;
    case OBJ_GRIEWANK:
        b = 1.;
#pragma unroll UNR
        for (int j = g; j < n; j += G) {
            a += x[j] * x[j];
            b *= cos(x[j] * aux[j]);
        }
        a = group_sum<G>(a);
        b = group_prod<G>(b);
        return 1. + a / 4000. - b;
    case OBJ_CIGAR:
#pragma unroll UNR
        for (int j = g; j < n; j += G)
            if (j > 0) a += x[j] * x[j];
        return x[0] * x[0] + 1.0e6 * group_sum<G>(a);
    case OBJ_DISCUS:
#pragma unroll UNR
        for (int j = g; j < n; j += G)
            if (j > 0) a += x[j] * x[j];
        return 1.0e6 * (x[0] * x[0]) + group_sum<G>(a);
    case OBJ_DIFFPOW:
#pragma unroll UNR
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

// ---------------------------------------------------------------------------
// The same objectives on a tile that is still in MFMA accumulator layout: x[t][r] is
// column 16 t + (lane & 15) of row (lane >> 4) + 4 r.  The 16 lanes that share a row are
// one DPP row, so sums, products, neighbours and prefix sums stay on the cross-lane data
// path (no LDS round trip).  Every lane of a row returns the row's f in f[r].  Only the
// objectives without transcendental terms are offered here (frag_objective_ok): 32 inlined
// cos/pow bodies per lane do not fit the register budget next to the accumulators, those
// objectives go through eval_row_group from LDS instead.
// ---------------------------------------------------------------------------
template<int CTRL>
__device__ inline double row16_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ inline double row16_sum(double v)
{
    v += row16_dpp<0x128>(v);   // row_ror:8
    v += row16_dpp<0x124>(v);   // row_ror:4
    v += row16_dpp<0x122>(v);   // row_ror:2
    v += row16_dpp<0x121>(v);   // row_ror:1
    return v;
}

__device__ inline double row16_prod(double v)
{
    v *= row16_dpp<0x128>(v);
    v *= row16_dpp<0x124>(v);
    v *= row16_dpp<0x122>(v);
    v *= row16_dpp<0x121>(v);
    return v;
}

// lane i <- lane (i + 1) mod 16 of the same row
__device__ inline double row16_next(double v) { return row16_dpp<0x12F>(v); }   // row_ror:15

// inclusive prefix sum over the 16 lanes of a row (row_shr:1,2,4,8; lanes shifted in read 0)
__device__ inline double row16_scan(double v)
{
    v += row16_dpp<0x111>(v);
    v += row16_dpp<0x112>(v);
    v += row16_dpp<0x114>(v);
    v += row16_dpp<0x118>(v);
    return v;
}

__host__ __device__ inline bool frag_objective_ok(int obj)
{
    return obj == OBJ_SPHERE || obj == OBJ_ROSENBROCK || obj == OBJ_ELLIPSOID || obj == OBJ_CIGAR
            || obj == OBJ_DISCUS || obj == OBJ_SCHWEFEL12;
}

template<int NT>
__device__ inline void eval_frag_rows(int obj, int n, const double (&x)[NT][4], const double *aux,
        int lane, double (&f)[4])
{
    const int c0 = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double a = 0.;
        switch (obj) {
        case OBJ_SPHERE:
#pragma unroll
            for (int t = 0; t < NT; t++) a += (16 * t + c0 < n) ? x[t][r] * x[t][r] : 0.;
            f[r] = row16_sum(a);
            break;
        case OBJ_ROSENBROCK:
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const double xj = x[t][r];
                const double same = row16_next(xj);
                const double wrap = row16_next(x[t + 1 < NT ? t + 1 : t][r]);
                const double xn = c0 < 15 ? same : wrap;
                const double tt = xn - xj * xj;
                const double u = 1. - xj;
                a += (16 * t + c0 + 1 < n) ? 100. * (tt * tt) + u * u : 0.;
            }
            f[r] = row16_sum(a);
            break;
        case OBJ_ELLIPSOID:
#pragma unroll
            for (int t = 0; t < NT; t++)
                a += (16 * t + c0 < n) ? aux[16 * t + c0] * (x[t][r] * x[t][r]) : 0.;
            f[r] = row16_sum(a);
            break;
        case OBJ_CIGAR:
        case OBJ_DISCUS: {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int j = 16 * t + c0;
                a += (j > 0 && j < n) ? x[t][r] * x[t][r] : 0.;
            }
            a = row16_sum(a);
            const double x0 = row16_sum(c0 == 0 ? x[0][r] : 0.);
            f[r] = obj == OBJ_CIGAR ? x0 * x0 + 1.0e6 * a : 1.0e6 * (x0 * x0) + a;
            break;
        }
        case OBJ_SCHWEFEL12: {
            double carry = 0.;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const double v = (16 * t + c0 < n) ? x[t][r] : 0.;
                const double run = carry + row16_scan(v);
                a += (16 * t + c0 < n) ? run * run : 0.;
                carry += row16_sum(v);
            }
            f[r] = row16_sum(a);
            break;
        }
        default:
            f[r] = 0.;
        }
    }
}

} // namespace bbo
