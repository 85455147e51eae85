// bbo_rng.hpp -- device-resident counter-based generator (Philox4x32-10).
//
// The reference draws every random number from ONE process-global std::mt19937
// in strict program order (/root/reference/src/random.hpp:677-680, call sites
// listed in SURVEY.md section 8 row a17).  A sequential stream cannot feed a
// generation-parallel GPU kernel, so the HIP path keys every draw by WHAT it is
// for instead of WHEN it is drawn:
//     key     = 64-bit seed of the optimizer handle
//     counter = (row, column block, generation, stream << 24 | population)
// which makes every kernel launch order-independent and lets the CPU oracle
// (oracle/philox.h, the same arithmetic on the host) regenerate any draw.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bbo {

enum Stream : uint32_t {
    STREAM_CMA_NORMAL = 1,
    STREAM_INIT = 2,
    STREAM_DE_PARAM = 3,
    STREAM_DE_CROSS = 4,
    STREAM_PSO_R = 5,
    STREAM_PSO_CTRL = 6,
    STREAM_RESTART = 7,
    STREAM_DE_ARCH = 8
};

struct u32x4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline u32x4 philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1,
        uint32_t c2, uint32_t c3)
{
    uint32_t k0 = (uint32_t) seed, k1 = (uint32_t) (seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t) 0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t) 0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t) p1;
        const uint32_t n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4 { c0, c1, c2, c3 };
}

__host__ __device__ inline uint32_t stream_word(uint32_t stream, uint32_t sub)
{
    return (stream << 24) | (sub & 0x00FFFFFFu);
}

// 53 random bits -> [0,1)
__host__ __device__ inline double u01(uint32_t lo, uint32_t hi)
{
    const uint64_t b = (((uint64_t) hi << 32) | lo) >> 11;
    return (double) b * 0x1.0p-53;
}

// 53 random bits -> (0,1]
__host__ __device__ inline double u01_open0(uint32_t lo, uint32_t hi)
{
    const uint64_t b = (((uint64_t) hi << 32) | lo) >> 11;
    return (double) (b + 1) * 0x1.0p-53;
}

// uniform integer in [0, range), multiply-shift
__host__ __device__ inline int uint_below(uint32_t w, int range)
{
    return (int) (((uint64_t) w * (uint64_t) (uint32_t) range) >> 32);
}

// ln(u) for u in [2^-53, 1].  Only +, *, /, fma -- oracle/philox.h (bbo_log_unit) states the
// same arithmetic on the CPU and gets the same bits.  u = m 2^e, m in [sqrt(1/2), sqrt(2)),
// ln m = 2 atanh(s), s = (m - 1)/(m + 1), odd series to s^23.  About a third of ocml's log.
__device__ inline double log_unit(double u)
{
    const uint64_t b = (uint64_t) __double_as_longlong(u);
    int e = (int) (b >> 52) - 1023;
    double m = __longlong_as_double((long long) ((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
    const bool big = m > 0x1.6a09e667f3bcdp+0;
    m = big ? m * 0.5 : m;
    e += big ? 1 : 0;
    const double s = (m - 1.) / (m + 1.);
    const double z = s * s;
    double p = 1. / 23.;
    p = __builtin_fma(p, z, 1. / 21.);
    p = __builtin_fma(p, z, 1. / 19.);
    p = __builtin_fma(p, z, 1. / 17.);
    p = __builtin_fma(p, z, 1. / 15.);
    p = __builtin_fma(p, z, 1. / 13.);
    p = __builtin_fma(p, z, 1. / 11.);
    p = __builtin_fma(p, z, 1. / 9.);
    p = __builtin_fma(p, z, 1. / 7.);
    p = __builtin_fma(p, z, 1. / 5.);
    p = __builtin_fma(p, z, 1. / 3.);
    const double s2 = s + s;
    const double lm = __builtin_fma(s2 * z, p, s2);
    const double de = (double) e;
    return __builtin_fma(de, 0x1.62e42fee00000p-1, __builtin_fma(de, 0x1.a39ef35793c76p-33, lm));
}

// sin and cos of 2 pi t, t in [0, 1) a multiple of 2^-53: exact octant reduction, then the
// fdlibm kernel polynomials on [0, pi/4] (oracle twin: bbo_sincos_turn)
__device__ inline void sincos_turn(double t, double &sn, double &cs)
{
    const double v = t * 8.;
    const int k = (int) v;
    double f = v - (double) k;
    const bool odd = (k & 1) != 0;
    f = odd ? 1. - f : f;
    const double x = f * 0x1.921fb54442d18p-1;
    const double z = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sx = __builtin_fma(x * z, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cx = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.));
    const int q = ((k + 1) >> 1) & 3;
    const double sy = odd ? -sx : sx;
    const double s_even = (q & 2) ? -sy : sy, c_even = (q & 2) ? -cx : cx;   // q = 0, 2
    const double s_odd = (q & 2) ? -cx : cx, c_odd = (q & 2) ? sy : -sy;     // q = 1, 3
    sn = (q & 1) ? s_odd : s_even;
    cs = (q & 1) ? c_odd : c_even;
}

// one Philox call -> two standard normals (Box-Muller), bit-identical to bbo_normal_pair
// of oracle/philox.h
__device__ inline void normal_pair(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
        uint32_t c3, double &z0, double &z1)
{
    const u32x4 w = philox4x32_10(seed, c0, c1, c2, c3);
    const double u1 = u01_open0(w.x, w.y);
    const double u2 = u01(w.z, w.w);
    const double r = sqrt(-2. * log_unit(u1));
    double s, c;
    sincos_turn(u2, s, c);
    z0 = r * c;
    z1 = r * s;
}

// ---------------------------------------------------------------------------
// normal_quad: the samplers' generator -- ONE Philox call -> FOUR standard normals (two
// Box-Muller pairs from 32 + 32 bits each), about half the instructions of two normal_pair
// calls.  Per pair:
//   radius  r = sqrt(-2 ln u), u = (a + 1) 2^-32 in (0, 1]      (|z| <= 6.66)
//           -2 ln u from a 91-entry table in LDS (bbo_normal_table.inc; bin width 1/128 on
//           m in [0.70703125, 1.4140625), centre 1 exactly in bin 37) and a degree-8
//           polynomial in r = m inv_i - 1, |r| < 2^-7.5: no division
//   angle   x = (k + 1/2) (pi/4) 2^-29 in (0, pi/4) from 29 bits, fdlibm kernels, and three
//           more bits pick one of the 8 symmetries of the square (swap, -sin, -cos): a
//           uniform direction without the octant bookkeeping
// Only +, *, fma, frexp/ldexp, sqrt: oracle/philox.h (bbo_normal_quad) gets the same bits.
// ---------------------------------------------------------------------------
static __device__ const double NORMAL_TABLE[91][2] = {
#include "bbo_normal_table.inc"
};
constexpr int NORMAL_TABLE_N = 91;

// cooperative copy of the table into LDS (caller synchronises)
__device__ inline void normal_table_fill(double2 *tab, int tid, int nthreads)
{
    for (int i = tid; i < NORMAL_TABLE_N; i += nthreads)
        tab[i] = make_double2(NORMAL_TABLE[i][0], NORMAL_TABLE[i][1]);
}

// -2 ln((a + 1) 2^-32)
__device__ inline double neg2log32(uint32_t a, const double2 *tab)
{
    const double d = (double) a + 1.;                       // 1 .. 2^32, exact
    double m = __builtin_amdgcn_frexp_mant(d);              // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(d);                  // d = m 2^e
    const int s = m < 0.70703125 ? 1 : 0;
    m = __builtin_amdgcn_ldexp(m, s);                       // [0.70703125, 1.4140625)
    e -= s;
    const int i = (int) __builtin_fma(m, 128., -90.5);      // exact; floor
    const double2 ent = tab[i];
    const double r = __builtin_fma(m, ent.x, -1.);
    double p = 2. / 8.;
    p = __builtin_fma(p, r, -2. / 7.);
    p = __builtin_fma(p, r, 2. / 6.);
    p = __builtin_fma(p, r, -2. / 5.);
    p = __builtin_fma(p, r, 2. / 4.);
    p = __builtin_fma(p, r, -2. / 3.);
    p = __builtin_fma(p, r, 1.);
    p = __builtin_fma(p, r, -2.);
    const double base = __builtin_fma((double) (32 - e), 0x1.62e42fefa39efp+0, ent.y);
    return __builtin_fma(p, r, base);
}

// (sin, cos) of a uniform direction from 32 bits
__device__ inline void sincos_oct(uint32_t b, double &sn, double &cs)
{
    const double x = __builtin_fma((double) (b >> 3), 0x1.921fb54442d18p-30, 0x1.921fb54442d18p-31);
    const double z = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double sx = __builtin_fma(x * z, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double cx = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.));
    const bool sw = (b & 1u) != 0;
    const double s0 = sw ? cx : sx, c0 = sw ? sx : cx;
    sn = __longlong_as_double(__double_as_longlong(s0) ^ ((long long) (b & 2u) << 62));
    cs = __longlong_as_double(__double_as_longlong(c0) ^ ((long long) (b & 4u) << 61));
}

__device__ inline void normal_quad(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
        uint32_t c3, const double2 *tab, double &z0, double &z1, double &z2, double &z3)
{
    const u32x4 w = philox4x32_10(seed, c0, c1, c2, c3);
    double s, c;
    const double ra = sqrt(neg2log32(w.x, tab));
    sincos_oct(w.y, s, c);
    z0 = ra * c;
    z1 = ra * s;
    const double rb = sqrt(neg2log32(w.z, tab));
    sincos_oct(w.w, s, c);
    z2 = rb * c;
    z3 = rb * s;
}

// CMA-ES sampling: Philox call q of a candidate fills columns 16 (q >> 2) + (q & 3) + 4 i,
// i = 0..3, i.e. exactly the four k-steps lane group (q & 3) feeds to the MFMA A operand
__host__ __device__ inline int cma_quad_col0(int q) { return 16 * (q >> 2) + (q & 3); }

// Keyed bijection of [0, np): 4-round Feistel network on 2 kb bits (kb = half the bits of
// np - 1, rounded up), one Philox word per round keyed by (half-word, 8 + round, generation),
// cycle-walked back into range.  The device's std::shuffle (CSO slots, CCPSO coordinates);
// oracle twins: Cso::feistel_perm, Ccpso::feistel_perm.
__device__ inline uint32_t cso_perm(uint32_t s0, int kb, uint32_t np, uint64_t seed, uint32_t gen,
        uint32_t sw)
{
    const uint32_t mask = (1u << kb) - 1u;
    uint32_t x = s0;
    do {
        uint32_t L = x >> kb, R = x & mask;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const u32x4 w = philox4x32_10(seed, R, (uint32_t) (8 + r), gen, sw);
            const uint32_t t = L ^ (w.x & mask);
            L = R;
            R = t;
        }
        x = (L << kb) | R;
    } while (x >= np);
    return x;
}

} // namespace bbo
