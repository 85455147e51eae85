/*
 * oracle/philox.h -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * CPU statement of the counter-based generator the HIP kernels use
 * (bboptpy_amd/csrc/bbo_rng.hpp holds the device twin): Philox4x32-10
 * (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
 * Random123 known-answer vectors are checked in tests/test_rng.py) plus the
 * bit -> double conversions and the Box-Muller transform.  The reference has
 * no counterpart: it consumes one process-global std::mt19937
 * (/root/reference/src/random.hpp:677-680), which a device generator cannot
 * reproduce (hard part 2) -- the oracle therefore carries
 * BOTH generators (mt19937 replay in bbo_oracle.cpp, this one here).
 *
 * Counter layout (shared with the device code):
 *   c0 = row (candidate / individual / particle index)
 *   c1 = column block (pair index for normals, word index for uniforms)
 *   c2 = generation (or draw sequence number)
 *   c3 = (stream << 24) | sub-stream
 * key  = 64-bit seed of the optimizer handle.
 */
#ifndef BBO_ORACLE_PHILOX_H_
#define BBO_ORACLE_PHILOX_H_

#include <math.h>
#include <stdint.h>

enum {
    BBO_STREAM_CMA_NORMAL = 1,   /* CMA-ES sampling normals             */
    BBO_STREAM_INIT = 2,         /* uniform initial populations         */
    BBO_STREAM_DE_PARAM = 3,     /* DE per-individual parameter draws   */
    BBO_STREAM_DE_CROSS = 4,     /* DE per-coordinate crossover draws   */
    BBO_STREAM_PSO_R = 5,        /* PSO r1/r2 per coordinate            */
    BBO_STREAM_PSO_CTRL = 6,     /* APSO delta1/delta2/elitist draws    */
    BBO_STREAM_RESTART = 7,      /* IPOP/BIPOP restart points, u, u'    */
    BBO_STREAM_DE_ARCH = 8       /* DE archive slot draws               */
};

static inline void bbo_philox4x32_10(const uint32_t key_in[2],
        const uint32_t ctr_in[4], uint32_t out[4])
{
    uint32_t k0 = key_in[0], k1 = key_in[1];
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
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
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline void bbo_philox(uint64_t seed, uint32_t c0, uint32_t c1,
        uint32_t c2, uint32_t c3, uint32_t out[4])
{
    const uint32_t key[2] = { (uint32_t) seed, (uint32_t) (seed >> 32) };
    const uint32_t ctr[4] = { c0, c1, c2, c3 };
    bbo_philox4x32_10(key, ctr, out);
}

static inline uint32_t bbo_stream(int stream, uint32_t sub)
{
    return ((uint32_t) stream << 24) | (sub & 0x00FFFFFFu);
}

/* 53 random bits -> [0,1) */
static inline double bbo_u01(uint32_t lo, uint32_t hi)
{
    const uint64_t b = (((uint64_t) hi << 32) | lo) >> 11;
    return (double) b * 0x1.0p-53;
}

/* 53 random bits -> (0,1] (safe for log) */
static inline double bbo_u01_open0(uint32_t lo, uint32_t hi)
{
    const uint64_t b = (((uint64_t) hi << 32) | lo) >> 11;
    return (double) (b + 1) * 0x1.0p-53;
}

/* uniform integer in [0, range): multiply-shift (bias <= range / 2^32) */
static inline int bbo_uint_below(uint32_t w, int range)
{
    return (int) (((uint64_t) w * (uint64_t) (uint32_t) range) >> 32);
}

/*
 * ln(u) for u in [2^-53, 1].  Written out in +, *, /, fma so that the device twin
 * (bbo_rng.hpp: log_unit) produces the same bits: u = m 2^e with m in [sqrt(1/2), sqrt(2)),
 * ln m = 2 atanh(s), s = (m - 1)/(m + 1), odd series to s^23 (|s| <= 0.1716).
 */
static inline double bbo_log_unit(double u)
{
    union { double d; uint64_t b; } v;
    v.d = u;
    int e = (int) (v.b >> 52) - 1023;
    v.b = (v.b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull;
    double m = v.d;
    if (m > 0x1.6a09e667f3bcdp+0) {
        m *= 0.5;
        e += 1;
    }
    const double s = (m - 1.) / (m + 1.);
    const double z = s * s;
    double p = 1. / 23.;
    p = fma(p, z, 1. / 21.);
    p = fma(p, z, 1. / 19.);
    p = fma(p, z, 1. / 17.);
    p = fma(p, z, 1. / 15.);
    p = fma(p, z, 1. / 13.);
    p = fma(p, z, 1. / 11.);
    p = fma(p, z, 1. / 9.);
    p = fma(p, z, 1. / 7.);
    p = fma(p, z, 1. / 5.);
    p = fma(p, z, 1. / 3.);
    const double s2 = s + s;
    const double lm = fma(s2 * z, p, s2);
    const double de = (double) e;
    return fma(de, 0x1.62e42fee00000p-1, fma(de, 0x1.a39ef35793c76p-33, lm));
}

/*
 * sin and cos of 2 pi t for t in [0, 1) a multiple of 2^-53: exact reduction to an octant,
 * then the fdlibm kernel polynomials on [0, pi/4] (device twin: sincos_turn).
 */
static inline void bbo_sincos_turn(double t, double *sn, double *cs)
{
    const double v = t * 8.;
    const int k = (int) v;
    double f = v - (double) k;
    if (k & 1) f = 1. - f;
    const double x = f * 0x1.921fb54442d18p-1;
    const double z = x * x;
    double ps = 1.58969099521155010221e-10;
    ps = fma(ps, z, -2.50507602534068634195e-08);
    ps = fma(ps, z, 2.75573137070700676789e-06);
    ps = fma(ps, z, -1.98412698298579493134e-04);
    ps = fma(ps, z, 8.33333333332248946124e-03);
    ps = fma(ps, z, -1.66666666666666324348e-01);
    const double sx = fma(x * z, ps, x);
    double pc = -1.13596475577881948265e-11;
    pc = fma(pc, z, 2.08757232129817482790e-09);
    pc = fma(pc, z, -2.75573143513906633035e-07);
    pc = fma(pc, z, 2.48015872894767294178e-05);
    pc = fma(pc, z, -1.38888888888741095749e-03);
    pc = fma(pc, z, 4.16666666666666019037e-02);
    const double cx = fma(z * z, pc, fma(z, -0.5, 1.));
    /* angle = q pi/2 + y, y = +x (even octant) or -x (odd octant) */
    const int q = ((k + 1) >> 1) & 3;
    const double sy = (k & 1) ? -sx : sx;
    switch (q) {
    case 0: *sn = sy; *cs = cx; break;
    case 1: *sn = cx; *cs = -sy; break;
    case 2: *sn = -sy; *cs = -cx; break;
    default: *sn = -cx; *cs = sy; break;
    }
}

/* one Philox call -> two standard normals (Box-Muller) */
static inline void bbo_normal_pair(uint64_t seed, uint32_t c0, uint32_t c1,
        uint32_t c2, uint32_t c3, double *z0, double *z1)
{
    uint32_t w[4];
    bbo_philox(seed, c0, c1, c2, c3, w);
    const double u1 = bbo_u01_open0(w[0], w[1]);
    const double u2 = bbo_u01(w[2], w[3]);
    const double r = sqrt(-2. * bbo_log_unit(u1));
    double sn, cs;
    bbo_sincos_turn(u2, &sn, &cs);
    *z0 = r * cs;
    *z1 = r * sn;
}

/*
 * The samplers' generator (device twin: normal_quad in bbo_rng.hpp): ONE Philox call -> FOUR
 * standard normals by the Marsaglia-Tsang ziggurat on 1024 strips of exp(-x^2 / 2)
 * (zig_table.inc, generated by scripts/gen_ziggurat_table.py).  A 32-bit word is one draw:
 *   strip i = w & 1023,  t = (w >> 10) | 1 (odd, 22 bits),  sign = bit 10,  z = +/- t W[i];
 *   t < K[i]: the point lies under the curve for sure (99.57 % of the draws);
 *   the rest (bbo_zig_slow) takes fresh Philox words at counters no first draw uses
 *   (c1 | (slot + 1) << 12 | attempt << 16): the tail beyond r by Marsaglia's -ln(u)/r method in
 *   the base strip, the wedge test in the others, a fresh strip after a rejection.
 * Only integer operations, +, *, fma, ldexp and conversions, in the device's order.
 */
#include "zig_table.inc"
static const double bbo_zig_w[BBO_ZIG_N] = BBO_ZIG_TABLE_W;
static const uint32_t bbo_zig_k[BBO_ZIG_N] = BBO_ZIG_TABLE_K;
static const double bbo_zig_f[BBO_ZIG_N + 1] = BBO_ZIG_TABLE_F;

/* exp(-s), s in [0, 700]: s = k ln 2 + r, |r| <= 0.35, Taylor to the 13th power
 * (device twin: exp_neg) */
static inline double bbo_exp_neg(double s)
{
    const int k = (int) fma(s, 0x1.71547652b82fep+0, 0.5);
    const double dk = (double) k;
    double r = fma(-dk, 0x1.62e42fefa3800p-1, s);
    r = fma(-dk, 0x1.ef35793c76730p-45, r);
    const double y = -r;
    double p = 0x1.6124613a86d09p-33;
    p = fma(p, y, 0x1.1eed8eff8d898p-29);
    p = fma(p, y, 0x1.ae64567f544e4p-26);
    p = fma(p, y, 0x1.27e4fb7789f5cp-22);
    p = fma(p, y, 0x1.71de3a556c734p-19);
    p = fma(p, y, 0x1.a01a01a01a01ap-16);
    p = fma(p, y, 0x1.a01a01a01a01ap-13);
    p = fma(p, y, 0x1.6c16c16c16c17p-10);
    p = fma(p, y, 0x1.1111111111111p-7);
    p = fma(p, y, 0x1.5555555555555p-5);
    p = fma(p, y, 0x1.5555555555555p-3);
    p = fma(p, y, 0.5);
    p = fma(p, y, 1.);
    p = fma(p, y, 1.);
    return ldexp(p, -k);
}

static inline double bbo_zig_slow(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t slot,
        uint32_t c2, uint32_t c3, uint32_t idx, uint32_t t, uint32_t sign)
{
    for (uint32_t attempt = 0;; attempt++) {
        uint32_t w[4];
        bbo_philox(seed, c0, c1 | ((slot + 1u) << 12) | (attempt << 16), c2, c3, w);
        if (idx == 0) {
            const double xx = -bbo_log_unit(bbo_u01_open0(w[0], w[1])) * BBO_ZIG_INV_R;
            const double yy = -bbo_log_unit(bbo_u01_open0(w[2], w[3]));
            if (yy + yy > xx * xx) {
                const double v = BBO_ZIG_R + xx;
                return sign ? -v : v;
            }
        } else {
            const double x = (double) t * bbo_zig_w[idx];
            const double f0 = bbo_zig_f[idx], f1 = bbo_zig_f[idx + 1];
            const double y = fma(bbo_u01(w[0], w[1]), f1 - f0, f0);
            if (y < bbo_exp_neg(0.5 * (x * x))) return sign ? -x : x;
            idx = w[2] & 1023u;
            t = (w[2] >> 10) | 1u;
            sign = (w[2] >> 10) & 1u;
            if (t < bbo_zig_k[idx]) {
                const double x2 = (double) t * bbo_zig_w[idx];
                return sign ? -x2 : x2;
            }
        }
    }
}

static inline void bbo_normal_quad(uint64_t seed, uint32_t c0, uint32_t c1,
        uint32_t c2, uint32_t c3, double z[4])
{
    uint32_t w[4];
    bbo_philox(seed, c0, c1, c2, c3, w);
    for (uint32_t s = 0; s < 4; s++) {
        const uint32_t idx = w[s] & 1023u, t = (w[s] >> 10) | 1u, sign = (w[s] >> 10) & 1u;
        if (t < bbo_zig_k[idx]) {
            const double x = (double) t * bbo_zig_w[idx];
            z[s] = sign ? -x : x;
        } else {
            z[s] = bbo_zig_slow(seed, c0, c1, s, c2, c3, idx, t, sign);
        }
    }
}

/*
 * CMA-ES sampling: Philox call q of a candidate fills columns 16 (q >> 2) + (q & 3) + 4 i,
 * i = 0..3, with its four normals.  The layout follows the MFMA A fragment of the device
 * kernel (lane k-group q & 3 feeds k-steps 4 (q >> 2) .. + 3), so a lane draws exactly the
 * normals it multiplies.
 */
static inline int bbo_cma_quad_of_column(int j) { return ((j >> 4) << 2) | (j & 3); }
static inline int bbo_cma_slot_of_column(int j) { return (j >> 2) & 3; }

#endif /* BBO_ORACLE_PHILOX_H_ */
