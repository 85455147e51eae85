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
// normal_quad: the CMA samplers' generator -- ONE Philox call -> FOUR standard normals by the
// Marsaglia-Tsang ZIGGURAT (1024 strips of exp(-x^2 / 2); tables: scripts/gen_ziggurat_table.py).
// On gfx950 the fp64 matrix instruction shares the one vector pipe with everything else (no
// overlap), so a sampler is paid per vector instruction: a Box-Muller pair
// costs ~70 of them beyond Philox (logarithm, root, sine and cosine), a ziggurat normal 8.
// A 32-bit word w is one draw: strip i = w & 1023, t = (w >> 10) | 1 (an odd 22-bit integer: the
// position inside the strip, never 0), sign = bit 10:
//     z = +/- t W[i];   t < K[i]: the point lies under the curve for sure -- 99.57 % of the draws
// The rest (zig_slow) takes fresh Philox words at counters no first draw uses (c1 = q | (slot + 1)
// << 12 | attempt << 16): in the base strip the tail beyond r = 4.04 by Marsaglia's -ln(u)/r
// method, in the others the wedge test f(x_i) + U (f(x_{i+1}) - f(x_i)) < exp(-x^2 / 2) and, on
// rejection, a fresh draw.  Exact normal, unbounded tails (the Box-Muller pair of round 1 stopped
// at 6.66 sigma), 32 bits of entropy per normal as before.  Only integer operations, +, *, fma,
// ldexp and conversions: oracle/philox.h (bbo_normal_quad) states the same arithmetic on the CPU
// and gets the same bits, slow paths included.
// ---------------------------------------------------------------------------
#include "bbo_zig_table.inc"
struct alignas(16) ZigStrip { double w, kbits; };   // W[i]; K[i] in the low word of kbits
static __device__ const ZigStrip ZIG_WK[BBO_ZIG_N] = BBO_ZIG_TABLE_WK;
static __device__ const double ZIG_F[BBO_ZIG_N + 1] = BBO_ZIG_TABLE_F;
constexpr int NORMAL_TABLE_N = BBO_ZIG_N;           // LDS copy of ZIG_WK (16 KB)

// cooperative copy of the fast-path table into LDS (caller synchronises)
__device__ inline void normal_table_fill(double2 *tab, int tid, int nthreads)
{
    const double2 *src = reinterpret_cast<const double2*>(ZIG_WK);
    for (int i = tid; i < NORMAL_TABLE_N; i += nthreads) tab[i] = src[i];
}
// the global copies, as the pointers zig_slow / normal_quad_settle take when the caller has
// no LDS copy to offer
__device__ inline const double2 *zig_global_wk() { return reinterpret_cast<const double2*>(ZIG_WK); }
__device__ inline const double *zig_global_f() { return ZIG_F; }
constexpr int NORMAL_FTABLE_N = BBO_ZIG_N + 1;      // optional LDS copy of ZIG_F (8 KB)
__device__ inline void normal_ftable_fill(double *ftab, int tid, int nthreads)
{
    for (int i = tid; i < NORMAL_FTABLE_N; i += nthreads) ftab[i] = ZIG_F[i];
}

// exp(-s) for s in [0, 700]: s = k ln 2 + r, |r| <= 0.35, Taylor to the 13th power (oracle twin:
// bbo_exp_neg)
__device__ inline double exp_neg(double s)
{
    const int k = (int) __builtin_fma(s, 0x1.71547652b82fep+0, 0.5);
    const double dk = (double) k;
    double r = __builtin_fma(-dk, 0x1.62e42fefa3800p-1, s);
    r = __builtin_fma(-dk, 0x1.ef35793c76730p-45, r);
    const double y = -r;
    double p = 0x1.6124613a86d09p-33;               // 1/13!
    p = __builtin_fma(p, y, 0x1.1eed8eff8d898p-29); // 1/12!
    p = __builtin_fma(p, y, 0x1.ae64567f544e4p-26); // 1/11!
    p = __builtin_fma(p, y, 0x1.27e4fb7789f5cp-22); // 1/10!
    p = __builtin_fma(p, y, 0x1.71de3a556c734p-19); // 1/9!
    p = __builtin_fma(p, y, 0x1.a01a01a01a01ap-16); // 1/8!
    p = __builtin_fma(p, y, 0x1.a01a01a01a01ap-13); // 1/7!
    p = __builtin_fma(p, y, 0x1.6c16c16c16c17p-10); // 1/6!
    p = __builtin_fma(p, y, 0x1.1111111111111p-7);  // 1/5!
    p = __builtin_fma(p, y, 0x1.5555555555555p-5);  // 1/4!
    p = __builtin_fma(p, y, 0x1.5555555555555p-3);  // 1/3!
    p = __builtin_fma(p, y, 0.5);
    p = __builtin_fma(p, y, 1.);
    p = __builtin_fma(p, y, 1.);
    return __builtin_amdgcn_ldexp(p, -k);
}

// the 0.43 % of the draws the fast test does not settle (see above); slot = which of the four
// words of Philox call c1 this draw was.  wk / f: the tables, LDS or global copies (a settle round
// is a chain of dependent look-ups: from LDS it is latency the other wavefront of the SIMD does
// not have to cover)
__device__ inline double zig_slow(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t slot,
        uint32_t c2, uint32_t c3, uint32_t idx, uint32_t t, uint32_t sign, const double2 *wk,
        const double *f)
{
    for (uint32_t attempt = 0;; attempt++) {
        const u32x4 w = philox4x32_10(seed, c0, c1 | ((slot + 1u) << 12) | (attempt << 16), c2, c3);
        if (idx == 0) {
            const double xx = -log_unit(u01_open0(w.x, w.y)) * BBO_ZIG_INV_R;
            const double yy = -log_unit(u01_open0(w.z, w.w));
            if (yy + yy > xx * xx) {
                const double v = BBO_ZIG_R + xx;
                return sign ? -v : v;
            }
        } else {
            const double x = (double) t * wk[idx].x;
            const double f0 = f[idx], f1 = f[idx + 1];
            const double y = __builtin_fma(u01(w.x, w.y), f1 - f0, f0);
            if (y < exp_neg(0.5 * (x * x))) return sign ? -x : x;
            idx = w.z & 1023u;
            t = (w.z >> 10) | 1u;
            sign = (w.z >> 10) & 1u;
            const double2 e = wk[idx];
            if (t < (uint32_t) __double_as_longlong(e.y)) {
                const double x2 = (double) t * e.x;
                return sign ? -x2 : x2;
            }
        }
    }
}

// the fast path of one word: the candidate +/- t W[i] and whether it stands
__device__ inline double zig_candidate(uint32_t w, const double2 e, bool &settled)
{
    // e = tab[w & 1023]: the caller reads the strips of all its words first (four LDS reads in
    // flight instead of read, wait, use four times in a row)
    const uint32_t t = (w >> 10) | 1u;
    const double x = (double) t * e.x;
    settled = t < (uint32_t) __double_as_longlong(e.y);
    // (sign = bit 10 of the word -> bit 63 of the double)
    return __longlong_as_double(__double_as_longlong(x) ^ ((long long) (w & 0x400u) << 53));
}

// A slow path taken by ONE lane is paid by the whole wavefront, and with 256 draws per wavefront
// and call SOME lane takes it two times out of three.  The samplers therefore draw in two steps:
// normal_quad_fast for every call they hold (candidates + a 4-bit mask of the unsettled ones,
// no branch), then normal_quad_settle once per unsettled draw -- over the 16-32 calls a lane
// holds that is one or two rounds per wavefront instead of one per call.
__device__ inline uint32_t normal_quad_fast(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
        uint32_t c3, const double2 *tab, double &z0, double &z1, double &z2, double &z3)
{
    const u32x4 w = philox4x32_10(seed, c0, c1, c2, c3);
    bool s0, s1, s2, s3;
    const double2 e0 = tab[w.x & 1023u], e1 = tab[w.y & 1023u], e2 = tab[w.z & 1023u],
            e3 = tab[w.w & 1023u];
    z0 = zig_candidate(w.x, e0, s0);
    z1 = zig_candidate(w.y, e1, s1);
    z2 = zig_candidate(w.z, e2, s2);
    z3 = zig_candidate(w.w, e3, s3);
    return (s0 ? 0u : 1u) | (s1 ? 0u : 2u) | (s2 ? 0u : 4u) | (s3 ? 0u : 8u);
}

// NQ calls in a row (call q at counter c1 = c1_0 + q c1_step), software-pipelined by hand: the
// four strip reads of call q are issued, THEN the Philox rounds of call q + 1 run (some 60 vector
// instructions: longer than an LDS round trip), then call q's candidates are formed.  Written as
// NQ calls of normal_quad_fast the compiler emits `ds_read, s_waitcnt lgkmcnt(0), use` per word:
// 32 exposed LDS round trips per 16-row tile of the n = 128 sampler.  Same draws, same bits.
template<int NQ>
__device__ __forceinline__ uint32_t normal_quads_fast(uint64_t seed, uint32_t c0, uint32_t c1_0,
        uint32_t c1_step, uint32_t c2, uint32_t c3, const double2 *tab, double (&z)[4 * NQ])
{
    uint32_t pend = 0;
    u32x4 wn = philox4x32_10(seed, c0, c1_0, c2, c3);
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const u32x4 w = wn;
#ifdef BBO_DIAG_STRIP_NOCONFLICT
        // timing / counter diagnostic only (wrong normals): the 16 lanes of a b128 read group take
        // 16 consecutive records, so the strip reads cannot conflict (HISTORY.md section 6, round 4)
        const uint32_t dl = threadIdx.x & 15u;
        const double2 e0 = tab[((w.x & 1023u) & ~15u) | dl], e1 = tab[((w.y & 1023u) & ~15u) | dl],
                e2 = tab[((w.z & 1023u) & ~15u) | dl], e3 = tab[((w.w & 1023u) & ~15u) | dl];
#else
        const double2 e0 = tab[w.x & 1023u], e1 = tab[w.y & 1023u], e2 = tab[w.z & 1023u],
                e3 = tab[w.w & 1023u];
#endif
        __builtin_amdgcn_sched_barrier(0);       // the reads go out before the next call's rounds
        if (q + 1 < NQ) wn = philox4x32_10(seed, c0, c1_0 + (uint32_t) (q + 1) * c1_step, c2, c3);
        __builtin_amdgcn_sched_barrier(0);
        bool s0, s1, s2, s3;
        z[4 * q] = zig_candidate(w.x, e0, s0);
        z[4 * q + 1] = zig_candidate(w.y, e1, s1);
        z[4 * q + 2] = zig_candidate(w.z, e2, s2);
        z[4 * q + 3] = zig_candidate(w.w, e3, s3);
        pend |= ((s0 ? 0u : 1u) | (s1 ? 0u : 2u) | (s2 ? 0u : 4u) | (s3 ? 0u : 8u)) << (4 * q);
    }
    return pend;
}

// draw `slot` of call c1, which normal_quad_fast reported unsettled (the call is recomputed)
__device__ inline double normal_quad_settle(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t slot,
        uint32_t c2, uint32_t c3, const double2 *wk, const double *f)
{
    const u32x4 w4 = philox4x32_10(seed, c0, c1, c2, c3);
    const uint32_t w = slot == 0 ? w4.x : slot == 1 ? w4.y : slot == 2 ? w4.z : w4.w;
    return zig_slow(seed, c0, c1, slot, c2, c3, w & 1023u, (w >> 10) | 1u, (w >> 10) & 1u, wk, f);
}

// both steps at once, for the places that draw one call at a time
__device__ inline void normal_quad(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
        uint32_t c3, const double2 *tab, double &z0, double &z1, double &z2, double &z3)
{
    const u32x4 w = philox4x32_10(seed, c0, c1, c2, c3);
    bool s0, s1, s2, s3;
    const double2 e0 = tab[w.x & 1023u], e1 = tab[w.y & 1023u], e2 = tab[w.z & 1023u],
            e3 = tab[w.w & 1023u];
    z0 = zig_candidate(w.x, e0, s0);
    z1 = zig_candidate(w.y, e1, s1);
    z2 = zig_candidate(w.z, e2, s2);
    z3 = zig_candidate(w.w, e3, s3);
    if (!(s0 && s1 && s2 && s3)) {
        if (!s0) z0 = zig_slow(seed, c0, c1, 0, c2, c3, w.x & 1023u, (w.x >> 10) | 1u, (w.x >> 10) & 1u, tab, zig_global_f());
        if (!s1) z1 = zig_slow(seed, c0, c1, 1, c2, c3, w.y & 1023u, (w.y >> 10) | 1u, (w.y >> 10) & 1u, tab, zig_global_f());
        if (!s2) z2 = zig_slow(seed, c0, c1, 2, c2, c3, w.z & 1023u, (w.z >> 10) | 1u, (w.z >> 10) & 1u, tab, zig_global_f());
        if (!s3) z3 = zig_slow(seed, c0, c1, 3, c2, c3, w.w & 1023u, (w.w >> 10) | 1u, (w.w >> 10) & 1u, tab, zig_global_f());
    }
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
