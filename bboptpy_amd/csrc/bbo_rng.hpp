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

// one Philox call -> two standard normals (Box-Muller)
__device__ inline void normal_pair(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2,
        uint32_t c3, double &z0, double &z1)
{
    const u32x4 w = philox4x32_10(seed, c0, c1, c2, c3);
    const double u1 = u01_open0(w.x, w.y);
    const double u2 = u01(w.z, w.w);
    const double r = sqrt(-2. * log(u1));
    const double a = 6.283185307179586476925286766559 * u2;
    double s, c;
    sincos(a, &s, &c);
    z0 = r * c;
    z1 = r * s;
}

} // namespace bbo
