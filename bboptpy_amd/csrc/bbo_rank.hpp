// bbo_rank.hpp -- rank-by-counting, the device replacement of the reference's std::sort of
// the fitness array (base_cmaes.cpp:221, shade.cpp:215, jade.cpp:101):
//   rank[i] = #{ j : f_j < f_i  or  (f_j == f_i and j < i) }      (stable in the index)
// n^2 compares, but embarrassingly parallel and branch-free: a workgroup of 256 threads ranks
// 32 candidates, 8 lanes per candidate each scanning one eighth of the fitness array, which is
// staged through LDS in tiles so the inner loop is LDS-broadcast reads + 3 VALU ops.
#pragma once

#include <hip/hip_runtime.h>

namespace bbo {

constexpr int RANK_TILE = 2048;   // doubles per LDS tile (16 KiB)

// f: fitness of `count` candidates; cand: this thread's candidate (may be >= count);
// slice: 0..7; tile: __shared__ double[RANK_TILE].  Returns the rank in every lane of the
// 8-lane group (valid when cand < count).  All 256 threads of the workgroup must call it.
__device__ inline int rank_by_counting(const double *f, int count, int cand, int slice,
        double *tile)
{
    const int tid = threadIdx.x;
    const double fi = cand < count ? f[cand] : __builtin_huge_val();
    int cnt = 0;
    for (int base = 0; base < count; base += RANK_TILE) {
        const int len = min(RANK_TILE, count - base);
        __syncthreads();
        for (int q = tid; q < RANK_TILE; q += 256)
            tile[q] = q < len ? f[base + q] : __builtin_huge_val();
        __syncthreads();
        // lane `slice` reads the pairs (2 slice, 2 slice + 1) + 16 t: 8 lanes cover 16
        // consecutive doubles, the 8 candidate groups of a wavefront read the same addresses
        const int lenp = (len + 15) & ~15;
        for (int q = 2 * slice; q < lenp; q += 16) {
            const double2 v = *reinterpret_cast<const double2*>(&tile[q]);
            const int j = base + q;
            cnt += (v.x < fi) || (v.x == fi && j < cand);
            cnt += (v.y < fi) || (v.y == fi && j + 1 < cand);
        }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 8);
    return cnt;
}

} // namespace bbo

namespace bbo {

// Whole-array bitonic sort of (fitness, index) pairs in LDS by ONE workgroup of 1024 threads:
// count <= SORT_LDS_MAX candidates, padded to a power of two with (+inf, INT_MAX).  The index
// is the secondary key, so the order is total and identical to rank_by_counting's.
// order[r] = candidate of rank r, rank[i] = rank of candidate i.
constexpr int SORT_LDS_MAX = 8192;

__device__ inline bool pair_less(double fa, int ia, double fb, int ib)
{
    return fa < fb || (fa == fb && ia < ib);
}

// keys/idx: LDS arrays of `m` (power of two >= count) entries
__device__ inline void bitonic_sort_lds(const double *f, int count, int m, double *keys,
        int *idx, int *order, int *rank)
{
    const int tid = threadIdx.x, T = blockDim.x;
    for (int q = tid; q < m; q += T) {
        keys[q] = q < count ? f[q] : __builtin_huge_val();
        idx[q] = q < count ? q : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int q = tid; q < (m >> 1); q += T) {
                // q-th compare-exchange of this stage: partner indices differ in bit j
                const int lo = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const int hi = lo | j;
                const bool up = (lo & k) == 0;
                const double fa = keys[lo], fb = keys[hi];
                const int ia = idx[lo], ib = idx[hi];
                if (pair_less(fb, ib, fa, ia) == up) {
                    keys[lo] = fb;
                    keys[hi] = fa;
                    idx[lo] = ib;
                    idx[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    for (int q = tid; q < count; q += T) {
        const int cand = idx[q];
        order[q] = cand;
        rank[cand] = q;
    }
}

} // namespace bbo
