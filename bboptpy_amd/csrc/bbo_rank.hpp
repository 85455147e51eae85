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
// slice: 0..SL-1; tile: __shared__ double[RANK_TILE].  Returns the rank in every lane of the
// SL-lane group (valid when cand < count).  All 256 threads of the workgroup must call it.
// SL = 8: 32 candidates per workgroup; SL = 32 (one population at a time, round 5: the compares are
// 64-bit and one wavefront per SIMD issues them -- 512 per thread took 29 us at lambda = 4096 while
// most of the chip idled): 8 candidates per workgroup, a quarter of the compares per thread.
template<int SL = 8>
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
        // lane `slice` reads the pairs (2 slice, 2 slice + 1) + 2 SL t: the SL lanes of a group cover
        // 2 SL consecutive doubles, the candidate groups of a wavefront read the same addresses
        const int lenp = (len + 2 * SL - 1) & ~(2 * SL - 1);
        for (int q = 2 * slice; q < lenp; q += 2 * SL) {
            const double2 v = *reinterpret_cast<const double2*>(&tile[q]);
            const int j = base + q;
            cnt += (v.x < fi) || (v.x == fi && j < cand);
            cnt += (v.y < fi) || (v.y == fi && j + 1 < cand);
        }
    }
#pragma unroll
    for (int off = SL / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, SL);
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

// lane ^ LX exchange of a (key, index) pair inside a wavefront.  LX = 1, 2, 4, 8 stay inside a
// 16-lane row and go through DPP moves (VALU, no LDS crossbar: quad_perm for 1 and 2,
// row_half_mirror + quad reversal for 4, row_ror:8 for 8); 16 and 32 use the LDS permute.
template<int CTRL>
__device__ inline int sort_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}

template<int LX>
__device__ inline int sort_xor_lane(int v)
{
    if (LX == 1) return sort_dpp<0xB1>(v);                       // quad_perm:[1,0,3,2]
    if (LX == 2) return sort_dpp<0x4E>(v);                       // quad_perm:[2,3,0,1]
    if (LX == 4) return sort_dpp<0x1B>(sort_dpp<0x141>(v));      // half mirror, then [3,2,1,0]
    if (LX == 8) return sort_dpp<0x128>(v);                      // row_ror:8
    return __shfl_xor(v, LX, 64);
}

template<int LX>
__device__ inline void sort_partner(double f, int i, double &fb, int &ib)
{
    const int lo = sort_xor_lane<LX>(__double2loint(f));
    const int hi = sort_xor_lane<LX>(__double2hiint(f));
    fb = __hiloint2double(hi, lo);
    ib = sort_xor_lane<LX>(i);
}

// one compare-exchange stage between lanes LX apart (partner distance j = E LX)
template<int E, int LX>
__device__ inline void sort_wave_stage(double (&kf)[E], int (&ki)[E], int e0, int j, int k)
{
#pragma unroll
    for (int u = 0; u < E; u++) {
        double fb;
        int ib;
        sort_partner<LX>(kf[u], ki[u], fb, ib);
        const int e = e0 + u;
        // keep the smaller of the two where this element is the low end of an ascending pair (or
        // the high end of a descending one), the larger otherwise; equal pairs (padding only:
        // real entries differ in the index) may swap freely
        const bool want_min = ((e & j) == 0) == ((e & k) == 0);
        if (pair_less(fb, ib, kf[u], ki[u]) == want_min) {
            kf[u] = fb;
            ki[u] = ib;
        }
    }
}

// The sort proper, E = M / 1024 consecutive elements per thread in REGISTERS (M = the padded
// size, 1024 E).  A compare-exchange stage with partner distance j runs
//   j <  E          inside the thread,
//   j <  64 E       between lanes of a wavefront (lane ^ (j / E), DPP inside a 16-lane row): no barrier,
//   j >= 64 E       through LDS (elements dumped once per merge phase, the classic in-place
//                   stage with one barrier each, then reloaded).
// For M = 4096 that is 14 barriers instead of 78.  Element e keeps the smaller of (itself, its
// partner e ^ j) iff ((e & j) == 0) == ((e & k) == 0).
template<int E, int T = 1024>
__device__ inline void bitonic_sort_regs(const double *f, int count, double *keys, int *idx,
        int *order, int *rank)
{
    constexpr int M = T * E;            // T = threads of the workgroup (1024, or 256 for <= 256 keys)
    const int tid = threadIdx.x;
    const int e0 = tid * E;
    double kf[E];
    int ki[E];
#pragma unroll
    for (int u = 0; u < E; u++) {
        const int e = e0 + u;
        kf[u] = e < count ? f[e] : __builtin_huge_val();
        ki[u] = e < count ? e : 0x7fffffff;
    }
    for (int k = 2; k <= M; k <<= 1) {
        int j = k >> 1;
        if (j >= 64 * E) {
#pragma unroll
            for (int u = 0; u < E; u++) {
                keys[e0 + u] = kf[u];
                idx[e0 + u] = ki[u];
            }
            __syncthreads();
            for (; j >= 64 * E; j >>= 1) {
#pragma unroll
                for (int v = 0; v < (E > 1 ? E / 2 : 1); v++) {
                    const int q = tid + T * v;
                    if (q < (M >> 1)) {
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
                }
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < E; u++) {
                kf[u] = keys[e0 + u];
                ki[u] = idx[e0 + u];
            }
        }
        for (; j >= E; j >>= 1) {
            switch (j / E) {
            case 1: sort_wave_stage<E, 1>(kf, ki, e0, j, k); break;
            case 2: sort_wave_stage<E, 2>(kf, ki, e0, j, k); break;
            case 4: sort_wave_stage<E, 4>(kf, ki, e0, j, k); break;
            case 8: sort_wave_stage<E, 8>(kf, ki, e0, j, k); break;
            case 16: sort_wave_stage<E, 16>(kf, ki, e0, j, k); break;
            default: sort_wave_stage<E, 32>(kf, ki, e0, j, k); break;
            }
        }
#pragma unroll
        for (int jj = E >> 1; jj > 0; jj >>= 1) {
            if (jj < k) {
#pragma unroll
                for (int u = 0; u < E; u++) {
                    if ((u & jj) == 0) {
                        const bool up = ((e0 + u) & k) == 0;
                        if (pair_less(kf[u | jj], ki[u | jj], kf[u], ki[u]) == up) {
                            const double tf = kf[u];
                            const int ti = ki[u];
                            kf[u] = kf[u | jj];
                            ki[u] = ki[u | jj];
                            kf[u | jj] = tf;
                            ki[u | jj] = ti;
                        }
                    }
                }
            }
        }
    }
    // the sorted pairs also to LDS: callers read the extremes there
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; u++) {
        const int e = e0 + u;
        keys[e] = kf[u];
        idx[e] = ki[u];
        if (e < count) {
            order[e] = ki[u];
            rank[ki[u]] = e;
        }
    }
    __syncthreads();
}

// ---- merge sort by merge path: 2048 / 4096 keys, 1024 threads ------------------------------------
// The bitonic network above spends n log^2 n / 4 compare-exchanges (78 stages at 4096 keys, every
// one a dozen vector instructions per element on 64-bit keys + indices): 48 us per launch of 256
// populations at lambda = 4096, issue-bound.  Here a thread sorts its E keys in registers, then
// log2(M / E) merge levels double the run length: a thread owns E consecutive OUTPUT positions of
// its pair of runs, finds where they start in the two inputs by a binary search along its merge
// path diagonal (fixed trip count per level, loads from clamped indices: no branches), and merges
// E steps sequentially -- n log n work, a third of the instructions.  The runs ping-pong between
// two LDS buffers; a level whose pair of runs lies inside the 64 E keys of ONE wavefront needs no
// workgroup barrier.  Same total order (fitness, then index): same result.
// 48 -> 36.5 us per launch of 256 populations at lambda = 4096, 28.8 -> 23.3 at 2048.
// kbuf: 2 M doubles, ibuf: 2 M ints (LDS).  On return the sorted pairs are in (*keys_out, *idx_out).
template<int E, int T = 1024>
__device__ inline void merge_sort_lds(const double *f, int count, double *kbuf, int *ibuf,
        int *order, int *rank, double **keys_out, int **idx_out)
{
    constexpr int M = T * E;
    const int tid = threadIdx.x;
    const int e0 = tid * E;
    double kf[E];
    int ki[E];
#pragma unroll
    for (int u = 0; u < E; u++) {
        const int e = e0 + u;
        kf[u] = e < count ? f[e] : __builtin_huge_val();
        ki[u] = e < count ? e : 0x7fffffff;
    }
    // the thread's own E keys: a bitonic network in registers (the last phase ascending
    // throughout, inner phases alternating as the network asks)
#pragma unroll
    for (int kk = 2; kk <= E; kk <<= 1)
#pragma unroll
        for (int jj = kk >> 1; jj > 0; jj >>= 1)
#pragma unroll
            for (int u = 0; u < E; u++)
                if ((u & jj) == 0) {
                    const bool asc = kk == E ? true : ((u & kk) == 0);
                    if (pair_less(kf[u | jj], ki[u | jj], kf[u], ki[u]) == asc) {
                        const double tf = kf[u];
                        const int ti = ki[u];
                        kf[u] = kf[u | jj];
                        ki[u] = ki[u | jj];
                        kf[u | jj] = tf;
                        ki[u | jj] = ti;
                    }
                }
    double *ks = kbuf, *kd = kbuf + M;
    int *is = ibuf, *id = ibuf + M;
#pragma unroll
    for (int u = 0; u < E; u++) {
        ks[e0 + u] = kf[u];
        is[e0 + u] = ki[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int steps = 1;                       // binary-search trips of this level: log2(run) + 1
    for (int r = E; r > 1; r >>= 1) steps++;
    for (int run = E; run < M; run <<= 1, steps++) {
        // (inputs written by other wavefronts: everything from the level whose pair outgrows the
        // 64 E keys of a wavefront on)
        if (2 * run > 64 * E) __syncthreads();
        const int base = e0 & ~(2 * run - 1);
        const int o = e0 - base;         // first output position of this thread inside its pair
        const double *A = ks + base, *B = A + run;
        const int *Ai = is + base, *Bi = Ai + run;
        // i = number of entries of A among the first o outputs: the smallest i in
        // [max(0, o - run), min(o, run)] with NOT (A[i] < B[o - 1 - i]).  The index is read only
        // where the fitness ties.  (Measured and dropped, each against 36.5 us at 4096 keys: the
        // pairs as 16-byte LDS entries, one ds_read_b128 per probe -- 48 us, the scattered 16-byte
        // reads and 64-byte-strided stores conflict on the banks; a 4-ary search with the E outputs
        // merged by a bitonic network in registers from a window of 2 E entries -- half the
        // dependent LDS round trips, more instructions: 45 us.  With 16 wavefronts on the CU the
        // sort is bound by the instructions it issues, not by their latency.)
        int lo = max(0, o - run), hi = min(o, run);
        for (int t = 0; t < steps; t++) {
            const bool open = lo < hi;
            const int mid = (lo + hi) >> 1;
            const int ia = min(mid, run - 1), ib = min(max(o - 1 - mid, 0), run - 1);
            const double fa = A[ia], fb = B[ib];
            bool less = fa < fb;
            if (fa == fb) less = Ai[ia] < Bi[ib];
            lo = (open && less) ? mid + 1 : lo;
            hi = (open && !less) ? mid : hi;
        }
        int i = lo, j = o - lo;
        double a = A[min(i, run - 1)], b = B[min(j, run - 1)];
        int pos[E];                       // where output u came from (an offset into the source buffer)
#pragma unroll
        for (int u = 0; u < E; u++) {
            bool less = a < b;
            if (a == b) less = Ai[min(i, run - 1)] < Bi[min(j, run - 1)];
            const bool take_a = j >= run || (i < run && less);
            kf[u] = take_a ? a : b;
            pos[u] = take_a ? i : run + j;
            i += take_a ? 1 : 0;
            j += take_a ? 0 : 1;
            if (u + 1 < E) {
                // only the head that was consumed is replaced: one read per step
                const double nx = A[take_a ? min(i, run - 1) : run + min(j, run - 1)];
                a = take_a ? nx : a;
                b = take_a ? b : nx;
            }
        }
#pragma unroll
        for (int u = 0; u < E; u++) ki[u] = Ai[pos[u]];
#pragma unroll
        for (int u = 0; u < E; u++) {
            kd[e0 + u] = kf[u];
            id[e0 + u] = ki[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double *tk = ks; ks = kd; kd = tk;
        int *ti = is; is = id; id = ti;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < E; u++) {
        const int e = e0 + u;
        if (e < count) {
            order[e] = ki[u];
            rank[ki[u]] = e;
        }
    }
    *keys_out = ks;
    *idx_out = is;
}

// keys/idx: LDS arrays of max(m, 1024) entries (m = power of two >= count); 1024 threads, or
// 256 threads when m <= 256 (the caller's launch decides: sort_threads(m))
__host__ __device__ inline int sort_threads(int m) { return m <= 256 ? 256 : 1024; }

__device__ inline void bitonic_sort_lds(const double *f, int count, int m, double *keys,
        int *idx, int *order, int *rank)
{
    if (blockDim.x == 256) bitonic_sort_regs<1, 256>(f, count, keys, idx, order, rank);
    else if (m <= 1024) bitonic_sort_regs<1>(f, count, keys, idx, order, rank);
    else if (m == 2048) bitonic_sort_regs<2>(f, count, keys, idx, order, rank);
    else if (m == 4096) bitonic_sort_regs<4>(f, count, keys, idx, order, rank);
    else bitonic_sort_regs<8>(f, count, keys, idx, order, rank);
}

} // namespace bbo
