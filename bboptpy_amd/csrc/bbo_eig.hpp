// bbo_eig.hpp -- symmetric eigendecomposition C = B diag(D^2) B^T inside ONE workgroup.
//
// The algorithm is the reference's (Cmaes::eigenDecomposition, cmaes.cpp:229-283:
// Householder tridiagonalisation = tred2 :285-381, implicit-shift QL = tql2 :383-456,
// ascending order :459-477, repair :250-266, square roots :269-271) with the same sign
// conventions, so the eigenvectors -- not only C -- agree with the reference.  What
// changes is the execution: CDNA4 has no fast serial core, so
//   * tred2's matrix-vector product and rank-2 update run over a 4-lanes-per-row tiling
//     of the active block (kept symmetric in full), every wavefront recomputes the O(n)
//     reductions redundantly instead of waiting for a broadcast: 2 barriers per step;
//   * the back-accumulation of the reflectors runs column-parallel: 2 barriers per step;
//   * tql2 is split into a PRODUCER (wavefront 0: the scalar shift/rotation recurrence on
//     (d, e) only, recording the Givens pairs) and CONSUMERS (one lane per row of the
//     eigenvector matrix, applying a recorded chunk) that overlap through two LDS
//     buffers -- the serial recurrence, ~1e2 cycles per rotation, is the critical path.
// The matrix lives in LDS when it fits (n <= 128), else in HBM/L2.
#pragma once

#include <type_traits>

#include "bbo_cma.hpp"
#include "bbo_eig_ql.hpp"
#include "bbo_eig_dc.hpp"

namespace bbo {

typedef double d4_eig __attribute__((ext_vector_type(4)));

constexpr int EIG_THREADS = 512;
constexpr int EIG_NMAX = 512;

struct EigPlan {
    int threads;      // workgroup size = 4 lanes per row: 128 (n <= 32), 256 (n <= 64), else 512
    int use_lds;      // matrix in LDS?
    int reg_path;     // n <= 128: tred2 + accumulation run out of registers
    int dc;           // tridiagonal stage by divide and conquer (bbo_eig_dc.hpp) instead of QL
    int hybrid;       // 128 < n <= 256: on-chip reduction, reflectors stashed (cma_eig_wy applies them)
    int lda;          // leading dimension of the work matrix
    int rc;           // Givens pairs per chunk buffer
    int vl;           // stride of the LDS vectors (>= 128, includes a 2-element front pad)
    size_t lds_bytes;
};

constexpr int EIG_NVEC = 9;   // d, e, u, w, g, h, tdiag, uh0, uh1

// one slab of the eigensolver's per-population global scratch ([work | Q_house | F | Q F])
// (+ 72: keeps the per-population stride off large powers of two -- 256 workgroups walking
// their slabs in step would otherwise land on the same L2 / HBM channels)
__host__ __device__ inline size_t eig_slab(int ld) { return (size_t) (ld + 32) * (ld + 32) + 72; }

// doubles of the `part` area behind the nine LDS vectors: [4][vl] column partial sums of the generic
// path; the hybrid (128 < n <= 256, D&C) keeps eight 128-column slabs of its transposed product and
// the previous step's u / w (2 x 128) there
__host__ __device__ inline int eig_part_doubles(int vl, bool reg_path, bool hybrid)
{
    if (reg_path) return 0;
    const int generic = 4 * vl;
    if (!hybrid) return generic;
    // vl >= 256 (n > 224): u / w of the previous step live in the free vector uh1 instead -- at
    // n = 256 the LDS matrix leaves no room for 256 more doubles here
    return vl >= 256 ? (generic > 1024 ? generic : 1024) : 1280;
}

inline EigPlan eig_plan(int n, int ld)
{
    EigPlan pl {};
    pl.threads = n <= 32 ? 128 : n <= 64 ? 256 : EIG_THREADS;
    pl.vl = (((n > 128 ? n : 128) + 31) & ~31) + 2;
    pl.reg_path = n <= 128 ? 1 : 0;
    pl.dc = 1;                         // (n <= EIG_NMAX) n > 128: matrix in global memory, top merge external
    pl.hybrid = (n > 128 && n <= 256) ? 1 : 0;   // above 256: streaming reduction with Q_house accumulated
    const size_t budget = 160 * 1024 - 1024;
    const size_t ints = (size_t) (2 * EIG_MAXSEQ * 3 + 8) * sizeof(int);
    const size_t vecs = ((size_t) EIG_NVEC * pl.vl
            + eig_part_doubles(pl.vl, pl.reg_path != 0, pl.hybrid != 0)) * sizeof(double);
    const int lda_lds = n | 1;
    const size_t mat = (size_t) n * lda_lds * sizeof(double);
    const size_t fixed = vecs + ints;
    // the chunk buffers need room for at least one full QL sweep (n-1 pairs) each
    if (fixed + mat + (size_t) 2 * n * 16 <= budget && pl.reg_path) {
        pl.use_lds = 1;
        pl.lda = lda_lds;
        size_t rc = (budget - fixed - mat) / (2 * 16);
        // (with the D&C stage the chunk buffers are only its work area -- the merge arrays, the
        // leaves' patches or one staged reflector panel, <= 2200 doubles counting from uv -- so a
        // few hundred pairs are ample, and several workgroups of a small matrix then fit one CU)
        const size_t cap = !pl.dc ? 4096 : n <= 64 ? 400 : 800;
        if (rc > cap) rc = cap;
        pl.rc = (int) rc;
        pl.lds_bytes = fixed + mat + (size_t) 2 * pl.rc * 16;
    } else {
        pl.use_lds = 0;
        pl.lda = (n + 31) & ~31;      // global work matrix: rows 256-byte aligned, unguarded 32-column groups
        if (pl.hybrid) {
            // no QL chunk buffers; a 128 x 128 LDS matrix for the register-resident tail
            pl.rc = 0;
            pl.lds_bytes = fixed + (size_t) 128 * 130 * sizeof(double);
        } else {
            pl.rc = 2048;
            pl.lds_bytes = fixed + (size_t) 2 * pl.rc * 16;
        }
    }
    return pl;
}

// 64 < n <= 128 with FEW matrices in flight (round 5): the plan of the kernels BEHIND the reduction
// when the decomposition is split over workgroups the way 128 < n <= 256 is -- the work matrix in
// global memory (L2), the top merge external, reflectors stashed.  The reduction itself keeps the
// LDS plan (eig_plan): its matrix lives in registers and its stash in LDS.
inline EigPlan eig_plan_split(int n, int ld)
{
    EigPlan pl = eig_plan(n, ld);
    pl.threads = EIG_THREADS;
    pl.vl = 128 + 2;
    pl.reg_path = 0;
    pl.hybrid = 1;
    pl.use_lds = 0;
    pl.lda = (n + 31) & ~31;
    pl.rc = 0;
    const size_t ints = (size_t) (2 * EIG_MAXSEQ * 3 + 8) * sizeof(int);
    const size_t vecs = ((size_t) EIG_NVEC * pl.vl + eig_part_doubles(pl.vl, false, true)) * sizeof(double);
    pl.lds_bytes = vecs + ints + (size_t) 128 * 130 * sizeof(double);
    return pl;
}

// Straight-line pieces of the register-resident tred2, specialised on how many 32-column
// groups the active block still covers (a guard inside the unrolled loops would split them into
// basic blocks and serialise the LDS reads they issue).
template<int AMAX>
__device__ inline double tred_matvec(const double (&a_)[4][8], double (&ur)[4][8], const double *uv,
        int q)
{
#pragma unroll
    for (int a = 0; a < AMAX; a++)
#pragma unroll
        for (int b = 0; b < 8; b += 2) {
            const double2 t2 = *reinterpret_cast<const double2*>(&uv[32 * a + 8 * q + b]);
            ur[a][b] = t2.x;
            ur[a][b + 1] = t2.y;
        }
    double acc0 = 0., acc1 = 0.;
#pragma unroll
    for (int a = 0; a < AMAX; a++)
#pragma unroll
        for (int b = 0; b < 8; b += 2) {
            acc0 = __builtin_fma(a_[a][b], ur[a][b], acc0);
            acc1 = __builtin_fma(a_[a][b + 1], ur[a][b + 1], acc1);
        }
    double acc = acc0 + acc1;
    acc = eig_quad_sum(acc);
    return acc;
}

template<int AMAX>
__device__ inline void tred_rank2(double (&a_)[4][8], const double (&ur)[4][8], const double *wv,
        double uj, double wj, int q)
{
    // The pieces of w for two 32-column groups are requested together (eight 16-byte reads in
    // flight): short of registers here, the scheduler otherwise asks for two or four pieces at a
    // time and waits for each lot -- six LDS round trips on the serial path of a step.
#pragma unroll
    for (int a0 = 0; a0 < AMAX; a0 += 2) {
        double2 w2[2][4];
#pragma unroll
        for (int da = 0; da < 2; da++)
#pragma unroll
            for (int b = 0; b < 8; b += 2)
                if (a0 + da < AMAX)
                    w2[da][b >> 1] = *reinterpret_cast<const double2*>(&wv[32 * (a0 + da) + 8 * q + b]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int da = 0; da < 2; da++)
#pragma unroll
            for (int b = 0; b < 8; b += 2)
                if (a0 + da < AMAX) {
                    const int a = a0 + da;
                    a_[a][b] = __builtin_fma(-ur[a][b], wj, __builtin_fma(-w2[da][b >> 1].x, uj, a_[a][b]));
                    a_[a][b + 1] = __builtin_fma(-ur[a][b + 1], wj,
                            __builtin_fma(-w2[da][b >> 1].y, uj, a_[a][b + 1]));
                }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template<int AMAX>
__device__ inline void accum_step(double (&a_)[4][8], const EigMat &As, int row, double h, int q,
        int n)
{
    double ur[4][8];
    double acc0 = 0., acc1 = 0.;
#pragma unroll
    for (int a = 0; a < AMAX; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int k = 32 * a + 8 * q + b;
            ur[a][b] = k < n ? As(row, k) : 0.;
        }
#pragma unroll
    for (int a = 0; a < AMAX; a++)
#pragma unroll
        for (int b = 0; b < 8; b += 2) {
            acc0 = __builtin_fma(ur[a][b], a_[a][b], acc0);
            acc1 = __builtin_fma(ur[a][b + 1], a_[a][b + 1], acc1);
        }
    double acc = acc0 + acc1;
    acc = eig_quad_sum(acc);
    const double gq = -(acc / h);
#pragma unroll
    for (int a = 0; a < AMAX; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) a_[a][b] = __builtin_fma(gq, ur[a][b], a_[a][b]);
}

// tred2 + reflector accumulation for n <= 128 with the matrix in REGISTERS: thread
// (row j = tid >> 2, class q = tid & 3) owns A(j, k) for k = 32a + 8q + b (a < 4, b < 8);
// LDS carries only the O(n) vectors (zero-padded beyond the active block, so the inner
// loops need no bounds tests) and the stashed Householder vectors (ROW i of Astash holds the
// vector of step i).  Phase 2 rebuilds Q column-tiled (thread owns Q(k, j) for its 32 rows k
// of column j) and needs no barrier at all: a column only reads the stashed vectors.
// Against cmaes.cpp:293-381 the Householder vectors are left unscaled (the reference divides
// by sum|d| first): the reflector I - u u^T / h is the same, one reduction per step is saved.
// (C, ld): the n x n matrix to reduce (the covariance, or -- for 128 < n' <= 256 -- the leading
// 128 x 128 block the global-memory steps have left); As: LDS matrix for the reflector stash;
// the accumulated Q goes to (qdst, ldq), which may be As itself.
template<int TT = EIG_THREADS>
__device__ inline void eig_tred_accum_reg128(const double *C, int ld, int n, const EigMat &As,
        double *dv, double *ev, double *uv, double *wv, double *gv, double *hvec, double *td,
        int tid, long long *stamps, double *qdst, int ldq, bool finish, bool accumulate = true)
{
    // (TT = 256 / 128: rows 0..63 / 0..31 only -- n <= 64 / 32)
    const int T = TT, lane = tid & 63;
    const int j = tid >> 2, q = tid & 3;
    double a_[4][8];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int k = 32 * a + 8 * q + b;
            a_[a][b] = (j < n && k < n) ? C[(size_t) j * ld + k] : 0.;
        }
    for (int k = tid; k < 128; k += T) {
        dv[k] = k < n ? C[(size_t) (n - 1) * ld + k] : 0.;
        hvec[k] = 0.;
    }
    __syncthreads();      // (C may live in the LDS block that becomes the stash: read it first)
    for (int x = tid; x < n * As.ld; x += T) As.a[x] = 0.;

    // One Householder step, specialised on how many 32-column groups the active block (rows and
    // columns < i) still covers: the matrix-vector product and the rank-2 update then shrink with
    // it instead of multiplying the zeros beyond column i.
    auto step = [&](int i, auto amax_tag) {
        constexpr int AMAX = decltype(amax_tag)::value;
        if ((tid >> 6) * 16 >= i) {
            // this wavefront's 16 rows are finished (the active block is rows < i): it only
            // keeps the two barriers of the step company and leaves the SIMD to the others
            __syncthreads();
            __syncthreads();
            return;
        }
        __syncthreads();
        // (three reads, one wait: a read under its position test is a branch and a wait of its own)
        const double dl0 = dv[lane], dl1 = dv[lane + 64], f = dv[i - 1];
        const double d0 = lane < i ? dl0 : 0.;
        const double d1 = lane + 64 < i ? dl1 : 0.;
        const double h0 = eig_wave_sum_bf(d0 * d0 + d1 * d1);
        if (h0 == 0.) {
            __syncthreads();
            if (tid == 0) {
                ev[i] = f;
                hvec[i] = 0.;
            }
            if (j == i - 1) {
#pragma unroll
                for (int a = 0; a < AMAX; a++)
#pragma unroll
                    for (int b = 0; b < 8; b++) dv[32 * a + 8 * q + b] = a_[a][b];
            }
            return;
        }
        // sqrt and reciprocal from the hardware estimates + Newton corrections (full fp64 to a
        // rounding error; every wavefront computes the same bits): these sit on the serial
        // chain of the step, the IEEE sequences are twice as long
        double g;
        {
            double y = __builtin_amdgcn_rsq(h0);
            const double err = fma(-h0 * y, y, 1.);
            y = fma(y * err, fma(err, 0.375, 0.5), y);
            g = h0 * y;
            g = fma(fma(-g, g, h0), 0.5 * y, g);      // one more step on the root itself
        }
        if (f > 0) g = -g;
        const double h = h0 - f * g;
        const double rh = dc_rcp(h);
        // every wavefront writes the SAME u (one store per element), so a wavefront may read
        // what it wrote without waiting for the others
        uv[lane] = lane < i ? (lane == i - 1 ? f - g : d0) : 0.;
        uv[lane + 64] = lane + 64 < i ? (lane + 64 == i - 1 ? f - g : d1) : 0.;
        if (tid == 0) ev[i] = g;
        // g = A u (no bounds tests: u is zero beyond the active block); this thread's
        // entries of u stay in registers for the rank-2 update below
        double ur[4][8];
        {
            const double acc = tred_matvec<AMAX>(a_, ur, uv, q);
            if (q == 0 && j < i) gv[j] = acc;
        }
        __syncthreads();
        const double gl0 = gv[lane], gl1 = gv[lane + 64];
        const double uu0 = uv[lane], uu1 = uv[lane + 64];
        const double e0 = lane < i ? gl0 * rh : 0.;
        const double e1 = lane + 64 < i ? gl1 * rh : 0.;
        const double hh = eig_wave_sum_bf(e0 * uu0 + e1 * uu1) * (0.5 * rh);
        wv[lane] = lane < i ? e0 - hh * uu0 : 0.;
        wv[lane + 64] = lane + 64 < i ? e1 - hh * uu1 : 0.;
        // A -= u w^T + w u^T; rows >= i and columns >= i see zeros and do not move
        {
            const double uj = uv[j], wj = wv[j];
            // stash: row i = the Householder vector of step i (here, with the read of u_j the
            // update needs anyway, not as a round trip of its own before the barrier)
            if (q == 0 && j < i) As(i, j) = uj;
            tred_rank2<AMAX>(a_, ur, wv, uj, wj, q);
            if (j == i - 1) {
#pragma unroll
                for (int a = 0; a < AMAX; a++)
#pragma unroll
                    for (int b = 0; b < 8; b++) dv[32 * a + 8 * q + b] = a_[a][b];
            }
        }
        if (tid == 0) hvec[i] = h;
    };
    {
        int i = n - 1;
        for (; i > 96; i--) step(i, std::integral_constant<int, 4>());
        for (; i > 64; i--) step(i, std::integral_constant<int, 3>());
        for (; i > 32; i--) step(i, std::integral_constant<int, 2>());
        for (; i > 0; i--) step(i, std::integral_constant<int, 1>());
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[1] = wall_clock64();
    // diagonal of the tridiagonal matrix: A(j, j) as it stands
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 8; b++)
            if (32 * a + 8 * q + b == j && j < n) td[j] = a_[a][b];

    if (accumulate) {
    // ---- phase 2: Q = H(n-1) ... H(1), column-tiled in registers, starting from I.  Column
    // j only needs the stashed vectors (read-only now): no barrier inside the loop ------------
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) a_[a][b] = (32 * a + 8 * q + b == j) ? 1. : 0.;
    // H(i+1) touches rows k <= i only, and columns j > i of Q are still columns of I there
    // (zero in those rows): a wavefront joins at the first step that reaches its columns, and
    // 32-row groups beyond the vector are skipped
    for (int i = (tid >> 6) * 16 > 0 ? (tid >> 6) * 16 - 1 : 0; i < n - 1; i++) {
        const double h = hvec[i + 1];
        if (h != 0.) {
            // rows 0..i; the stash rows are zero beyond the vector and beyond n
            switch ((i + 32) >> 5) {
            case 1: accum_step<1>(a_, As, i + 1, h, q, n); break;
            case 2: accum_step<2>(a_, As, i + 1, h, q, n); break;
            case 3: accum_step<3>(a_, As, i + 1, h, q, n); break;
            default: accum_step<4>(a_, As, i + 1, h, q, n); break;
            }
        }
    }
    __syncthreads();
    if (j < n) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int k = 32 * a + 8 * q + b;
                if (k < n) qdst[(size_t) k * ldq + j] = a_[a][b];
            }
    }
    }   // accumulate (else: the caller applies the stashed reflectors itself, eig_dc_phase)
    __syncthreads();   // td is complete
    if (finish)
        for (int k = tid; k < n; k += T) dv[k] = td[k];
    __syncthreads();
}

// ---------------------------------------------------------------------------
// tred2 + accumulation for n > 128: the matrix does not fit LDS or registers, it lives in
// global memory (L2-resident: <= 2 MB), row-major with lda a multiple of 32.  Same arithmetic
// as the register path above (unscaled reflectors, stash of reflector i in ROW i, Q built from
// I by applying H(1)..H(n-1) to a growing block).  What matters here is memory-level
// parallelism: one workgroup streams the active block 3x per tred step and 2x per
// accumulation step, so every inner loop loads a batch of rows/columns into registers before
// it computes (one L2 round trip per batch), and nothing is bounds-tested: u and w are kept
// zero beyond the active block.
// ---------------------------------------------------------------------------
// hybrid (As != null, 128 < n <= 256): once the active block is 128 x 128 the steps move into
// the register-resident code above (reflector stash in the LDS matrix As) and only the 128
// biggest steps of either phase stream from L2.
__device__ inline void eig_tred_accum_global(const double *C, int ld, int n, const EigMat &A,
        double *dv, double *ev, double *uv, double *wv, double *gv, double *hvec, double *td,
        double *part, int nv, int tid, long long *stamps, const EigMat *As, bool accumulate = true)
{
    const int NS = As ? 128 : 0;             // steps i < NS run out of registers
    const int T = EIG_THREADS, lane = tid & 63;
    const int lda = A.ld;
    const int nr = (n + 31) & ~31;           // columns touched by the unguarded loops
    // A = C, zero-padded to nr columns
    for (int q = tid; q < n * (nr >> 1); q += T) {
        const int r = q / (nr >> 1), c2 = (q - r * (nr >> 1)) * 2;
        double2 v;
        v.x = c2 < n ? C[(size_t) r * ld + c2] : 0.;
        v.y = c2 + 1 < n ? C[(size_t) r * ld + c2 + 1] : 0.;
        *reinterpret_cast<double2*>(&A.a[(size_t) r * lda + c2]) = v;
    }
    for (int k = tid; k < nr; k += T) {
        dv[k] = k < n ? C[(size_t) (n - 1) * ld + k] : 0.;
        hvec[k] = 0.;
        uv[k] = 0.;
        wv[k] = 0.;
    }
    // hybrid: the leading 128 x 128 block -- part of every streaming step -- sits in the LDS
    // matrix that later holds the register tail's stash (row stride 130: conflict-free 16-byte
    // reads by 4 lanes per row); the steps stream only the other three quarters from L2
    constexpr int LB = 130;
    double *L11 = As ? As->a : nullptr;
    if (L11)
        for (int q = tid; q < 128 * 64; q += T) {
            const int r = q >> 6, c2 = (q & 63) * 2;
            *reinterpret_cast<double2*>(&L11[r * LB + c2]) =
                    make_double2(C[(size_t) r * ld + c2], C[(size_t) r * ld + c2 + 1]);
        }
    const int rq = tid & 3, rj0 = tid >> 2;   // 4 lanes per row, lane q owns columns 32a + 8q + b
    // ... and the block below it (rows 128.., columns < 128) stays in REGISTERS: thread (row
    // 128 + rj0, class rq) keeps its 32 entries from load to the end of the streaming steps (the
    // rows turn into reflector stash one by one, written to global memory whole).  What still
    // streams from L2 is the half of the matrix in columns >= 128.
    double2 r21[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = 128 + rj0, c2 = 32 * a + 8 * rq + 2 * b;
            r21[a][b] = (L11 && r < n) ? make_double2(C[(size_t) r * ld + c2], C[(size_t) r * ld + c2 + 1])
                                       : make_double2(0., 0.);
        }
    for (int i = n - 1; i > 0 && i >= NS; i--) {
        __syncthreads();
        const int ir = (i + 31) & ~31, GA = ir >> 5;
        double hs = 0.;
        for (int k = lane; k < i; k += 64) hs += dv[k] * dv[k];
        const double h0 = eig_wave_sum(hs);
        const double f = dv[i - 1];
        if (h0 == 0.) {
            __syncthreads();
            if (tid == 0) {
                ev[i] = f;
                hvec[i] = 0.;
            }
            for (int k = tid; k < ir; k += T)
                dv[k] = k < i ? ((L11 && i - 1 < 128 && k < 128) ? L11[(i - 1) * LB + k] : A(i - 1, k))
                              : 0.;
            if (L11 && i - 1 >= 128) {
                // (columns < 128 of that row are register-resident: its owners supply them)
                __syncthreads();
                if (128 + rj0 == i - 1) {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            *reinterpret_cast<double2*>(&dv[32 * a + 8 * rq + 2 * b]) = r21[a][b];
                }
            }
            continue;
        }
        double g = sqrt(h0);
        if (f > 0) g = -g;
        const double h = h0 - f * g;
        // every wavefront writes the SAME u, so it may read its own copy without a barrier
        for (int k = lane; k < ir; k += 64) uv[k] = k < i ? (k == i - 1 ? f - g : dv[k]) : 0.;
        if (tid == 0) ev[i] = g;
        // g = A u over rows j < i
        for (int j = rj0; j < i; j += T / 4) {
            const double *row = A.a + (size_t) j * lda + 8 * rq;
            const double *lrow = (L11 && j < 128) ? L11 + j * LB + 8 * rq : nullptr;
            double acc0 = 0., acc1 = 0.;
            for (int a0 = 0; a0 < GA; a0 += 4) {
                const double *src = (lrow && a0 == 0) ? lrow : row + 32 * a0;   // columns < 128 of
                const bool inreg = L11 && a0 == 0 && j >= 128;                 // rows < 128: LDS,
                double2 x[4][4];                                               // rows >= 128: regs
                if (inreg) {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) x[a][b] = r21[a][b];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            x[a][b] = a0 + a < GA
                                    ? *reinterpret_cast<const double2*>(src + 32 * a + 2 * b)
                                    : make_double2(0., 0.);
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
                    if (a0 + a < GA) {
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const double2 u2 = *reinterpret_cast<const double2*>(
                                    &uv[32 * (a0 + a) + 8 * rq + 2 * b]);
                            acc0 = __builtin_fma(x[a][b].x, u2.x, acc0);
                            acc1 = __builtin_fma(x[a][b].y, u2.y, acc1);
                        }
                    }
            }
            double acc = acc0 + acc1;
            acc = eig_quad_sum(acc);
            if (rq == 0) gv[j] = acc;
        }
        __syncthreads();
        const double rh = 1. / h;
        double fs = 0.;
        for (int k = lane; k < i; k += 64) fs += (gv[k] * rh) * uv[k];
        const double hh = eig_wave_sum(fs) * (0.5 * rh);
        for (int k = lane; k < ir; k += 64) wv[k] = k < i ? gv[k] * rh - hh * uv[k] : 0.;
        // A -= u w^T + w u^T on rows j < i; row i-1 becomes the next d
        for (int j = rj0; j < i; j += T / 4) {
            double *row = A.a + (size_t) j * lda + 8 * rq;
            double *lrow = (L11 && j < 128) ? L11 + j * LB + 8 * rq : nullptr;
            const double uj = uv[j], wj = wv[j];
            for (int a0 = 0; a0 < GA; a0 += 4) {
                double *src = (lrow && a0 == 0) ? lrow : row + 32 * a0;
                const bool inreg = L11 && a0 == 0 && j >= 128;
                double2 x[4][4];
                if (inreg) {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++) x[a][b] = r21[a][b];
                } else {
#pragma unroll
                    for (int a = 0; a < 4; a++)
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            x[a][b] = a0 + a < GA
                                    ? *reinterpret_cast<const double2*>(src + 32 * a + 2 * b)
                                    : make_double2(0., 0.);
                }
#pragma unroll
                for (int a = 0; a < 4; a++)
                    if (a0 + a < GA) {
#pragma unroll
                        for (int b = 0; b < 4; b++) {
                            const int kk = 32 * (a0 + a) + 8 * rq + 2 * b;
                            const double2 u2 = *reinterpret_cast<const double2*>(&uv[kk]);
                            const double2 w2 = *reinterpret_cast<const double2*>(&wv[kk]);
                            double2 v;
                            v.x = __builtin_fma(-u2.x, wj, __builtin_fma(-w2.x, uj, x[a][b].x));
                            v.y = __builtin_fma(-u2.y, wj, __builtin_fma(-w2.y, uj, x[a][b].y));
                            if (inreg) r21[a][b] = v;
                            else *reinterpret_cast<double2*>(src + 32 * a + 2 * b) = v;
                            if (j == i - 1) *reinterpret_cast<double2*>(&dv[kk]) = v;
                        }
                    }
            }
        }
        // stash: row i = the Householder vector of step i (read back by the accumulation)
        for (int k = tid; k < i; k += T) A(i, k) = uv[k];
        if (tid == 0) hvec[i] = h;
    }
    __syncthreads();
    if (stamps && tid == 0) stamps[2] = wall_clock64();
    for (int j = NS + tid; j < n; j += T) td[j] = A(j, j);
    if (As) {
        // the leading 128 x 128 block: reduce and accumulate in registers, Q block back to A
        __syncthreads();
        eig_tred_accum_reg128(L11, LB, 128, *As, dv, ev, uv, wv, gv, hvec, td, tid, nullptr,
                A.a, lda, false, accumulate);
        for (int k = tid; k < nr; k += T) {
            uv[k] = 0.;
            wv[k] = 0.;
        }
    }
    __syncthreads();
    if (As && !accumulate) {
        // the reflectors are applied later in blocked form (cma_eig_wy): leave A = V, row i = u_i
        // in columns < i and zero from column i on; the first 128 rows come from the LDS stash
        for (int q = tid; q < n * (nr >> 1); q += T) {
            const int r = q / (nr >> 1), c2 = (q - r * (nr >> 1)) * 2;
            double2 *cell = reinterpret_cast<double2*>(&A.a[(size_t) r * lda + c2]);
            double2 v = r < 128 ? make_double2(c2 < 128 ? (*As)(r, c2) : 0.,
                                               c2 + 1 < 128 ? (*As)(r, c2 + 1) : 0.)
                                : *cell;
            if (c2 >= r) v.x = 0.;
            if (c2 + 1 >= r) v.y = 0.;
            *cell = v;
        }
        __syncthreads();
        for (int k = tid; k < n; k += T) dv[k] = td[k];
        __syncthreads();
        return;
    }
    // ---- Q = H(n-1) ... H(1), in place: the block [0..i]^2 holds the product so far, the rows
    // below still hold the stashed vectors.  The strict upper triangle must read as zero ------
    for (int q = tid; q < n * (nr >> 1); q += T) {
        const int r = q / (nr >> 1), c2 = (q - r * (nr >> 1)) * 2;
        double2 *cell = reinterpret_cast<double2*>(&A.a[(size_t) r * lda + c2]);
        double2 v = *cell;
        // (with the hybrid, rows < NS keep their first NS columns: the accumulated block)
        if (c2 > r && !(r < NS && c2 < NS)) v.x = 0.;
        if (c2 + 1 > r && !(r < NS && c2 + 1 < NS)) v.y = 0.;
        *cell = v;
    }
    if (tid == 0 && NS == 0) A(0, 0) = 1.;
    // column tiling: lane group cq in 0..3 owns the rows k = cq (mod 4) of columns cj, cj+128, ..
    const int cq = tid >> 7, cj0 = tid & 127;
    for (int i = NS > 0 ? NS - 1 : 0; i < n - 1; i++) {
        const double h = hvec[i + 1];
        __syncthreads();
        if (h != 0.) {
            for (int k = tid; k <= i; k += T) uv[k] = A(i + 1, k);
            __syncthreads();
            for (int j = cj0; j <= i; j += 128) {
                double acc0 = 0., acc1 = 0.;
                const double *col = A.a + j;
                int k = cq;
                for (; k + 28 <= i; k += 32) {
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = col[(size_t) (k + 4 * u) * lda];
#pragma unroll
                    for (int u = 0; u < 8; u += 2) {
                        acc0 = __builtin_fma(x[u], uv[k + 4 * u], acc0);
                        acc1 = __builtin_fma(x[u + 1], uv[k + 4 * u + 4], acc1);
                    }
                }
                for (; k <= i; k += 4) acc0 = __builtin_fma(col[(size_t) k * lda], uv[k], acc0);
                part[cq * nv + j] = acc0 + acc1;
            }
            __syncthreads();
            for (int j = cj0; j <= i; j += 128) {
                const double gq = -((part[j] + part[nv + j]) + (part[2 * nv + j] + part[3 * nv + j])) / h;
                double *col = A.a + j;
                int k = cq;
                for (; k + 28 <= i; k += 32) {
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) x[u] = col[(size_t) (k + 4 * u) * lda];
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        col[(size_t) (k + 4 * u) * lda] = __builtin_fma(gq, uv[k + 4 * u], x[u]);
                }
                for (; k <= i; k += 4) col[(size_t) k * lda] = __builtin_fma(gq, uv[k], col[(size_t) k * lda]);
            }
        }
        // row i+1 joins the block as a row of I (its stash is spent; column i+1 is already zero)
        for (int k = tid; k <= i + 1; k += T) A(i + 1, k) = k == i + 1 ? 1. : 0.;
    }
    __syncthreads();
    for (int k = tid; k < n; k += T) dv[k] = td[k];
    __syncthreads();
}

// ---------------------------------------------------------------------------
// 128 < n <= 256 (BIPOP's n = 256), D&C stage: the first n - 128 Householder steps with the
// whole ACTIVE matrix on chip, by symmetry.  Blocks of A = [L11 L21^T; L21 L22] (128 + (n - 128)):
//   L11  LDS (row stride 130), full, as before;
//   L22  registers, full, thread (row 128 + (tid >> 2), class tid & 3) owns 32 columns of its row
//        (the layout of the register-resident code above);
//   L21  registers, as 4-row x 8-column PATCHES: wavefront w, lane l owns rows 128 + 16 w +
//        4 (l >> 4) + k (k < 4) and columns 8 (l & 15) + c (c < 8);
//   L12  is NOT kept: round 2 streamed it (and L22) from L2 every step -- 29 k cycles per step,
//        19 k of them the rank-2 update's loads queueing behind its own stores.
// The product A u then needs L21 twice: rows (L21 u_top: four sums per thread, all-reduced over
// the 16 lanes of a DPP row by rotations) and columns (L21^T u_bot: eight sums per thread, reduced
// over the four row groups of the wavefront by two half-exchanges, v_permlane16_swap and
// v_permlane32_swap -- after the swap of (A, B) a lane holds its own and its partner's copy of the
// half it keeps, no select -- and over the wavefronts through an 8 x 128 LDS slab added in wave
// order, so the result does not depend on timing).  The rank-2 update touches L11 in LDS and the
// two register blocks; nothing streams.  Reflector i (i >= 128) goes to row i of the global work
// matrix (read back by cma_eig_wy), the leading 128 x 128 block then finishes in
// eig_tred_accum_reg128 exactly as before.  Sums are formed in a different order than the
// streaming code formed them: equal to rounding.
// ---------------------------------------------------------------------------
__device__ inline void eig_swap_rows16(double &a, double &b)     // a: odd 16-lane rows <-> b: even rows
{
    unsigned alo = __double2loint(a), ahi = __double2hiint(a);
    unsigned blo = __double2loint(b), bhi = __double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int) r1[0], (int) r0[0]);
    b = __hiloint2double((int) r1[1], (int) r0[1]);
}

__device__ inline void eig_swap_half32(double &a, double &b)     // a: lanes 32..63 <-> b: lanes 0..31
{
    unsigned alo = __double2loint(a), ahi = __double2hiint(a);
    unsigned blo = __double2loint(b), bhi = __double2hiint(b);
    auto r0 = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto r1 = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int) r1[0], (int) r0[0]);
    b = __hiloint2double((int) r1[1], (int) r0[1]);
}

__device__ inline double eig_row16_sum(double v)      // all 16 lanes of a DPP row get the row's sum
{
    v += eig_dpp<0x128>(v);   // row_ror:8
    v += eig_dpp<0x124>(v);   // row_ror:4
    v += eig_dpp<0x122>(v);   // row_ror:2
    v += eig_dpp<0x121>(v);   // row_ror:1
    return v;
}

// A REAL function (noinline): inlined into cma_eigen_impl, whose register allocation already runs
// at 256 VGPRs with ~490 spilled SGPRs, the 128 matrix registers of this loop were spilled and
// reloaded EVERY step (1.69 ms for the 128 steps of n = 256, no better than streaming from L2).
// It names the LDS vectors itself (same layout as cma_eigen_impl, from nv): through pointer
// arguments the compiler would lose the address space and emit flat loads.
#ifdef BBO_EIG_STEP_CLOCKS
#define EIG_CLK(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
        clk[slot] += t_ - tprev; tprev = t_; } while (0)
#else
#define EIG_CLK(slot) do { } while (0)
#endif
__device__ __attribute__((noinline)) void eig_tred_sym256_steps(const double *C, int ld, int n,
        double *Aglob, int lda, int nv, long long *stamps)
{
#ifdef BBO_EIG_STEP_CLOCKS
    unsigned long long clk[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int T = EIG_THREADS, LB = 130;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int nr = (n + 31) & ~31;
    double *dv = lds + 2, *ev = dv + nv, *uv = ev + nv, *wv = uv + nv, *gv = wv + nv;
    double *hvec = gv + nv, *td = hvec + nv, *gv2 = td + nv /* uh0 */, *uh1 = gv2 + nv;
    double *part = uh1 + nv - 2;
    // (hybrid plan: no chunk buffers, rc = 0 -- the LDS matrix follows the integer block)
    double *L11 = reinterpret_cast<double*>(reinterpret_cast<int*>(part + eig_part_doubles(nv, false, true))
            + 2 * EIG_MAXSEQ * 3 + 8);
    // ---- load: L11 -> LDS, L22 and L21 -> registers ------------------------------------------
    for (int q = tid; q < 128 * 64; q += T) {
        const int r = q >> 6, c2 = (q & 63) * 2;
        *reinterpret_cast<double2*>(&L11[r * LB + c2]) =
                make_double2(C[(size_t) r * ld + c2], C[(size_t) r * ld + c2 + 1]);
    }
    for (int k = tid; k < nr; k += T) {
        dv[k] = k < n ? C[(size_t) (n - 1) * ld + k] : 0.;
        hvec[k] = 0.;
        uv[k] = 0.;
        wv[k] = 0.;
    }
    const int rq = tid & 3, rj0 = tid >> 2;          // L11 / L22 rows: 4 lanes per row
    const int R22 = 128 + rj0;
    double2 r22[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int c2 = 128 + 32 * a + 8 * rq + 2 * b;
            r22[a][b] = make_double2((R22 < n && c2 < n) ? C[(size_t) R22 * ld + c2] : 0.,
                                     (R22 < n && c2 + 1 < n) ? C[(size_t) R22 * ld + c2 + 1] : 0.);
        }
    const int cg = lane & 15, rgw = lane >> 4;        // L21 patches
    const int P0 = 128 + 16 * wave + 4 * rgw;         // first row of this thread's patch
    double2 p21[4][4];                                // [row k][column pair b]: columns 8 cg + 2 b
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int r = P0 + k, c2 = 8 * cg + 2 * b;
            p21[k][b] = r < n ? make_double2(C[(size_t) r * ld + c2], C[(size_t) r * ld + c2 + 1])
                              : make_double2(0., 0.);
        }
    const int bit4 = (lane >> 4) & 1, bit5 = lane >> 5;
    const int mycol = 8 * cg + 4 * bit4 + 2 * bit5;   // the column pair this lane ends up with

    // per-thread LDS bases: everything below is base + immediate offset
    double *uq = uv + 8 * rq, *wq = wv + 8 * rq, *dq = dv + 8 * rq;     // 4-lanes-per-row layout
    double *uc = uv + 8 * cg, *wc = wv + 8 * cg, *dc = dv + 8 * cg;     // patch columns
    double *uP = uv + P0, *wP = wv + P0;                                // patch rows
    double *lrow = L11 + rj0 * LB + 8 * rq;
    // L11 lives in LDS, so its rank-2 update is DEFERRED by one step and applied on the fly when
    // the next product reads the block (one read + one write of the block per step instead of a
    // read for the product and a read + write for the update): up / wp keep the previous step's
    // u and w for rows / columns < 128 (zero before the first step)
    double *up = nv >= 256 ? uh1 : part + 1024, *wp = up + 128;
    double *upq = up + 8 * rq, *wpq = wp + 8 * rq;
    if (tid < 256) up[tid] = 0.;               // (up and wp are adjacent)
#define EIG_SEQ() __builtin_amdgcn_sched_barrier(0)   /* keeps one column group's loads in flight, not all */
    // The steps synchronise through LDS only.  __syncthreads() would also wait for the stash row's
    // GLOBAL stores (vmcnt(0): a round trip to L2, ~1 us, every step); nothing in this loop reads
    // them -- the full barrier after the loop orders them before the V pass.
#define EIG_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    __syncthreads();
    for (int i = n - 1; i >= 128; i--) {
        EIG_CLK(7);
        EIG_LDS_BARRIER();                             // dv = row i of A (columns < i, and A(i, i))
        EIG_CLK(0);
        const int ir = (i + 31) & ~31;
        const bool wact = 128 + 16 * wave < i;         // this wavefront's bottom rows are active
        const int GA2 = (i - 128 + 31) >> 5;           // 32-column groups of L22 still active
        double dk[4];
        double hs = 0.;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            // (unconditional read: lane + 64 m <= 255 stays inside the LDS vectors; masked after)
            const double t = dv[lane + 64 * m];
            dk[m] = lane + 64 * m < i ? t : 0.;
            hs = __builtin_fma(dk[m], dk[m], hs);
        }
        const double h0 = eig_wave_sum_bf(hs);
        const double f = dv[i - 1];
        const double tdi = dv[i];
        // h0 == 0 (the row is already reduced): u = 0, w = 0 -- the step then changes nothing and
        // publishes the next row like any other (tred2 skips the transformation, cmaes.cpp:300-305)
        const bool live = h0 != 0.;
        // root and reciprocal as in the register-resident steps: hardware estimates + Newton
        double g;
        {
            double y = __builtin_amdgcn_rsq(live ? h0 : 1.);
            const double hq = live ? h0 : 1.;
            const double err = fma(-hq * y, y, 1.);
            y = fma(y * err, fma(err, 0.375, 0.5), y);
            g = hq * y;
            g = fma(fma(-g, g, hq), 0.5 * y, g);
            if (!live) g = 0.;
        }
        if (f > 0) g = -g;
        const double h = h0 - f * g;
        const double rh = live ? dc_rcp(h) : 0.;
        // every wavefront writes the SAME u, so it may read its own copy without a barrier
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int k = lane + 64 * m;
            if (k < ir) uv[k] = (live && k < i) ? (k == i - 1 ? f - g : dk[m]) : 0.;
        }
        if (tid == 0) {
            ev[i] = live ? g : f;
            td[i] = tdi;
            hvec[i] = live ? h : 0.;
        }
        // stash: row i of the global work matrix = the Householder vector of step i (issued here so
        // that the store drains behind the product)
        if (tid < i) Aglob[(size_t) i * lda + tid] = (live && tid < i) ? (tid == i - 1 ? f - g : dv[tid]) : 0.;
        EIG_CLK(1);
        // ---- p = A u -------------------------------------------------------------------------
        {   // top rows j = rj0 < 128: L11 u_top (the L21^T u_bot part arrives through `part`),
            // with the previous step's update of L11 applied on the way
            double acc0 = 0., acc1 = 0.;
            const double ujp = up[rj0], wjp = wp[rj0];
#pragma unroll
            for (int a = 0; a < 4; a++) {
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    double2 x = *reinterpret_cast<const double2*>(lrow + 32 * a + 2 * b);
                    const double2 u2 = *reinterpret_cast<const double2*>(uq + 32 * a + 2 * b);
                    const double2 p2 = *reinterpret_cast<const double2*>(upq + 32 * a + 2 * b);
                    const double2 q2 = *reinterpret_cast<const double2*>(wpq + 32 * a + 2 * b);
                    x.x = __builtin_fma(-p2.x, wjp, __builtin_fma(-q2.x, ujp, x.x));
                    x.y = __builtin_fma(-p2.y, wjp, __builtin_fma(-q2.y, ujp, x.y));
                    *reinterpret_cast<double2*>(lrow + 32 * a + 2 * b) = x;
                    acc0 = __builtin_fma(x.x, u2.x, acc0);
                    acc1 = __builtin_fma(x.y, u2.y, acc1);
                }
                EIG_SEQ();
            }
            double acc = acc0 + acc1;
            acc = eig_quad_sum(acc);
            if (rq == 0) gv[rj0] = acc;
        }
        EIG_CLK(2);
        if (wact) {
            // bottom rows: L22 u_bot (row sums of the full symmetric block)
            double acc0 = 0., acc1 = 0.;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (a < GA2) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const double2 u2 = *reinterpret_cast<const double2*>(uq + 128 + 32 * a + 2 * b);
                        acc0 = __builtin_fma(r22[a][b].x, u2.x, acc0);
                        acc1 = __builtin_fma(r22[a][b].y, u2.y, acc1);
                    }
                }
                EIG_SEQ();
            }
            double acc = acc0 + acc1;
            acc = eig_quad_sum(acc);
            if (rq == 0) gv[R22] = acc;
            // L21: rows (L21 u_top -> gv2) and columns (L21^T u_bot -> part[wave])
            double2 ut[4];
#pragma unroll
            for (int b = 0; b < 4; b++) ut[b] = *reinterpret_cast<const double2*>(uc + 2 * b);
            const double2 ub0 = *reinterpret_cast<const double2*>(uP);
            const double2 ub1 = *reinterpret_cast<const double2*>(uP + 2);
            const double ub[4] = { ub0.x, ub0.y, ub1.x, ub1.y };
            double rp[4];
            double2 cp[4];
#pragma unroll
            for (int b = 0; b < 4; b++) cp[b] = make_double2(0., 0.);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                double a0 = 0., a1 = 0.;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    a0 = __builtin_fma(p21[k][b].x, ut[b].x, a0);
                    a1 = __builtin_fma(p21[k][b].y, ut[b].y, a1);
                    cp[b].x = __builtin_fma(p21[k][b].x, ub[k], cp[b].x);
                    cp[b].y = __builtin_fma(p21[k][b].y, ub[k], cp[b].y);
                }
                rp[k] = eig_row16_sum(a0 + a1);
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (cg == k) gv2[P0 + k] = rp[k];
            // half-exchange over lane bit 4: keep columns 0..3 (bit clear) or 4..7 (bit set)
            eig_swap_rows16(cp[0].x, cp[2].x);
            eig_swap_rows16(cp[0].y, cp[2].y);
            eig_swap_rows16(cp[1].x, cp[3].x);
            eig_swap_rows16(cp[1].y, cp[3].y);
            double2 s0 = make_double2(cp[0].x + cp[2].x, cp[0].y + cp[2].y);
            double2 s1 = make_double2(cp[1].x + cp[3].x, cp[1].y + cp[3].y);
            // ... and over lane bit 5: keep the first pair (bit clear) or the second (bit set)
            eig_swap_half32(s0.x, s1.x);
            eig_swap_half32(s0.y, s1.y);
            *reinterpret_cast<double2*>(&part[wave * 128 + mycol]) =
                    make_double2(s0.x + s1.x, s0.y + s1.y);
        } else {
            // retired (all rows of this wavefront are reflectors by now): its slab reads as zero
            *reinterpret_cast<double2*>(&part[wave * 128 + 2 * lane]) = make_double2(0., 0.);
        }
        EIG_CLK(3);
        EIG_LDS_BARRIER();
        EIG_CLK(4);
        // ---- w = p / h - (u^T p / 2 h^2) u, every wavefront for itself --------------------------
        {
            double ek[4], uk[4];
            double fs = 0.;
            // the column sums of L21^T u_bot: one slab per active wavefront, added in wave order
            // (all eight slabs, unconditionally: one LDS round trip; a retired wavefront's slab is zero)
            double pr[16];
#pragma unroll
            for (int w = 0; w < 8; w++) {
                pr[2 * w] = part[w * 128 + lane];
                pr[2 * w + 1] = part[w * 128 + 64 + lane];
            }
            double ps0 = 0., ps1 = 0.;
#pragma unroll
            for (int w = 0; w < 8; w++) {
                ps0 += pr[2 * w];
                ps1 += pr[2 * w + 1];
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int k = lane + 64 * m;
                const double g0 = gv[k], g2 = m >= 2 ? gv2[k] : (m == 0 ? ps0 : ps1);
                const double u0 = uv[k];
                const double gk = k < i ? g0 + g2 : 0.;
                uk[m] = k < i ? u0 : 0.;
                ek[m] = gk * rh;
                fs = __builtin_fma(ek[m], uk[m], fs);
            }
            const double hh = eig_wave_sum_bf(fs) * (0.5 * rh);
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int k = lane + 64 * m;
                const double wk = k < i ? ek[m] - hh * uk[m] : 0.;
                if (k < ir) wv[k] = wk;
                if (m < 2) {            // (every wavefront writes the same values)
                    up[k] = uk[m];
                    wp[k] = wk;
                }
            }
        }
        EIG_CLK(5);
        // ---- A -= u w^T + w u^T ------------------------------------------------------------------
        if (wact) {
            const double uj = uv[R22], wj = wv[R22];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                if (a < GA2) {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const double2 u2 = *reinterpret_cast<const double2*>(uq + 128 + 32 * a + 2 * b);
                        const double2 w2 = *reinterpret_cast<const double2*>(wq + 128 + 32 * a + 2 * b);
                        r22[a][b].x = __builtin_fma(-u2.x, wj, __builtin_fma(-w2.x, uj, r22[a][b].x));
                        r22[a][b].y = __builtin_fma(-u2.y, wj, __builtin_fma(-w2.y, uj, r22[a][b].y));
                    }
                }
                EIG_SEQ();
            }
            if (R22 == i - 1) {                // this row is the next d: its L22 part ...
#pragma unroll
                for (int a = 0; a < 4; a++)
                    if (128 + 32 * a < nr) {
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            *reinterpret_cast<double2*>(dq + 128 + 32 * a + 2 * b) = r22[a][b];
                    }
            }
            double2 ut[4], wt[4];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                ut[b] = *reinterpret_cast<const double2*>(uc + 2 * b);
                wt[b] = *reinterpret_cast<const double2*>(wc + 2 * b);
            }
            const double2 ub0 = *reinterpret_cast<const double2*>(uP);
            const double2 ub1 = *reinterpret_cast<const double2*>(uP + 2);
            const double2 wb0 = *reinterpret_cast<const double2*>(wP);
            const double2 wb1 = *reinterpret_cast<const double2*>(wP + 2);
            const double ub[4] = { ub0.x, ub0.y, ub1.x, ub1.y };
            const double wb[4] = { wb0.x, wb0.y, wb1.x, wb1.y };
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    p21[k][b].x = __builtin_fma(-ut[b].x, wb[k], __builtin_fma(-wt[b].x, ub[k], p21[k][b].x));
                    p21[k][b].y = __builtin_fma(-ut[b].y, wb[k], __builtin_fma(-wt[b].y, ub[k], p21[k][b].y));
                }
            // ... and its L21 part (columns < 128); four static blocks: a run-time index into
            // p21 would move the whole array to scratch memory
            const int pk = i - 1 - P0;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (pk == k) {
#pragma unroll
                    for (int b = 0; b < 4; b++) *reinterpret_cast<double2*>(dc + 2 * b) = p21[k][b];
                }
        }
    }
#ifdef BBO_EIG_STEP_CLOCKS
    if (stamps && tid == 0)
        for (int q = 0; q < 8; q++) stamps[40 + q] = (long long) clk[q];
#endif
    // the last step's update of L11, and row 127 as the d of the register-resident steps
    EIG_LDS_BARRIER();
    {
        const double ujp = up[rj0], wjp = wp[rj0];
#pragma unroll
        for (int a = 0; a < 4; a++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                double2 x = *reinterpret_cast<const double2*>(lrow + 32 * a + 2 * b);
                const double2 p2 = *reinterpret_cast<const double2*>(upq + 32 * a + 2 * b);
                const double2 q2 = *reinterpret_cast<const double2*>(wpq + 32 * a + 2 * b);
                x.x = __builtin_fma(-p2.x, wjp, __builtin_fma(-q2.x, ujp, x.x));
                x.y = __builtin_fma(-p2.y, wjp, __builtin_fma(-q2.y, ujp, x.y));
                *reinterpret_cast<double2*>(lrow + 32 * a + 2 * b) = x;
                if (rj0 == 127) *reinterpret_cast<double2*>(dq + 32 * a + 2 * b) = x;
            }
            EIG_SEQ();
        }
    }
#undef EIG_SEQ
#undef EIG_LDS_BARRIER
    __syncthreads();
}

// the rest of the reduction for 128 < n <= 256: the leading block in registers, then V for cma_eig_wy
__device__ inline void eig_tred_sym256(const double *C, int ld, int n, const EigMat &A,
        double *dv, double *ev, double *uv, double *wv, double *gv, double *hvec, double *td,
        int nv, int tid, long long *stamps, const EigMat &As, bool accumulate, double *Vout)
{
    constexpr int T = EIG_THREADS, LB = 130;
    const int nr = (n + 31) & ~31;
    eig_tred_sym256_steps(C, ld, n, A.a, A.ld, nv, stamps);
    if (stamps && tid == 0) stamps[2] = wall_clock64();
    // the leading 128 x 128 block: reduce in registers (its reflectors stay in the LDS stash)
    eig_tred_accum_reg128(As.a, LB, 128, As, dv, ev, uv, wv, gv, hvec, td, tid, nullptr,
            A.a, A.ld, false, accumulate);
    for (int k = tid; k < nr; k += T) {
        uv[k] = 0.;
        wv[k] = 0.;
    }
    __syncthreads();
    // V for cma_eig_wy, written straight to its place (Vout: n x n, dense): row i = u_i in
    // columns < i, zero from column i on; the first 128 rows come from the LDS stash, the others
    // from the rows of A they were stashed in
    const int lda = A.ld;
    for (int q = tid; q < n * n; q += T) {
        const int r = q / n, cidx = q - r * n;
        double v = 0.;
        if (cidx < r) v = r < 128 ? As(r, cidx) : A.a[(size_t) r * lda + cidx];
        Vout[q] = v;
    }
    __syncthreads();
    for (int k = tid; k < n; k += T) dv[k] = td[k];
    __syncthreads();
}

// LDSM: the work matrix A (reflector stash, then the eigenvector blocks of the divide and conquer)
// lives in LDS (pl.use_lds) -- a COMPILE-TIME fact here.  As the run-time choice
// `pl.use_lds ? LDS : global` the pointer is generic and every access to A -- each rotation of a QL
// leaf, the fragment reads of the merge products -- became a FLAT load or store, waited for with
// vmcnt(0) & lgkmcnt(0) in the middle of the recurrences (round 3: read in the ISA).
// HYB (matrix in global memory only): 1 = 128 < n <= 256, 0 = 256 < n <= 512 -- a kernel each
// STAGE (128 < n <= 256 only, round 4): 0 = the whole decomposition in this workgroup; 1 = the
// Householder reduction only -- the tridiagonal form (d, e) and the reflectors' scalars go to
// eig_work[3] = [d | e | h | .] and cma_eig_halves / STAGE 2 take over; 2 = the top merge of the two
// halves cma_eig_halves has solved (their eigenvalues at eig_work[3] + 3 n, their eigenvector blocks
// on the diagonal of the work matrix), the reflectors' T factors and the closing repair / root.
template<int TT, bool LDSM, int HYB = 1, int STAGE = 0>
__device__ __forceinline__ void cma_eigen_impl(const CmaDev &d, const CmaConst &c, const EigPlan &pl,
        int force)
{
    const int p = blockIdx.x;
    CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (STAGE == 2 || STAGE == 4) {
        if (sc->eig_stage != 1) return;
    } else if (STAGE == 3) {
        // (cma_tred_mw made the decision and did -- or did not -- its part: checked below)
    } else
    // cmaes.cpp:233: skip until enough evaluations have passed
    if (!force && !((double) (sc->fev - sc->eigenlastev) > c.eigenfreq)) {
        if (threadIdx.x == 0) {
            sc->eigen_done = 0;
            sc->eig_stage = 0;
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, T = TT, lane = tid & 63, wave = tid >> 6;
    // (LDSM: n <= 128 and the vectors are 130 apart, eig_plan -- a compile-time fact makes every LDS
    // vector and the whole merge work area constant addresses instead of scalar registers)
    const int n = c.n, ld = c.ld, nv = LDSM ? 130 : pl.vl;
    double *dv = lds + 2;         // diagonal / eigenvalues        (each vector: 2-element front pad)
    double *ev = dv + nv;         // sub-diagonal
    double *uv = ev + nv;         // Householder vector u / scaled reflector column
    double *wv = uv + nv;         // w = (A u)/h - hh u
    double *gv = wv + nv;         // A u, later sorted eigenvalues
    double *hvec = gv + nv;       // h of every Householder step
    double *td = hvec + nv;       // diagonal of the tridiagonal matrix
    double *uh0 = td + nv, *uh1 = uh0 + nv;
    double *part = uh1 + nv - 2;  // [4][nv] column partial sums (generic path only)
    double2 *rot = reinterpret_cast<double2*>(part
            + eig_part_doubles(nv, pl.reg_path != 0, pl.hybrid != 0));   // [2][rc]
    int *ibuf = reinterpret_cast<int*>(rot + 2 * pl.rc);        // desc[2][MAXSEQ*3], nseq[2], done[2], misc
    int *desc = ibuf;
    int *nseq = ibuf + 2 * EIG_MAXSEQ * 3;
    int *sdone = nseq + 2;
    int *perm = reinterpret_cast<int*>(uv);                     // reused after QL
    EigMat A { LDSM ? reinterpret_cast<double*>(ibuf + 2 * EIG_MAXSEQ * 3 + 8)
                    : d.eig_work + (size_t) p * 4 * eig_slab(ld), pl.lda };
    double *C = d.C + (size_t) p * ld * ld;

#define EIG_STAMP(slot) do { if (d.stamps && p == 0 && tid == 0) d.stamps[slot] = wall_clock64(); } while (0)
    EIG_STAMP(0);
    if (tid < 2) {   // front pads read by the QL prefetch
        dv[-1 - tid] = 0.;
        ev[-1 - tid] = 0.;
    }
    double *tri = d.eig_work + (size_t) (4 * p + 3) * eig_slab(ld);     // (STAGE 1 / 2 hand-over)
    if (STAGE == 3) {
        // the tail of a reduction cma_tred_mw began (bbo_eig_mw.hpp): the leading 128 x 128 block, as
        // the spread steps left it in eig_work[0], through the register-resident reduction; d, e, h of
        // rows < 128 and the reflectors' rows < 128 join what cma_tred_mw has written already
        // (the hand-over flag: 1 behind one spread kernel, 2 behind two -- `force` says which; anything
        // else: it skipped this generation, or a wavefront gave up)
        if (tri[4 * n + 1] != (force ? 2. : 1.) || sc->eig_mw_fail) return;
        constexpr int LB = 130;
        EigMat Ast { reinterpret_cast<double*>(ibuf + 2 * EIG_MAXSEQ * 3 + 8), LB };
        const double *L11 = d.eig_work + (size_t) (4 * p) * eig_slab(ld);
        for (int q = tid; q < 128 * 64; q += T) {
            const int r = q >> 6, c2 = (q & 63) * 2;
            *reinterpret_cast<double2*>(&Ast.a[r * LB + c2]) =
                    *reinterpret_cast<const double2*>(&L11[(size_t) r * 128 + c2]);
        }
        __syncthreads();
        eig_tred_accum_reg128(Ast.a, LB, 128, Ast, dv, ev, uv, wv, gv, hvec, td, tid, nullptr, A.a, A.ld,
                false, false);
        double *Vout = d.eig_work + (size_t) (4 * p + 1) * eig_slab(ld);
        for (int i = tid; i < 128; i += T) {
            tri[i] = td[i];
            if (i >= 1) tri[n + i - 1] = ev[i];
            tri[2 * n + i] = hvec[i];
        }
        // (a row per wavefront at a time: indexed by q / n this copy was an integer division per
        // element, 4 us of the kernel)
        for (int r = wave; r < 128; r += T / 64)
            for (int cidx = lane; cidx < n; cidx += 64) Vout[(size_t) r * n + cidx] = cidx < r ? Ast(r, cidx) : 0.;
        if (tid == 0) sc->eig_stage = 1;
        return;
    }
    if (STAGE == 2 || STAGE == 4) {
        // (STAGE 4, 256 < n <= 512: the tridiagonal matrix itself -- the reduction ran as
        // cma_tred_mw512 + cma_tred_tail and left its reflectors stashed; tri lies where the merges'
        // scratch begins, so it is read before anything else)
        for (int i = tid; i < n; i += T) {
            dv[i] = STAGE == 2 ? tri[3 * n + i] : tri[i];
            ev[i] = tri[n + i];
            hvec[i] = tri[2 * n + i];
        }
        __syncthreads();
    } else {
    if (LDSM) {      // (use_lds implies the register-resident reduction: n <= 128)
        // (with the D&C stage the reflectors stay stashed in A: eig_dc_phase applies them to the
        // tridiagonal eigenvectors in blocked form on the matrix cores)
        eig_tred_accum_reg128<TT>(C, ld, n, A, dv, ev, uv, wv, gv, hvec, td, tid,
                (d.stamps && p == 0) ? d.stamps : nullptr, A.a, A.ld, true,
                !(pl.dc && !(d.dbg & 2)));
    } else if (TT == EIG_THREADS && !LDSM) {
        const bool hybrid = HYB != 0;        // 128 < n <= 256: LDS holds a 128 x 128 stash matrix
        EigMat Ast { reinterpret_cast<double*>(ibuf + 2 * EIG_MAXSEQ * 3 + 8), 128 };
        // (with the D&C stage the reflectors stay stashed: cma_eig_wy applies them in blocked form)
        // (diagnostic bit 1024: round 2's form of the first n - 128 steps, streaming from L2)
        if (hybrid && !(d.dbg & 2) && !(d.dbg & 1024))
            eig_tred_sym256(C, ld, n, A, dv, ev, uv, wv, gv, hvec, td, nv, tid,
                    (d.stamps && p == 0) ? d.stamps : nullptr, Ast, false,
                    d.eig_work + (size_t) (4 * p + 1) * eig_slab(ld));
        else
        eig_tred_accum_global(C, ld, n, A, dv, ev, uv, wv, gv, hvec, td, part, nv, tid,
                (d.stamps && p == 0) ? d.stamps : nullptr, hybrid ? &Ast : nullptr,
                !(hybrid && !(d.dbg & 2)));
    }   // generic path
    {   // tql2 prologue: shift the sub-diagonal down (cmaes.cpp:384-387); T >= n
        const double t = (tid + 1 < n) ? ev[tid + 1] : 0.;
        __syncthreads();
        if (tid < n) ev[tid] = t;
    }
    __syncthreads();
    }   // STAGE != 2, 4
    EIG_STAMP(3);
    if (STAGE == 1 || STAGE == 5) {
        for (int i = tid; i < n; i += T) {
            tri[i] = dv[i];
            tri[n + i] = ev[i];
            tri[2 * n + i] = hvec[i];
        }
        if (STAGE == 5) {
            // (64 < n <= 128, few matrices: the reflectors leave the LDS stash -- row i = u_i, zero
            // from column i on -- for the place cma_eig_halves' third workgroup and cma_eig_wy4 read)
            double *Vout = d.eig_work + (size_t) (4 * p + 1) * eig_slab(ld);
            for (int r = wave; r < n; r += T / 64)
                for (int cidx = lane; cidx < n; cidx += 64) Vout[(size_t) r * n + cidx] = A(r, cidx);
        }
        if (tid == 0) {
            tri[4 * n] = 0.;               // (T factors: not built yet, cma_eig_halves' third workgroup)
            sc->eig_stage = 1;
        }
        return;
    }

    const bool use_dc = pl.dc && !(d.dbg & 2);
    if (use_dc) {
        // divide and conquer on the tridiagonal matrix; writes B (ascending eigenvalues)
        double *scr = uv;
        DcMat Qm { A.a, A.ld };
        // per-population global scratch: [work matrix | Q_house | F | Q F], eig_slab(ld) each
        double *Gp = d.eig_work + (size_t) (4 * p + 1) * eig_slab(ld);
        double *Bp_ = d.B + (size_t) p * ld * ld;
        long long *st_ = (d.stamps && p == 0 && !(STAGE == 2 && (d.dbg & 2048))) ? d.stamps : nullptr;
        if (LDSM || HYB || TT != EIG_THREADS) {
            // n <= 256: the reflectors are stashed (hv = 1 / their scalars)
            eig_dc_phase<TT, false, !LDSM>(Qm, n, dv, ev, Gp, Bp_, ld, scr, st_, d.dbg, LDSM ? 0 : 1, hvec,
                    !LDSM && !(d.dbg & 2) && !(d.dbg & 1024),   // (hybrid: V already in its place)
                    nullptr, STAGE == 2 ? 2 : 0, STAGE == 2 && tri[4 * n] != 0.,
                    STAGE == 2 ? force : 0, tri + 4 * n + 8);      // (STAGE 2: `force` = the part)
            if (STAGE == 2 && force == 1) return;
        } else if (TT == EIG_THREADS && !LDSM) {
            // 256 < n <= 512: the streaming reduction has accumulated Q_house, and
            // B = Q_house ((Q_1 (+) Q_2) F) is two cma_eig_gemm launches; merges the
            // register-resident product cannot hold go through the slab the first of them fills
            // (STAGE 4: stashed reflectors instead of an accumulated Q_house -- V is in its place,
            // the panels' T factors are built behind the top merge, cma_eig_wy4_512 applies them)
            eig_dc_phase<TT, true, true>(Qm, n, dv, ev, Gp, Bp_, ld, scr, st_, d.dbg, 1,
                    STAGE == 4 ? hvec : nullptr, STAGE == 4,
                    d.eig_work + (size_t) (4 * p + 3) * eig_slab(ld));
        }
    } else
    // ---- implicit QL (cmaes.cpp:388-456), producer / consumer over two chunk buffers -----
    {
        QlState st { 0, 0, 1, 0, 0., 0., 0 };
        int prev_done = 0;
        for (int epoch = 0;; epoch++) {
            const int cur = epoch & 1;
            if (wave == 0) {
                if (!prev_done) {
                    const int ns = ql_produce(st, n, dv, ev, rot + (size_t) cur * pl.rc,
                            desc + cur * EIG_MAXSEQ * 3, pl.rc, lane);
                    if (lane == 0) {
                        nseq[cur] = ns;
                        sdone[cur] = st.done;
                        if (d.stamps && p == 0) {   // diagnostic: sweeps and Givens pairs so far
                            int pairs = 0;
                            for (int q = 0; q < ns; q++)
                                pairs += desc[cur * EIG_MAXSEQ * 3 + 3 * q + 1]
                                        - desc[cur * EIG_MAXSEQ * 3 + 3 * q];
                            d.stamps[8] = (epoch == 0 ? 0 : d.stamps[8]) + ns;
                            d.stamps[9] = (epoch == 0 ? 0 : d.stamps[9]) + pairs;
                            d.stamps[10] = epoch + 1;
                        }
                    }
                }
            } else if (epoch > 0) {
                const int prv = cur ^ 1;
                const int ns = nseq[prv];
                if (!(d.dbg & 1))
                for (int k = tid - 64; k < n; k += T - 64)
                    ql_apply_row(A, k, rot + (size_t) prv * pl.rc, desc + prv * EIG_MAXSEQ * 3,
                            ns);
            }
            __syncthreads();
            if (prev_done) break;
            prev_done = sdone[cur];
        }
    }
    __syncthreads();
    EIG_STAMP(4);

    // ---- ascending order (cmaes.cpp:459-477), repair (:250-266), sqrt (:269-271) ---------
    if (use_dc) {
        for (int j = tid; j < n; j += T) gv[j] = dv[j];   // already ascending
    } else {
    for (int j = tid; j < n; j += T) {
        const double dj = dv[j], kj = dc_key(dj);
        int r = 0;
        for (int k = 0; k < n; k++) {
            const double dk = dc_key(dv[k]);      // (NaN ranks as +inf: perm stays a permutation)
            r += (dk < kj) || (dk == kj && k < j);
        }
        perm[j] = r;
        gv[r] = dj;   // sorted eigenvalues
    }
    }
    __syncthreads();
    const double lo = gv[0], hi = gv[n - 1];
    __syncthreads();
    if (lo <= 0.) {
        const double shift = fmax(hi, 0.) / 1e14;
        for (int i = tid; i < n; i += T) {
            gv[i] = fmax(gv[i], 0.) + shift;
            C[(size_t) i * ld + i] += shift;
        }
        __syncthreads();
    }
    const double lo2 = gv[0], hi2 = gv[n - 1];
    __syncthreads();
    if (hi2 > 1e14 * lo2) {
        const double shift = hi2 / 1e14 - lo2;
        for (int i = tid; i < n; i += T) {
            gv[i] += shift;
            C[(size_t) i * ld + i] += shift;
        }
        __syncthreads();
    }
    double *Dp = d.D + (size_t) p * ld;
    double *Bp = d.B + (size_t) p * ld * ld;
    for (int i = tid; i < ld; i += T) Dp[i] = i < n ? sqrt(gv[i]) : 1.;
    if (!use_dc)
        for (int k = wave; k < n; k += T / 64)
            for (int j = lane; j < n; j += 64) Bp[(size_t) k * ld + perm[j]] = A(k, j);
    if (tid == 0) {
        sc->eigenlastev = sc->fev;
        sc->eigen_done = 1;
        sc->eig_stage = 0;
    }
    if (use_dc && LDSM && c.lazy_isc) {
        // the sampler's packed operand B D straight from the LDS copy of B (what cma_post would
        // re-read B for; C^-1/2 is not formed in this configuration, see CmaConst::lazy_isc):
        // element (i, j) -> column tile i >> 4, k-step j >> 2, lane (j & 3, i & 15)
        __syncthreads();
        for (int j = tid; j < n; j += T) dv[j] = sqrt(gv[j]);
        __syncthreads();
        double *BDp = d.BDp + (size_t) p * ld * ld;
        const int KS = ld >> 2;
        for (int q = tid; q < ld * ld; q += T) {
            const int t4 = q >> 6, l = q & 63;
            const int nt = t4 / KS, ks = t4 - nt * KS;
            const int i = nt * 16 + (l & 15), j = 4 * ks + (l >> 4);
            BDp[q] = (i < n && j < n) ? A(i, j) * dv[j] : 0.;
        }
        if (tid == 0) sc->basis_ok = 1;
    }
    EIG_STAMP(5);
#undef EIG_STAMP
}

// The kernels proper: four lanes per matrix row, so the workgroup shrinks with n and several small
// matrices share a CU (the 512-thread form owns a CU's whole register file).
// (two kernels for the 512-thread form -- matrix in LDS, n <= 128, and matrix in global memory --
// so that each gets a register allocation of its own)
__global__ __launch_bounds__(512) void cma_eigen(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<512, true>(d, c, pl, force);
}
__global__ __launch_bounds__(512) void cma_eigen_g(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<512, false, 1>(d, c, pl, force);
}
__global__ __launch_bounds__(512) void cma_eigen_b(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<512, false, 0>(d, c, pl, force);
}
// 128 < n <= 256 split over workgroups (round 4): the reduction (one workgroup, as before), then the
// two HALVES of the torn tridiagonal matrix as problems of their own, each by a workgroup with its
// eigenvector block in LDS (the n <= 128 divide and conquer: 8-row leaves two to a wavefront, four
// merge levels in LDS -- ~0.2 ms side by side, where the single workgroup spent ~0.55 ms on 16-row
// leaves in two rounds and three merge levels with the blocks in L2), then the top merge.
__global__ __launch_bounds__(512) void cma_eigen_g1(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<512, false, 1, 1>(d, c, pl, force);
}
// 64 < n <= 128 with few matrices in flight (round 5): the reduction alone -- matrix in registers,
// reflector stash in LDS, 1.15 us per step: nothing spread over compute units beats it -- and the
// hand-over to cma_eig_halves / cma_eigen_g2 / cma_eig_secular / cma_eig_gemm1 / cma_eig_wy4, which
// put the divide and conquer and the reflectors of ONE matrix on 2, 1, n / 32, 16 and n / 16
// workgroups where cma_eigen keeps them on the workgroup that reduced
__global__ __launch_bounds__(512) void cma_eigen_r1(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<512, true, 1, 5>(d, c, pl, force);
}
// 256 < n <= 512 behind a spread reduction (cma_tred_mw512 + cma_tred_tail): the divide and conquer
// on the tridiagonal matrix they left, reflectors stashed (the top merge's products follow as
// cma_eig_gemm and cma_eig_wy4_512)
__global__ __launch_bounds__(512) void cma_eigen_b4(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<512, false, 0, 4>(d, c, pl, force);
}
// the tail of a reduction that cma_tred_mw began (the leading 128 x 128 block on one workgroup)
__global__ __launch_bounds__(512) void cma_tred_tail(CmaDev d, CmaConst c, EigPlan pl, int chained)
{
    cma_eigen_impl<512, false, 1, 3>(d, c, pl, chained);
}
// (`part`: 0 = the whole top merge; 1 = up to the secular equation, 2 = from behind it, with
// cma_eig_secular in between)
__global__ __launch_bounds__(512) void cma_eigen_g2(CmaDev d, CmaConst c, EigPlan pl, int part)
{
    cma_eigen_impl<512, false, 1, 2>(d, c, pl, part);
}
// The secular equation of the top merge of a 128 < n <= 256 matrix on ceil(n / 32) workgroups: 32
// roots each, 16 lanes per root, the poles of a lane in registers.  On the one workgroup of
// cma_eigen_g2 it is 82 us of vector issue (two lanes per root, 128 poles per lane from LDS) while
// the other CUs idle: the loop is bound by that CU's issue rate, not by latency -- 1024 threads
// there changed nothing.  grid (ceil(n / 32), P), 512 threads.
__global__ __launch_bounds__(512) void cma_eig_secular(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (sc->eig_stage != 1) return;
    const int n = c.n;
    double *tri = d.eig_work + (size_t) (4 * p + 3) * eig_slab(c.ld);
    const DcWork W = dc_work_layout(tri + 4 * n + 8, n + 2, nullptr);
    const int k = W.cnt[0];
    const double rho = W.red[14];
    dc_secular<16>(W, k, rho, 512 * (int) blockIdx.x + (int) threadIdx.x, true, nullptr, false, n);
}

// ---------------------------------------------------------------------------
// The rest of a split top merge on ceil(n / 32) workgroups each (round 5): on the one workgroup of
// cma_eigen_g2 part 2 the Loewner vector and the eigenvector factor F -- two passes of k^2
// reciprocals -- were 2 x 22 us at n = 256 (2 x 7 at 128), the CU's issue rate again, as the secular
// equation was.  Two launches because each needs ALL of the one before: the Loewner product of a
// pole runs over every root, the norm of a column over every pole's Loewner value.
//   cma_eig_lowner: pole i = 32 bx + tid / 16 by 16 lanes: what[i] = sign(z_i) sqrt |prod_j
//       ((d_orgj - d_i) + mu_j) / (d_j - d_i)| -> the work-area image in global memory.
//   cma_eig_fcols: every workgroup ranks the m eigenvalues (roots and deflated poles: two sorted
//       lists, or by counting when a deflation rotation or an escaped root may have disturbed them)
//       into LDS -- output order, row map, column map, 1 us of redundant work --, then its 32
//       output columns of F by 16 lanes each: the column's norm over all poles, its m entries
//       (rows in ORIGINAL column order), the deflation rotations on its columns.  Workgroup 0 also
//       does what cma_eigen_impl does behind the merge: eigenvalues in ascending order, the
//       reference's repair (cmaes.cpp:250-266), D = sqrt, the counters.
// Same expressions as dc_merge_level; the sums and products run over the lanes in another order,
// so the bits differ from the one-workgroup form (forms-agree tests: eigenvalues to 1e-12, each
// form's own residual and orthogonality).  Even n only (the T factors were built beside the halves).
// ---------------------------------------------------------------------------
__device__ inline double eig_row16_prod(double v)
{
    v *= eig_dpp<0x128>(v);   // row_ror:8
    v *= eig_dpp<0x124>(v);   // row_ror:4
    v *= eig_dpp<0x122>(v);   // row_ror:2
    v *= eig_dpp<0x121>(v);   // row_ror:1
    return v;
}

__global__ __launch_bounds__(512) void cma_eig_lowner(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (sc->eig_stage != 1) return;
    const int n = c.n;
    double *tri = d.eig_work + (size_t) (4 * p + 3) * eig_slab(c.ld);
    const DcWork W = dc_work_layout(tri + 4 * n + 8, n + 2, nullptr);
    const int k = W.cnt[0];
    const int i = 32 * (int) blockIdx.x + ((int) threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (k == 1) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            W.what[0] = 1.;
            W.ninv[0] = 1.;
        }
        return;
    }
    double prod = 1.;
    if (i < k) {
        // lam_jj - d_i = (d_org - d_i) + mu, paired with a denominator d_jj - d_i (the origin poles
        // were left in W.lam by cma_eig_secular); four roots at a time, all loads first
        const double di2 = W.dl[i];
        double p4[4] = { 1., 1., 1., 1. };
        const int npl = (k + 15) >> 4;
        for (int t0 = 0; t0 < npl; t0 += 4) {
            double lo_[4], mu_[4], dj_[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jc = min(sub + (t0 + u) * 16, k - 1);
                lo_[u] = W.lam[jc];
                mu_[u] = W.mu[jc];
                dj_[u] = W.dl[jc];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int jj = sub + (t0 + u) * 16;
                const double num = (lo_[u] - di2) + mu_[u];
                const double f = num * dc_rcp(jj == i ? 1. : dj_[u] - di2);
                p4[u] *= jj < k ? f : 1.;
            }
        }
        prod = (p4[0] * p4[1]) * (p4[2] * p4[3]);
    }
    prod = eig_row16_prod(prod);
    if (i < k && sub == 0) {
        const double v = sqrt(fabs(prod));
        W.what[i] = W.ws[i] >= 0. ? v : -v;
    }
}

__global__ __launch_bounds__(512) void cma_eig_fcols(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y;
    CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (sc->eig_stage != 1) return;
    constexpr int MMAX = 256 + 2;
    __shared__ double lamS[MMAX], gvs[MMAX];
    __shared__ int outposS[MMAX], rowmapS[MMAX], colrootS[MMAX];
    const int n = c.n, ld = c.ld, tid = threadIdx.x, T = 512;
    const size_t slab = eig_slab(ld);
    double *base = d.eig_work + (size_t) 4 * p * slab;
    double *tri = base + 3 * slab;
    const DcWork W = dc_work_layout(tri + 4 * n + 8, n + 2, nullptr);
    const int m = n, a = 0;
    const int k = W.cnt[0], nd = W.cnt[1], nr = W.cnt[2];
    // ---- all eigenvalues, their output order, the row / column maps of F (dc_merge_level) ----------
    if (tid < k) lamS[tid] = W.dl[W.org[tid]] + W.mu[tid];
    if (tid < nd) lamS[k + tid] = W.dS[W.dp[tid]];
    __syncthreads();
    if (tid < m) {
        const double key = dc_key(lamS[tid]);
        const int trips = 32 - __builtin_clz(max(m, 1));
        outposS[tid] = (nr > 0 || W.cnt[3] != 0) ? dc_rank_of(lamS, m, key, tid)
                : dc_rank_sorted2(lamS, k, nd, key, tid, trips);
    }
    __syncthreads();
    if (tid < k) {
        rowmapS[W.srcS[W.kp[tid]] - a] = tid;
        colrootS[outposS[tid]] = tid;
    }
    if (tid < nd) {
        rowmapS[W.srcS[W.dp[tid]] - a] = -(1 + outposS[k + tid]);
        colrootS[outposS[k + tid]] = -1;
    }
    __syncthreads();
    // ---- this workgroup's 32 output columns of F, 16 lanes each ---------------------------------------
    double *Fg = base + slab + (size_t) n * n;            // F: m x m, rows in ORIGINAL column order
    const int cidx = 32 * (int) blockIdx.x + (tid >> 4), sub = tid & 15;
    const int j = cidx < m ? colrootS[cidx] : -1;
    const double dorgj = j >= 0 ? W.dl[W.org[j]] : 0.;
    const double muj = j >= 0 ? W.mu[j] : 0.;
    double ss = 0.;
    if (j >= 0 && k > 1) {
        const int npl = (k + 15) >> 4;
        for (int t0 = 0; t0 < npl; t0 += 4) {
            double wh[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ic = min(sub + (t0 + u) * 16, k - 1);
                wh[u] = W.what[ic];
                dd[u] = W.dl[ic];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ii = sub + (t0 + u) * 16;
                const double sv = wh[u] * dc_rcp((dd[u] - dorgj) - muj);
                ss += ii < k ? sv * sv : 0.;
            }
        }
    }
    ss = dc_quad_sum<16>(ss);
    const double nj = (j >= 0 && k > 1) ? 1. / sqrt(ss) : 1.;
    if (cidx < m) {
        for (int r0 = sub; r0 < m; r0 += 64) {
            int iv[4];
#pragma unroll
            for (int u = 0; u < 4; u++) iv[u] = rowmapS[min(r0 + 16 * u, m - 1)];
            double wh[4], dd[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int ic = iv[u] >= 0 ? iv[u] : 0;
                wh[u] = W.what[ic];
                dd[u] = W.dl[ic];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int r = r0 + 16 * u;
                const int i = iv[u];
                const double root = k == 1 ? 1. : wh[u] * dc_rcp((dd[u] - dorgj) - muj) * nj;
                const double v = i >= 0 ? (j >= 0 ? root : 0.) : (-(1 + i) == cidx ? 1. : 0.);
                if (r < m) Fg[(size_t) r * m + cidx] = v;
            }
        }
    }
    // deflation rotations, in reverse, on this workgroup's columns: Q G with G = [[c, -s], [s, c]] on
    // sorted columns (p, j); a column's 16 lanes wrote its rows, so the lanes of the column meet first
    if (nr > 0) {
        __threadfence_block();
        __syncthreads();
        if (cidx < m && sub == 0)
            for (int r = nr - 1; r >= 0; r--) {
                const int rp = W.srcS[W.rotp[r]] - a, rj = W.srcS[W.rotj[r]] - a;
                const double cth = W.rotc[r], s = W.rots[r];
                const double x = Fg[(size_t) rp * m + cidx], y = Fg[(size_t) rj * m + cidx];
                Fg[(size_t) rp * m + cidx] = cth * x - s * y;
                Fg[(size_t) rj * m + cidx] = s * x + cth * y;
            }
    }
    if (blockIdx.x != 0) return;
    // ---- workgroup 0: what follows the merge in cma_eigen_impl ---------------------------------------
    // the scale of eig_dc_phase (a power of two from the largest |d|, |e| of the torn problem), undone
    double am = 0.;
    for (int i = tid; i < n; i += T) am = fmax(am, fmax(fabs(tri[3 * n + i]), fabs(tri[n + i])));
    am = dc_wave_max(am);
    __shared__ double reds[8];
    if ((tid & 63) == 0) reds[tid >> 6] = am;
    __syncthreads();
    am = reds[0];
#pragma unroll
    for (int w = 1; w < 8; w++) am = fmax(am, reds[w]);
    int ex = 0;
    if (am > 0.) frexp(am, &ex);
    const double inv = 1. / (am > 0. ? ldexp(1., 1 - ex) : 1.);
    if (tid < m) gvs[outposS[tid]] = lamS[tid] * inv;         // ascending
    __syncthreads();
    // repair (cmaes.cpp:250-266) and sqrt (:269-271)
    double *C = d.C + (size_t) p * ld * ld;
    const double lo = gvs[0], hi = gvs[n - 1];
    __syncthreads();
    if (lo <= 0.) {
        const double shift = fmax(hi, 0.) / 1e14;
        for (int i = tid; i < n; i += T) {
            gvs[i] = fmax(gvs[i], 0.) + shift;
            C[(size_t) i * ld + i] += shift;
        }
        __syncthreads();
    }
    const double lo2 = gvs[0], hi2 = gvs[n - 1];
    __syncthreads();
    if (hi2 > 1e14 * lo2) {
        const double shift = hi2 / 1e14 - lo2;
        for (int i = tid; i < n; i += T) {
            gvs[i] += shift;
            C[(size_t) i * ld + i] += shift;
        }
        __syncthreads();
    }
    double *Dp = d.D + (size_t) p * ld;
    for (int i = tid; i < ld; i += T) Dp[i] = i < n ? sqrt(gvs[i]) : 1.;
    if (tid == 0) {
        sc->eigenlastev = sc->fev;
        sc->eigen_done = 1;
        // (eig_stage stays 1 for the workgroups of this launch that have not started yet: the last
        // kernel of the chain, cma_eig_wy4, clears it)
    }
}

// grid (2, P), 512 threads, dynamic LDS of the n = 128 plan (plh); lda_work: row stride of the
// global work matrix the blocks go to
__global__ __launch_bounds__(512) void cma_eig_halves(CmaDev d, CmaConst c, EigPlan plh, int lda_work)
{
    const int p = blockIdx.y, h = blockIdx.x;
    const CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (sc->eig_stage != 1) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, T = 512;
    const int n = c.n, ld = c.ld;
    constexpr int nv = 130;
    double *dv = lds + 2, *ev = dv + nv, *uv = ev + nv;
    double *uh1 = uv + 6 * nv;                    // (the layout of cma_eigen_impl<512, true>)
    double *part = uh1 + nv - 2;
    double2 *rot = reinterpret_cast<double2*>(part);
    int *ibuf = reinterpret_cast<int*>(rot + 2 * plh.rc);
    double *Am = reinterpret_cast<double*>(ibuf + 2 * EIG_MAXSEQ * 3 + 8);
    const int mid = n / 2;                        // (= eig_dc_phase's edge of two blocks, mode 2)
    const int off = h ? mid : 0, m = h ? n - mid : mid;
    const size_t slab = eig_slab(ld);
    double *base = d.eig_work + (size_t) 4 * p * slab;
    double *tri = base + 3 * slab;
    if (h == 2) {
        // a third workgroup, beside the two halves: the T factors of the reflector panels depend on
        // V and the reflectors' scalars only, and took 26 us of the top-merge kernel that follows.
        // (even n: for odd n the second half's merge scratch reaches one element into tau)
        if (n & 1) return;
        double *taug = base + slab + (size_t) 2 * n * n;
        for (int i = tid; i < n; i += T) {
            const double hi = tri[2 * n + i];
            taug[i] = hi != 0. ? 1. / hi : 0.;
        }
        __threadfence_block();
        __syncthreads();
        dc_build_T(n, base + slab, taug, taug + n, uv);
        if (tid == 0) tri[4 * n] = 1.;         // (cma_eigen_g2: built)
        return;
    }
    if (tid < 2) {
        dv[-1 - tid] = 0.;
        ev[-1 - tid] = 0.;
    }
    // the half as a tridiagonal problem of its own: the rank-one tear at `mid` comes off its end
    const double tear = fabs(tri[n + mid - 1]);
    for (int i = tid; i < m; i += T) {
        double di = tri[off + i];
        if ((h == 0 && i == m - 1) || (h == 1 && i == 0)) di -= tear;
        dv[i] = di;
        ev[i] = i + 1 < m ? tri[n + off + i] : 0.;
    }
    __syncthreads();
    DcMat Qm { Am, m | 1 };
    // 2 m^2 doubles of merge scratch per half, in the room the top merge's factor F takes later
    // (behind Q_house = V: [n^2, 2 n^2) from eig_work[1]; 4 (n - mid)^2 <= n^2 + 2 n + 1 fits with
    // the slab's padding -- a fixed stride sized for n = 256 ran into eig_work[3] at n = 141..144)
    const size_t mm = (size_t) (n - mid) * (n - mid);
    double *G = base + slab + (size_t) n * n + (size_t) h * 2 * mm;
    double *Bout = base + (size_t) off * lda_work + off;
    // (diagnostic bit 2048: the phase clocks show the FIRST HALF's leaves and merges, the top merge
    // behind it keeps its hands off them)
    eig_dc_phase<512, false, false>(Qm, m, dv, ev, G, Bout, lda_work, uv,
            (d.stamps && p == 0 && h == 0 && (d.dbg & 2048)) ? d.stamps : nullptr, d.dbg, 0, nullptr,
            true, nullptr, 1);
    // the rest of this half's rows of the work matrix: the other half's columns are zero
    const int c0 = h ? 0 : mid, cw = h ? mid : n - mid;
    for (int r = tid >> 6; r < m; r += T / 64)
        for (int cc = tid & 63; cc < cw; cc += 64) base[(size_t) (off + r) * lda_work + c0 + cc] = 0.;
    for (int i = tid; i < m; i += T) tri[3 * n + off + i] = dv[i];
}

__global__ __launch_bounds__(256, 2) void cma_eigen_256(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<256, true>(d, c, pl, force);       // n <= 64: the matrix always fits LDS
}
__global__ __launch_bounds__(128, 2) void cma_eigen_128(CmaDev d, CmaConst c, EigPlan pl, int force)
{
    cma_eigen_impl<128, true>(d, c, pl, force);       // n <= 32
}

// ---------------------------------------------------------------------------
// C = A B (n x n, row-major) on the matrix cores, operands straight from global memory (L2):
// the two products of the external top merge for 128 < n <= 256, B = Q_house ((Q_1 (+) Q_2) F).
// which = 0: eig_work[0] (block-diagonal Q) * eig_work[2] (F) -> eig_work[3];
// which = 1: eig_work[1] (Q_house) * eig_work[3] -> d.B.
// grid (ceil(n/64), ceil(n/64), P), 256 threads: wavefront w owns rows 16w.. of a 64 x 64 block
// ---------------------------------------------------------------------------
// CT = column tiles per wavefront: 4 (a 64 x 64 block per workgroup) when many populations fill the
// chip, 1 (64 x 16) when a handful do not -- one population at n = 256 is 16 workgroups of CT = 4,
// each wavefront a chain of 1024 MFMAs (30 us), or 64 workgroups of CT = 1
template<int CT>
__device__ __forceinline__ void eig_gemm_body(const CmaDev &d, const CmaConst &c, int lda_work, int which)
{
    const int p = blockIdx.z;
    const CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (!sc->eigen_done) return;
    const int n = c.n, ld = c.ld;
    const size_t slab = eig_slab(ld);
    const double *base = d.eig_work + (size_t) 4 * p * slab;
    const double *A = which == 0 ? base : base + slab;
    const int lda = which == 0 ? lda_work : n;
    const double *Bm = which == 0 ? base + slab + (size_t) n * n : base + 3 * slab;
    double *Cm = which == 0 ? const_cast<double*>(base) + 3 * slab : d.B + (size_t) p * ld * ld;
    const int ldc = which == 0 ? n : ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int row = blockIdx.y * 64 + wave * 16 + fr;
    const int col0 = blockIdx.x * 16 * CT + fr;
    d4_eig acc[CT];
#pragma unroll
    for (int t = 0; t < CT; t++) acc[t] = d4_eig { 0., 0., 0., 0. };
    const int ksteps = (n + 3) >> 2;
    // which = 0: the left factor is BLOCK DIAGONAL (Q_1 on [0, n/2), Q_2 on [n/2, n), zeros
    // elsewhere): a workgroup whose 64 rows lie inside one block contracts over that block's columns
    // only (the other k-steps multiply stored zeros: x + 0 f = x, the same sums)
    int klo = 0, khi = ksteps;
    if (which == 0) {
        const int mid = n / 2, r0 = blockIdx.y * 64, r1 = min(r0 + 64, n);
        if (r1 <= mid) khi = (mid + 3) >> 2;
        else if (r0 >= mid) klo = (mid >> 2) & ~3;
    }
    for (int ks0 = klo; ks0 < khi; ks0 += 4) {
        double av[4], bv[4][CT];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int kk = 4 * (ks0 + u) + fk;
            av[u] = (row < n && kk < n) ? A[(size_t) row * lda + kk] : 0.;
#pragma unroll
            for (int t = 0; t < CT; t++) {
                const int col = col0 + 16 * t;
                bv[u][t] = (kk < n && col < n) ? Bm[(size_t) kk * n + col] : 0.;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < CT; t++)
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u][t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < CT; t++) {
        const int col = col0 + 16 * t;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int orow = blockIdx.y * 64 + wave * 16 + fk + 4 * r;
            if (orow < n && col < n) Cm[(size_t) orow * ldc + col] = acc[t][r];
        }
    }
}

__global__ __launch_bounds__(256) void cma_eig_gemm(CmaDev d, CmaConst c, int lda_work, int which)
{
    eig_gemm_body<4>(d, c, lda_work, which);
}
__global__ __launch_bounds__(256) void cma_eig_gemm1(CmaDev d, CmaConst c, int lda_work, int which)
{
    eig_gemm_body<1>(d, c, lda_work, which);       // grid (ceil(n/16), ceil(n/64), P)
}

// ---------------------------------------------------------------------------
// B = H(n-1) ... H(1) M for 128 < n <= 256, the reflectors applied in blocked (compact WY) form:
// the second product of the external top merge, M = eig_work[3] = (Q_1 (+) Q_2) F, with
// V = eig_work[1] (row i = u_i), [tau | T] behind F (written by cma_eigen / dc_build_T).
// A wavefront owns one 16-column tile of M and keeps all of it in registers (n / 16 accumulator
// tiles): W = V_b^T M (the accumulator layout of a row tile is the B-operand layout of the
// contraction over its rows), W <- T_b^T W, M -= V_b W, panel after panel, then stores B.
// The 16 x n panel is staged in LDS for the four wavefronts of the workgroup.
// grid (ceil(n / 64), P), 256 threads
// ---------------------------------------------------------------------------
constexpr int WY_LDV = 256 + 4;
__global__ __launch_bounds__(256) void cma_eig_wy(CmaDev d, CmaConst c)
{
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (!sc->eigen_done) return;
    __shared__ __attribute__((aligned(16))) double Vp[16 * WY_LDV];
    const int n = c.n, ld = c.ld;
    const size_t slab = eig_slab(ld);
    const double *base = d.eig_work + (size_t) 4 * p * slab;
    const double *V = base + slab;
    const double *tau = base + slab + (size_t) 2 * n * n;
    const double *Tg = tau + n;
    const double *M = base + 3 * slab;
    double *Bp = d.B + (size_t) p * ld * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int npanel = (n + 15) >> 4, nrt = (n + 15) >> 4;
    const int col = (blockIdx.x * 4 + wave) * 16 + fr;
    const bool live = (blockIdx.x * 4 + wave) * 16 < n;
    d4_eig q[16];
#pragma unroll
    for (int rt = 0; rt < 16; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * rt + fk + 4 * r;
            q[rt][r] = (live && row < n && col < n) ? M[(size_t) row * n + col] : 0.;
        }
    // panel b + 1 is on its way from L2 (in registers) while panel b is applied: staged straight
    // from global memory every panel exposed a round trip, 16 times per launch
    double pre[16];
    auto fetch = [&](int b) {
        const int i0 = 16 * b, reach = min(n, i0 + 16);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int k = tid;      // e = tid + 256 u: row j = u, column k = tid
            pre[u] = (i0 + u < n && k < reach) ? V[(size_t) (i0 + u) * n + k] : 0.;
        }
    };
    fetch(0);
    for (int b = 0; b < npanel; b++) {
        const int i0 = 16 * b, reach = min(n, i0 + 16);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; u++) Vp[u * WY_LDV + tid] = pre[u];
        if (b + 1 < npanel) fetch(b + 1);
        double tv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = 4 * ks + fk;
            tv[ks] = kk <= fr ? Tg[(size_t) b * 256 + kk * 16 + fr] : 0.;
        }
        __syncthreads();
        if (!live) continue;
        d4_eig w = { 0., 0., 0., 0. };
#pragma unroll
        for (int rt = 0; rt < 16; rt++) {
            if (16 * rt < reach) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    w = __builtin_amdgcn_mfma_f64_16x16x4f64(Vp[fr * WY_LDV + 16 * rt + 4 * r + fk],
                            q[rt][r], w, 0, 0, 0);
            }
        }
        d4_eig w2 = { 0., 0., 0., 0. };
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[ks], w[ks], w2, 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < 16; rt++) {
            if (16 * rt < reach) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    q[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            -Vp[(4 * ks + fk) * WY_LDV + 16 * rt + fr], w2[ks], q[rt], 0, 0, 0);
            }
        }
    }
    if (live) {
#pragma unroll
        for (int rt = 0; rt < 16; rt++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 16 * rt + fk + 4 * r;
                if (rt < nrt && row < n && col < n) Bp[(size_t) row * ld + col] = q[rt][r];
            }
    }
}

// The same product for FEW matrices (round 4): cma_eig_wy is 16 wavefronts per matrix, each a chain of
// ~1100 dependent-issue MFMAs (64 us at n = 256 on an otherwise idle chip).  Here the four
// wavefronts of a workgroup SHARE one 16-column tile: wavefront w keeps the row tiles rt = w, w + 4,
// ... (cyclic: the reflectors of panel b reach rows < 16 b + 16 only, so the work stays even),
// forms its part of W = V_b^T M, the parts are added through LDS (in wavefront order: the sum does
// not depend on timing), every wavefront forms T_b^T W itself and updates its own row tiles.
// Per panel 36 MFMAs per wavefront instead of 132, two barriers (the panels alternate between two
// LDS buffers).  grid (ceil(n / 16), P), 256 threads
// pack != 0 (lazy_isc: C^-1/2 is not formed): the sampler's packed operand B diag(D) leaves with B
// -- element (i, j) -> column tile i >> 4, k-step j >> 2, lane (j & 3, i & 15), what cma_post would
// re-read B for in a launch of its own
template<int NMAX>
__device__ __forceinline__ void eig_wy4_body(const CmaDev &d, const CmaConst &c, int pack)
{
    constexpr int NRT = NMAX / 64;          // row tiles per wavefront
    constexpr int CPT = NMAX / 256;         // columns of a staged panel per thread
    constexpr int LDV = NMAX + 4;
    const int p = blockIdx.y;
    const CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (!sc->eigen_done) return;
    __shared__ __attribute__((aligned(16))) double Vp[2][16 * LDV];
    __shared__ __attribute__((aligned(16))) double wpart[4][256];
    const int n = c.n, ld = c.ld;
    const size_t slab = eig_slab(ld);
    const double *base = d.eig_work + (size_t) 4 * p * slab;
    const double *V = base + slab;
    const double *tau = base + slab + (size_t) 2 * n * n;
    const double *Tg = tau + n;
    const double *M = base + 3 * slab;
    double *Bp = d.B + (size_t) p * ld * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const int npanel = (n + 15) >> 4;
    const int col = blockIdx.x * 16 + fr;
    // this wavefront's row tiles: rt = wave + 4 j
    d4_eig q[NRT];
#pragma unroll
    for (int j = 0; j < NRT; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * (wave + 4 * j) + fk + 4 * r;
            q[j][r] = (row < n && col < n) ? M[(size_t) row * n + col] : 0.;
        }
    double pre[16 * CPT];
    auto fetch = [&](int b) {
        const int i0 = 16 * b, reach = min(n, i0 + 16);
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int e = 0; e < CPT; e++) {
                const int k = tid + 256 * e;
                pre[CPT * u + e] = (i0 + u < n && k < reach) ? V[(size_t) (i0 + u) * n + k] : 0.;
            }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int e = 0; e < CPT; e++) Vp[buf][u * LDV + tid + 256 * e] = pre[CPT * u + e];
    };
    fetch(0);
    stage(0);
    if (npanel > 1) fetch(1);
    __syncthreads();
    for (int b = 0; b < npanel; b++) {
        const double *Vb = Vp[b & 1];
        const int reach = min(n, 16 * b + 16);
        double tv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const int kk = 4 * ks + fk;
            tv[ks] = kk <= fr ? Tg[(size_t) b * 256 + kk * 16 + fr] : 0.;
        }
        // the next panel goes to the other buffer (its last readers passed the barrier at the end of
        // panel b - 1), the one after it starts its way from L2
        if (b + 1 < npanel) {
            stage((b + 1) & 1);
            if (b + 2 < npanel) fetch(b + 2);
        }
        d4_eig w = { 0., 0., 0., 0. };
#pragma unroll
        for (int j = 0; j < NRT; j++) {
            const int rt = wave + 4 * j;
            if (16 * rt < reach) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    w = __builtin_amdgcn_mfma_f64_16x16x4f64(Vb[fr * LDV + 16 * rt + 4 * r + fk],
                            q[j][r], w, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) wpart[wave][64 * r + lane] = w[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++)
            w[r] = ((wpart[0][64 * r + lane] + wpart[1][64 * r + lane]) + wpart[2][64 * r + lane])
                    + wpart[3][64 * r + lane];
        d4_eig w2 = { 0., 0., 0., 0. };
#pragma unroll
        for (int ks = 0; ks < 4; ks++)
            w2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[ks], w[ks], w2, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NRT; j++) {
            const int rt = wave + 4 * j;
            if (16 * rt < reach) {
#pragma unroll
                for (int ks = 0; ks < 4; ks++)
                    q[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            -Vb[(4 * ks + fk) * LDV + 16 * rt + fr], w2[ks], q[j], 0, 0, 0);
            }
        }
        __syncthreads();      // wpart and this panel's buffer are free again; the next panel is staged
    }
    const double dcol = (pack && col < n) ? d.D[(size_t) p * ld + col] : 0.;
    double *BDp = d.BDp + (size_t) p * ld * ld;
    const int KS = ld >> 2;
#pragma unroll
    for (int j = 0; j < NRT; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * (wave + 4 * j) + fk + 4 * r;
            if (row < n && col < n) {
                Bp[(size_t) row * ld + col] = q[j][r];
                if (pack)
                    BDp[((size_t) (row >> 4) * KS + (col >> 2)) * 64 + ((col & 3) << 4) + (row & 15)] =
                            q[j][r] * dcol;
            }
        }
    if (pack && tid == 0 && blockIdx.x == 0) {
        d.scal[p].basis_ok = 1;
        if (pack == 2) d.scal[p].eig_stage = 0;      // (behind cma_eig_fcols, which left it to the chain's end)
    }
}

__global__ __launch_bounds__(256) void cma_eig_wy4(CmaDev d, CmaConst c, int pack)
{
    eig_wy4_body<256>(d, c, pack);
}
// 256 < n <= 512 (reflectors stashed by cma_tred_mw512 / cma_tred_tail): eight row tiles per wavefront
__global__ __launch_bounds__(256) void cma_eig_wy4_512(CmaDev d, CmaConst c)
{
    eig_wy4_body<512>(d, c, 0);
}

// ---------------------------------------------------------------------------
// n <= 16: the whole decomposition by ONE WAVEFRONT, four matrices per workgroup.  cma_eigen
// spends a 512-thread workgroup -- the whole register file of a CU -- on a matrix whatever its
// size; with thousands of small populations in flight (C1: n = 10) that is what a generation
// waits for.  Same arithmetic as the big path (unscaled reflectors, the reference's QL with its
// sign conventions as the only "leaf", reflectors applied to the tridiagonal eigenvectors,
// ascending order, repair, square roots), lanes (row j = lane & 15, part q = lane >> 4) sharing
// the O(n^2) loops, wavefront-level fences only.
// grid (ceil(P / 4)), 256 threads
// ---------------------------------------------------------------------------
constexpr int EIGS_LD = 17;
constexpr int EIGS_DOUBLES = 2 * 16 * EIGS_LD + 8 * 20 + 272 + 8;

// one wavefront, one matrix: population p, `A` = EIGS_DOUBLES doubles of LDS owned by the wavefront
__device__ __forceinline__ void eigen_small_body(const CmaDev &d, const CmaConst &c, int p, int lane,
        double *A, int force, int with_post)
{
    CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    if (!force && !((double) (sc->fev - sc->eigenlastev) > c.eigenfreq)) {
        if (lane == 0) sc->eigen_done = 0;
        return;
    }
    const int n = c.n, ld = c.ld;
    if (d.stamps && lane == 0) d.stamps[25] = wall_clock64();      // (diagnostic)
    // A: work matrix; row i ends as the reflector u_i
    double *Qs = A + 16 * EIGS_LD;              // eigenvectors of T, then B
    double *dv = Qs + 16 * EIGS_LD + 2;         // (front pads: the QL producer prefetches index -1)
    double *ev = dv + 20, *uv = ev + 20, *wv = uv + 20, *hv = wv + 20, *td = hv + 20, *gv = td + 20;
    int *perm = reinterpret_cast<int*>(gv + 20);
    double *ws = gv + 40;
    double *C = d.C + (size_t) p * ld * ld;
    const int j = lane & 15, q = lane >> 4;
    // symmetrise from the lower triangle (cmaes.cpp:238-242)
    for (int x = lane; x < 16 * 16; x += 64) {
        const int r = x >> 4, k = x & 15;
        A[r * EIGS_LD + k] = (r < n && k < n) ? (k <= r ? C[(size_t) r * ld + k] : C[(size_t) k * ld + r])
                                               : 0.;
    }
    if (lane < 20) {
        dv[lane - 2] = 0.; ev[lane - 2] = 0.; uv[lane - 2] = 0.; wv[lane - 2] = 0.; hv[lane - 2] = 0.;
    }
    dc_wave_sync();
    // ---- Householder tridiagonalisation (cmaes.cpp:285-381; reflectors left unscaled) --------
    for (int i = n - 1; i > 0; i--) {
        const double dk = lane < i ? A[i * EIGS_LD + lane] : 0.;
        const double h0 = eig_wave_sum(dk * dk);
        const double f = A[i * EIGS_LD + i - 1];
        if (h0 == 0.) {
            if (lane == 0) {
                ev[i] = f;
                hv[i] = 0.;
            }
            dc_wave_sync();
            continue;
        }
        double g = sqrt(h0);
        if (f > 0) g = -g;
        const double h = h0 - f * g;
        if (lane < 16) uv[lane] = lane < i ? (lane == i - 1 ? f - g : dk) : 0.;
        if (lane == 0) ev[i] = g;
        dc_wave_sync();
        // p = A u / h over the block [0, i)^2, w = p - (u^T p / 2h) u
        double acc = 0.;
        if (j < i)
            for (int k = q; k < i; k += 4) acc = fma(A[j * EIGS_LD + k], uv[k], acc);
        acc += __shfl_xor(acc, 16, 64);
        acc += __shfl_xor(acc, 32, 64);
        const double pj = acc / h;
        const double hh = eig_wave_sum((lane < 16 && lane < i) ? pj * uv[lane] : 0.) / (h + h);
        if (lane < 16) wv[lane] = lane < i ? pj - hh * uv[lane] : 0.;
        dc_wave_sync();
        if (j < i) {
            const double uj = uv[j], wj = wv[j];
            for (int k = q; k < i; k += 4)
                A[j * EIGS_LD + k] -= uj * wv[k] + wj * uv[k];
        }
        dc_wave_sync();
        if (lane < i) A[i * EIGS_LD + lane] = uv[lane];        // stash: row i = u_i
        if (lane == 0) hv[i] = h;
        dc_wave_sync();
    }
    if (lane < 16) td[lane] = lane < n ? A[lane * EIGS_LD + lane] : 0.;
    dc_wave_sync();
    // tql2's prologue: the sub-diagonal shifted down (cmaes.cpp:384-387)
    {
        const double t = (lane + 1 < n && lane < 16) ? ev[lane + 1] : 0.;
        dc_wave_sync();
        if (lane < 16) ev[lane] = t;
    }
    dc_wave_sync();
    // ---- the reference's implicit QL on (td, ev) ----------------------------------------------
    DcMat Qm { Qs, EIGS_LD };
    if (d.stamps && lane == 0) d.stamps[26] = wall_clock64();
    dc_leaf_ql(Qm, 0, n, td, ev, dv, ws, lane, nullptr);
    if (d.stamps && lane == 0) d.stamps[27] = wall_clock64();
    // ---- B = H(n-1) ... H(1) Q_T: column j, rows k = q (mod 4) -----------------------------------
    for (int i = 1; i < n; i++) {
        const double h = hv[i];
        if (h != 0.) {
            double acc = 0.;
            if (j < n)
                for (int k = q; k < i; k += 4) acc = fma(A[i * EIGS_LD + k], Qs[k * EIGS_LD + j], acc);
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            const double gq = -(acc / h);
            dc_wave_sync();
            if (j < n)
                for (int k = q; k < i; k += 4)
                    Qs[k * EIGS_LD + j] = fma(gq, A[i * EIGS_LD + k], Qs[k * EIGS_LD + j]);
            dc_wave_sync();
        }
    }
    // ---- ascending order (cmaes.cpp:459-477), repair (:250-266), sqrt (:269-271) ----------------
    if (lane < n) {
        const double dj = dv[lane], kj = dc_key(dj);
        int r = 0;
        for (int k = 0; k < n; k++) {
            const double dk = dc_key(dv[k]);
            r += (dk < kj) || (dk == kj && k < lane);
        }
        perm[lane] = r;
        gv[r] = dj;
    }
    dc_wave_sync();
    const double lo = gv[0], hi = gv[n - 1];
    dc_wave_sync();
    if (lo <= 0.) {
        const double shift = fmax(hi, 0.) / 1e14;
        if (lane < n) {
            gv[lane] = fmax(gv[lane], 0.) + shift;
            C[(size_t) lane * ld + lane] += shift;
        }
        dc_wave_sync();
    }
    const double lo2 = gv[0], hi2 = gv[n - 1];
    dc_wave_sync();
    if (hi2 > 1e14 * lo2) {
        const double shift = hi2 / 1e14 - lo2;
        if (lane < n) {
            gv[lane] += shift;
            C[(size_t) lane * ld + lane] += shift;
        }
        dc_wave_sync();
    }
    double *Dp = d.D + (size_t) p * ld;
    double *Bp = d.B + (size_t) p * ld * ld;
    for (int i = lane; i < ld; i += 64) Dp[i] = i < n ? sqrt(gv[i]) : 1.;
    for (int x = lane; x < n * n; x += 64) {
        const int k = x / n, jj = x - k * n;
        Bp[(size_t) k * ld + perm[jj]] = Qs[k * EIGS_LD + jj];
    }
    if (lane == 0) {
        sc->eigenlastev = sc->fev;
        sc->eigen_done = 1;
    }
    if (!with_post) return;
    // ---- what cma_post does after a decomposition, by the same wavefront: C^-1/2 = B D^-1 B^T
    // term by term in the reference's order (cmaes.cpp:274-282), the packed MFMA operands of the
    // sampler (B D) and of the whitening GEMM (C^-1/2); ld = 16 here.  B in sorted column order
    // goes to the work matrix (the reflectors are spent), D to dv.
    dc_wave_sync();
    for (int x = lane; x < 16 * 16; x += 64) {
        const int k = x >> 4, jj = x & 15;
        if (k < n && jj < n) A[k * EIGS_LD + perm[jj]] = Qs[k * EIGS_LD + jj];
    }
    if (lane < 16) dv[lane] = lane < n ? sqrt(gv[lane]) : 1.;
    dc_wave_sync();
    double *isc = d.isc + (size_t) p * ld * ld;
    double *ISp = d.ISp + (size_t) p * ld * ld, *BDp = d.BDp + (size_t) p * ld * ld;
    const int KS = ld >> 2;
    for (int x = lane; x < ld * ld; x += 64) {
        const int i = x / ld, jc = x - i * ld;
        double v = 0.;
        if (i < n && jc < n)
            for (int k = 0; k < n; k++) v += A[i * EIGS_LD + k] / dv[k] * A[jc * EIGS_LD + k];
        isc[(size_t) i * ld + jc] = v;
        ISp[((size_t) (i >> 4) * KS + (jc >> 2)) * 64 + ((jc & 3) << 4) + (i & 15)] = v;
    }
    for (int qq = lane; qq < ld * ld; qq += 64) {
        const int t4 = qq >> 6, l = qq & 63;             // t4 = nt * KS + ks
        const int nt = t4 / KS, ks = t4 - nt * KS;
        const int i = nt * 16 + (l & 15), jc = 4 * ks + (l >> 4);
        BDp[qq] = (i < n && jc < n) ? A[i * EIGS_LD + jc] * dv[jc] : 0.;
    }
    if (lane == 0) sc->basis_ok = 1;
    if (d.stamps && lane == 0) d.stamps[28] = wall_clock64();
}

__global__ __launch_bounds__(256) void cma_eigen_small(CmaDev d, CmaConst c, int force, int with_post)
{
    __shared__ __attribute__((aligned(16))) double lds_all[4][EIGS_DOUBLES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + wave;
    if (p >= c.npop) return;
    eigen_small_body(d, c, p, lane, lds_all[wave], force, with_post);
}

} // namespace bbo
