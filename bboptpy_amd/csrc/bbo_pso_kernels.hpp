// bbo_pso_kernels.hpp -- one APSO generation as gfx950 kernels.
//
//   kernel          reference lines it replaces                            bound
//   pso_init        apso.cpp:70-97 (uniform swarm, v = 0, pbest = x)        HBM
//   pso_center/mean/nrm   centring for the distance kernel                  HBM (8n B/particle)
//   pso_ese_sym     getf, apso.cpp:300-339: d_i = mean_j ||x_i - x_j||      fp64 MFMA, n np flop/particle
//   pso_control_a/b nextState/mu/updatec1c2/updateElitist :200-298,:347-452 latency (1 WG)
//   pso_update      updateParticle :159-198, fused with the objective       HBM: 3 row reads + 2 row
//                                                                           writes = 40n+16 B
//   pso_finish      gbest arg-min + converged() :129-145                    latency (1 WG)
#pragma once

#include "bbo_pso.hpp"
#include "bbo_objectives.hpp"
#include "bbo_rng.hpp"

namespace bbo {

typedef double pso_d4 __attribute__((ext_vector_type(4)));
#define PSO_INF (__builtin_huge_val())

__device__ inline bool pso_frozen(const PsoConst &c, const PsoScal *sc)
{
    return c.honor_stop && sc->stop != 0;
}

template<int G>
__device__ inline double pso_group_sum(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
    return v;
}

// block-wide (256 threads) reductions through a small LDS scratch
__device__ inline double pso_block_sum(double v, double *scratch)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// arg-min (SIGN = +1) or arg-max (SIGN = -1) with first-index tie-break, 256 threads
template<int SIGN>
__device__ inline void pso_block_arg(double &v, int &idx, double *sval, int *sidx)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        const bool take = SIGN > 0 ? (ov < v || (ov == v && oi < idx))
                                   : (ov > v || (ov == v && oi < idx));
        if (take) {
            v = ov;
            idx = oi;
        }
    }
    __syncthreads();
    if ((tid & 63) == 0) {
        sval[tid >> 6] = v;
        sidx[tid >> 6] = idx;
    }
    __syncthreads();
    v = sval[0];
    idx = sidx[0];
    for (int w = 1; w < 4; w++) {
        const bool take = SIGN > 0 ? (sval[w] < v || (sval[w] == v && sidx[w] < idx))
                                   : (sval[w] > v || (sval[w] == v && sidx[w] < idx));
        if (take) {
            v = sval[w];
            idx = sidx[w];
        }
    }
}

// ---------------------------------------------------------------------------
// grid (ceil(np/16), P), 256 threads, LDS 16 * ld doubles
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pso_init(PsoDev d, PsoConst c)
{
    const int p = blockIdx.y;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * (blockDim.x >> 4) + r, ld = c.ld;
    double *row = lds + r * ld;
    const size_t base = ((size_t) p * c.np + i) * ld;
    double ssq = 0.;
    if (i < c.np) {
        for (int pj = g; pj < ld / 2; pj += 16) {
            const u32x4 w = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) pj, 0,
                    stream_word(STREAM_INIT, (uint32_t) p));
            const int j = 2 * pj;
            double2 v = make_double2(0., 0.);
            if (j < c.n) v.x = u01(w.x, w.y) * (d.upper[j] - d.lower[j]) + d.lower[j];
            if (j + 1 < c.n) v.y = u01(w.z, w.w) * (d.upper[j + 1] - d.lower[j + 1]) + d.lower[j + 1];
            *reinterpret_cast<double2*>(&row[j]) = v;
            *reinterpret_cast<double2*>(&d.X[base + j]) = v;
            *reinterpret_cast<double2*>(&d.XB[base + j]) = v;
            *reinterpret_cast<double2*>(&d.V[base + j]) = make_double2(0., 0.);
            ssq += v.x * v.x + v.y * v.y;
        }
    }
    __syncthreads();
    ssq = pso_group_sum<16>(ssq);
    if (c.obj >= 0) {
        double f = eval_row_group<16>(c.obj, c.n, row, d.aux, g);
        if (g == 0 && i < c.np) {
            if (f != f) f = PSO_INF;
            d.f[(size_t) p * c.np + i] = f;
            d.fb[(size_t) p * c.np + i] = f;
        }
    }
    if (g == 0 && i < c.np) d.radius[(size_t) p * c.np + i] = sqrt(ssq);
}

// first arg-min of f -> xbest / fbest (init) ; one workgroup of 256 per population
__global__ __launch_bounds__(256) void pso_gbest_init(PsoDev d, PsoConst c)
{
    const int p = blockIdx.x;
    __shared__ double sval[4];
    __shared__ int sidx[4];
    const int tid = threadIdx.x;
    const double *f = d.f + (size_t) p * c.np;
    double v = PSO_INF;
    int idx = 0x7fffffff;
    for (int i = tid; i < c.np; i += 256)
        if (f[i] < v) {
            v = f[i];
            idx = i;
        }
    pso_block_arg<1>(v, idx, sval, sidx);
    if (idx == 0x7fffffff) idx = 0;
    for (int j = tid; j < c.ld; j += 256)
        d.xbest[(size_t) p * c.ld + j] = d.X[((size_t) p * c.np + idx) * c.ld + j];
    if (tid == 0) {
        d.scal[p].fbest = v;
        // pbest of a host-evaluated swarm is its first fitness
    }
}

__global__ __launch_bounds__(256) void pso_copy_fb(PsoDev d, PsoConst c)
{
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < c.np) d.fb[(size_t) p * c.np + i] = d.f[(size_t) p * c.np + i];
}

// ---------------------------------------------------------------------------
// centroid: column partial sums (grid (parts, P)), then the mean (grid P), then the
// squared norms of the centred particles (grid (ceil(np/16), P))
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pso_center(PsoDev d, PsoConst c, int parts)
{
    const int p = blockIdx.y, part = blockIdx.x;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    const int rows_per = (c.np + parts - 1) / parts;
    const int r0 = part * rows_per, r1 = min(c.np, r0 + rows_per);
    for (int j = threadIdx.x; j < c.ld; j += 256) {
        double s = 0.;
        int i = r0;
        for (; i + 8 <= r1; i += 8) {          // eight independent row reads in flight
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = d.X[((size_t) p * c.np + i + u) * c.ld + j];
#pragma unroll
            for (int u = 0; u < 8; u++) s += x[u];
        }
        for (; i < r1; i++) s += d.X[((size_t) p * c.np + i) * c.ld + j];
        d.colpart[((size_t) p * parts + part) * c.ld + j] = s;
    }
}

__global__ __launch_bounds__(256) void pso_mean(PsoDev d, PsoConst c, int parts)
{
    const int p = blockIdx.x;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    for (int j = threadIdx.x; j < c.ld; j += 256) {
        double s = 0.;
        int q = 0;
        for (; q + 8 <= parts; q += 8) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; u++) x[u] = d.colpart[((size_t) p * parts + q + u) * c.ld + j];
#pragma unroll
            for (int u = 0; u < 8; u++) s += x[u];
        }
        for (; q < parts; q++) s += d.colpart[((size_t) p * parts + q) * c.ld + j];
        d.mean[(size_t) p * c.ld + j] = s / c.np;
    }
}

__global__ __launch_bounds__(256) void pso_nrm(PsoDev d, PsoConst c)
{
    const int p = blockIdx.y;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = blockIdx.x * 16 + r;
    double s = 0.;
    if (i < c.np) {
        // the centred particle also goes to Xc (row stride ldc, a multiple of 16; rows up to the
        // next multiple of 128 and columns n..ldc-1 stay zero from the allocation): pso_ese_sym
        // then stages its operands without arithmetic, clamps or masks
        double *xc = d.Xc + ((size_t) p * c.npad + i) * c.ldc;
        for (int j = g; j < c.n; j += 16) {
            const double v = d.X[((size_t) p * c.np + i) * c.ld + j] - d.mean[(size_t) p * c.ld + j];
            xc[j] = v;
            s += v * v;
        }
    }
    s = pso_group_sum<16>(s);
    if (g == 0 && i < c.np) d.nrm[(size_t) p * c.np + i] = s;
}

// ---------------------------------------------------------------------------
// mean distance of every particle to the others (getf, apso.cpp:300-339):
// G = Xi' Xj'^T on v_mfma_f64_16x16x4_f64 over the CENTRED swarm, d_ij = sqrt(max(0, |xi'|^2 +
// |xj'|^2 - 2 G_ij)), using d_ij = d_ji: workgroup I owns the 128-row block I and sweeps the
// blocks J >= I only.  A 128 x 128 tile of the Gram matrix lives in the accumulators of four
// wavefronts (2 x 2, 64 x 64 each: 8 LDS fragment reads feed 16 MFMAs), the operands stream
// through LDS in 16-column chunks, double buffered, the next chunk's global loads in flight
// during the sweep.  The operands come from Xc, the centred and zero-padded copy pso_nrm wrote:
// staging a chunk is eight 16-byte loads and eight 16-byte LDS stores per thread and NO vector
// arithmetic (the fp64 matrix instruction shares the vector pipe; round 2
// centred while staging: four dependent loads of the mean, 16 subtractions and 32 selects per
// chunk, paid once per PAIR of blocks).  Row sums stay in registers for the whole sweep; for
// J > I the tile's COLUMN sums are the contribution of block I to the particles of block J and
// go to colpart2[I][j] (no atomics: pso_ese_finish adds the slabs in a fixed order, so the
// result is reproducible).
// grid (npad/128, P), 256 threads, dynamic LDS 2*2*128*18+640 doubles
// ---------------------------------------------------------------------------
constexpr int ESE2_KC = 16, ESE2_LT = ESE2_KC + 2, ESE2_TILE = 128 * ESE2_LT;
constexpr int ESE2_LDS_DOUBLES = 4 * ESE2_TILE + 640;

__device__ __forceinline__ double ese_root(double t2)
{
    // sqrt from the hardware reciprocal-root estimate + one third-order correction (full fp64
    // to a rounding error; 2.1e9 roots per generation at np = 65536 make the IEEE sequence 5 %
    // of the kernel)
    double y = __builtin_amdgcn_rsq(t2);
    const double err = fma(-t2 * y, y, 1.);
    y = fma(y * err, fma(err, 0.375, 0.5), y);
    return t2 > 0. ? t2 * y : 0.;
}

__global__ __launch_bounds__(256, 2) void pso_ese_sym(PsoDev d, PsoConst c)
{
    const int p = blockIdx.y, I = blockIdx.x;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double *cs = lds + 4 * ESE2_TILE;        // [2][128] column sums / [2][128] row sums
    double *nI = cs + 256;                   // [128] squared norms of block I
    double *rs = nI + 128;                   // [2][128] row sums so far (a lane owns its slots)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int np = c.np, ldc = c.ldc;
    const int NB = c.npad >> 7, NCH = ldc / ESE2_KC;
    const double *Xc = d.Xc + (size_t) p * c.npad * ldc;
    const double *nrm = d.nrm + (size_t) p * np;
    const int fr = lane & 15, fk = lane >> 4;
    const int sr = tid >> 1, sh = (tid & 1) * 8;     // staging: row, first of 8 columns
    const double *ra = Xc + (size_t) (I * 128 + sr) * ldc + sh;

    // (eight named registers, not arrays: a plain copy loop from a private array to LDS is
    // turned into a memcpy from scratch memory)
    double2 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;
    auto fetch = [&](int J, int ch) {
        const double2 *a = reinterpret_cast<const double2*>(ra + ch * ESE2_KC);
        const double2 *b = reinterpret_cast<const double2*>(Xc + (size_t) (J * 128 + sr) * ldc + sh
                + ch * ESE2_KC);
        pa0 = a[0]; pa1 = a[1]; pa2 = a[2]; pa3 = a[3];
        pb0 = b[0]; pb1 = b[1]; pb2 = b[2]; pb3 = b[3];
    };
    auto stash = [&](int buf) {
        double2 *A = reinterpret_cast<double2*>(lds + buf * 2 * ESE2_TILE + sr * ESE2_LT + sh);
        double2 *B = reinterpret_cast<double2*>(lds + buf * 2 * ESE2_TILE + ESE2_TILE + sr * ESE2_LT + sh);
        A[0] = pa0; A[1] = pa1; A[2] = pa2; A[3] = pa3;
        B[0] = pb0; B[1] = pb1; B[2] = pb2; B[3] = pb3;
    };

    // (row sums: per tile in registers, then folded over the 16 column lanes and added to the
    // lane's own LDS slot -- 32 registers the sweep needs for its operand fragments)
    pso_d4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; rt++)
#pragma unroll
        for (int ct = 0; ct < 4; ct++) acc[rt][ct] = pso_d4 { 0., 0., 0., 0. };

    // Balanced cover of the unordered block pairs: block I takes the next `half` blocks
    // cyclically (for even NB the opposite pair {I, I + NB/2} belongs to the smaller index), so
    // every workgroup sweeps the same number of tiles.
    const int half = NB >> 1;
    const int cnt = 1 + ((NB & 1) || I < half ? half : half - 1);
    const int total = cnt * NCH;
    if (tid < 128) nI[tid] = I * 128 + tid < np ? nrm[I * 128 + tid] : 0.;
    rs[tid] = 0.;
    fetch(I, 0);
    stash(0);
    __syncthreads();
    int t = 0, ch = 0, J = I;
    for (int it = 0; it < total; it++) {
        const int buf = it & 1;
        int nt = t, nch = ch + 1, nJ = J;
        if (nch == NCH) {
            nch = 0;
            nt = t + 1;
            nJ = I + nt < NB ? I + nt : I + nt - NB;
        }
        // (unconditional: past the last chunk this fetches chunk 0 of the last block again and
        // stores it where nobody reads it -- with the pair under a test the compiler cannot tell
        // that the loads of one iteration are consumed in the same one, and waits for them
        // before the sweep instead of after it)
        if (nt == cnt) {
            nt = t;
            nJ = J;
        }
        fetch(nJ, nch);
        const bool last = ch == NCH - 1;
        const double *A = lds + buf * 2 * ESE2_TILE + (64 * wr + fr) * ESE2_LT + fk;
        const double *B = lds + buf * 2 * ESE2_TILE + ESE2_TILE + (64 * wc + fr) * ESE2_LT + fk;
#pragma unroll
        for (int ks = 0; ks < ESE2_KC / 4; ks++) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                a[q] = A[16 * q * ESE2_LT + 4 * ks];
                b[q] = B[16 * q * ESE2_LT + 4 * ks];
            }
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int ct = 0; ct < 4; ct++)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[rt], b[ct], acc[rt][ct],
                            0, 0, 0);
        }
        if (last) {
            // tile (I, J) complete: distances, row sums, and (J != I) column sums
            double colsum[4] = { 0., 0., 0., 0. };
            double rowacc[4][4];
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int r = 0; r < 4; r++) rowacc[rt][r] = 0.;
            double nj[4];
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                const int gj = J * 128 + 64 * wc + 16 * ct + fr;
                nj[ct] = gj < np ? nrm[gj] : 0.;
            }
            const bool full = J != I && I * 128 + 128 <= np && J * 128 + 128 <= np;
            if (full) {
                // no particle of the tile is past the swarm or on the diagonal: no tests
#pragma unroll
                for (int rt = 0; rt < 4; rt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const double ni = nI[64 * wr + 16 * rt + fk + 4 * r];
#pragma unroll
                        for (int ct = 0; ct < 4; ct++) {
                            const double dist = ese_root(fmax(ni + nj[ct] - 2. * acc[rt][ct][r], 0.));
                            rowacc[rt][r] += dist;
                            colsum[ct] += dist;
                        }
                    }
            } else {
                // (the bases pass through an empty asm: the 32 row / column indices are then
                // formed here and not hoisted out of the sweep into 32 registers it has not got)
                int gi0 = I * 128 + 64 * wr + fk, gj0 = J * 128 + 64 * wc + fr;
                asm volatile("" : "+v"(gi0), "+v"(gj0));
#pragma unroll
                for (int rt = 0; rt < 4; rt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int gi = gi0 + 16 * rt + 4 * r;
                        const double ni = nI[64 * wr + 16 * rt + fk + 4 * r];
#pragma unroll
                        for (int ct = 0; ct < 4; ct++) {
                            const int gj = gj0 + 16 * ct;
                            double dist = 0.;
                            if (gi < np && gj < np && gi != gj)
                                dist = ese_root(fmax(ni + nj[ct] - 2. * acc[rt][ct][r], 0.));
                            rowacc[rt][r] += dist;
                            colsum[ct] += dist;
                        }
                    }
            }
#pragma unroll
            for (int rt = 0; rt < 4; rt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const double s = pso_group_sum<16>(rowacc[rt][r]);
                    if (fr == 0) rs[wc * 128 + 64 * wr + 16 * rt + fk + 4 * r] += s;
                }
#pragma unroll
            for (int ct = 0; ct < 4; ct++) {
                double csum = colsum[ct];
                csum += __shfl_xor(csum, 16, 64);
                csum += __shfl_xor(csum, 32, 64);
                colsum[ct] = csum;
                acc[0][ct] = acc[1][ct] = acc[2][ct] = acc[3][ct] = pso_d4 { 0., 0., 0., 0. };
            }
            if (t > 0) {
                if (fk == 0) {
#pragma unroll
                    for (int ct = 0; ct < 4; ct++) cs[wr * 128 + 64 * wc + 16 * ct + fr] = colsum[ct];
                }
                __syncthreads();
                if (tid < 128 && J * 128 + tid < np)
                    d.colpart2[((size_t) p * NB + I) * np + J * 128 + tid] = cs[tid] + cs[128 + tid];
            }
        }
        stash(buf ^ 1);
        __syncthreads();
        t = nt;
        ch = nch;
        J = nJ;
    }
    // row sums of block I over all J >= I: the two column-half wavefronts' slots (the loop's
    // last barrier is behind their last update)
    if (tid < 128 && I * 128 + tid < np)
        d.rowpart2[(size_t) p * np + I * 128 + tid] = rs[tid] + rs[128 + tid];
}

// ws_i = (row sums of i's own block + the column contributions of every earlier block) / (np-1)
__global__ __launch_bounds__(256) void pso_ese_finish(PsoDev d, PsoConst c)
{
    const int p = blockIdx.y;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    const int i = blockIdx.x * 256 + threadIdx.x, np = c.np;
    if (i >= np) return;
    const int NB = (np + 127) >> 7, Jb = i >> 7, half = NB >> 1;
    double s = d.rowpart2[(size_t) p * np + i];
    // the blocks that swept block Jb as one of their partners, in a fixed order
    for (int t = 1; t <= half; t++) {
        const int I = Jb - t >= 0 ? Jb - t : Jb - t + NB;
        if ((NB & 1) || t < half || I < half) s += d.colpart2[((size_t) p * NB + I) * np + i];
    }
    d.ws[(size_t) p * np + i] = s / (np - 1.);
}

// ---------------------------------------------------------------------------
// fuzzy memberships and rule table (apso.cpp:347-452)
// ---------------------------------------------------------------------------
__device__ inline double pso_mu(double f, int i)
{
    switch (i) {
    case 1:
        if (f >= 0. && f <= 0.4) return 0.;
        if (f > 0.4 && f <= 0.6) return 5. * f - 2.;
        if (f > 0.6 && f <= 0.7) return 1.;
        if (f > 0.7 && f <= 0.8) return -10. * f + 8.;
        return 0.;
    case 2:
        if (f >= 0. && f <= 0.2) return 0.;
        if (f > 0.2 && f <= 0.3) return 10. * f - 2.;
        if (f > 0.3 && f <= 0.4) return 1.;
        if (f > 0.4 && f <= 0.6) return -5. * f + 3.;
        return 0.;
    case 3:
        if (f >= 0. && f <= 0.1) return 1.;
        if (f > 0.1 && f <= 0.3) return -5. * f + 1.5;
        return 0.;
    default:
        if (f >= 0. && f <= 0.7) return 0.;
        if (f > 0.7 && f <= 0.9) return 5. * f - 3.5;
        return 1.;
    }
}

// The reference indexes its rule table [r][state] with state in 1..4 on rows of four
// entries (apso.cpp:384, apso.h:48-56): one column too far, and past the row for state 4.
// The fifth column below makes that read explicit and zero.
__device__ inline int pso_next_state(double f, int state)
{
    const int rule[7][5] = { { 1, 1, 1, 1, 0 }, { 2, 2, 2, 2, 0 }, { 3, 3, 3, 3, 0 },
            { 4, 4, 4, 4, 0 }, { 1, 2, 2, 1, 0 }, { 2, 2, 3, 3, 0 }, { 1, 1, 4, 4, 0 } };
    const double m1 = pso_mu(f, 1), m2 = pso_mu(f, 2), m3 = pso_mu(f, 3), m4 = pso_mu(f, 4);
    if (state == 0) {
        const double m[4] = { m1, m2, m3, m4 };
        int arg = 0;
        for (int i = 1; i < 4; i++)
            if (m[arg] < m[i]) arg = i;
        return 1 + arg;
    }
    int r;
    if (m1 > 0 && m2 > 0) r = 4;
    else if (m2 > 0 && m3 > 0) r = 5;
    else if (m1 > 0 && m4 > 0) r = 6;
    else if (m1 > 0) r = 0;
    else if (m2 > 0) r = 1;
    else if (m3 > 0) r = 2;
    else if (m4 > 0) r = 3;
    else return -1;
    return rule[r][state];
}

// control part A: evolutionary factor scalars, state, w, c1, c2, elitist candidate
// one workgroup of 256 threads per population
__global__ __launch_bounds__(256) void pso_control_a(PsoDev d, PsoConst c)
{
    const int p = blockIdx.x;
    PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    __shared__ double sval[4];
    __shared__ int sidx[4];
    const int tid = threadIdx.x, np = c.np;
    const double *ws = d.ws + (size_t) p * np, *f = d.f + (size_t) p * np;
    double lo = PSO_INF, hi = -PSO_INF, fv = PSO_INF;
    int ilo = 0x7fffffff, ihi = 0x7fffffff, ib = 0x7fffffff;
    for (int i = tid; i < np; i += 256) {
        if (ws[i] < lo) { lo = ws[i]; ilo = i; }
        if (ws[i] > hi) { hi = ws[i]; ihi = i; }
        if (f[i] < fv) { fv = f[i]; ib = i; }
    }
    pso_block_arg<1>(lo, ilo, sval, sidx);
    pso_block_arg<-1>(hi, ihi, sval, sidx);
    pso_block_arg<1>(fv, ib, sval, sidx);
    if (ib == 0x7fffffff) ib = 0;
    for (int j = tid; j < c.ld; j += 256)
        d.pvec[(size_t) p * c.ld + j] = d.xbest[(size_t) p * c.ld + j];
    __syncthreads();
    if (tid == 0) {
        const double evof = hi <= lo ? 1. : (ws[ib] - lo) / (hi - lo);
        const int ns = pso_next_state(evof, sc->state);
        sc->evof = evof;
        sc->ibest_cur = ib;
        sc->need_elite = 0;
        if (ns < 0) {
            sc->bad_rule = 1;
            sc->stop = 3;
        } else {
            // updatec1c2, apso.cpp:248-298
            const uint32_t sw = stream_word(STREAM_PSO_CTRL, (uint32_t) p);
            u32x4 w0 = philox4x32_10(c.seed, 0, 0, (uint32_t) sc->it, sw);
            u32x4 w1 = philox4x32_10(c.seed, 1, 0, (uint32_t) sc->it, sw);
            const double delta1 = u01(w0.x, w0.y) * (0.1 - 0.05) + 0.05;
            const double delta2 = u01(w1.x, w1.y) * (0.1 - 0.05) + 0.05;
            double c1 = sc->c1, c2 = sc->c2;
            sc->w = 1. / (1. + 1.5 * exp(-2.6 * evof));
            switch (ns) {
            case 1: c1 += delta1; c2 -= delta2; break;
            case 2: c1 += 0.5 * delta1; c2 -= 0.5 * delta2; break;
            case 3: c1 += 0.5 * delta1; c2 += 0.5 * delta2; sc->need_elite = 1; break;
            default: c1 -= 0.5 * delta1; c2 += 0.5 * delta2; break;
            }
            c1 = fmax(1.5, fmin(c1, 2.5));
            c2 = fmax(1.5, fmin(c2, 2.5));
            if (c1 + c2 > 4.) {
                const double fac = 4. / (c1 + c2);
                c1 *= fac;
                c2 *= fac;
            }
            sc->c1 = c1;
            sc->c2 = c2;
            if (ns == 3) {
                // updateElitist, apso.cpp:203-209: perturb one coordinate of gbest
                const u32x4 w2 = philox4x32_10(c.seed, 2, 0, (uint32_t) sc->it, sw);
                const int dd = uint_below(w2.x, c.n);
                double z0, z1;
                normal_pair(c.seed, 3, 0, (uint32_t) sc->it, sw, z0, z1);
                const double sigma = 1.0 - (1.0 - 0.1) * sc->it / sc->maxit;
                double v = d.pvec[(size_t) p * c.ld + dd] + (d.upper[dd] - d.lower[dd]) * z0 * sigma;
                if (c.correct) v = fmax(d.lower[dd], fmin(v, d.upper[dd]));
                d.pvec[(size_t) p * c.ld + dd] = v;
            }
            sc->state = ns;
        }
    }
}

// control part B: evaluate the elitist candidate (device objective) and place it
__global__ __launch_bounds__(256) void pso_control_b(PsoDev d, PsoConst c)
{
    const int p = blockIdx.x;
    PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    if (!sc->need_elite) return;
    __shared__ double sval[4];
    __shared__ int sidx[4];
    __shared__ double snu;
    const int tid = threadIdx.x, np = c.np, ld = c.ld;
    const double *pv = d.pvec + (size_t) p * ld;
    if (c.obj >= 0) {
        if (tid < 64) {
            double v = eval_row_group<64>(c.obj, c.n, pv, d.aux, tid);
            if (v != v) v = PSO_INF;
            if (tid == 0) snu = v;
        }
    } else if (tid == 0) {
        snu = sc->nu;
    }
    __syncthreads();
    const double nu = snu;
    double *f = d.f + (size_t) p * np;
    if (nu < sc->fbest) {
        for (int j = tid; j < ld; j += 256) d.xbest[(size_t) p * ld + j] = pv[j];
        __syncthreads();
        if (tid == 0) {
            sc->fbest = nu;
            sc->nu = nu;
            sc->fev += 1;
        }
        return;
    }
    // replace the worst CURRENT particle (first maximum), apso.cpp:220-232
    double hv = -PSO_INF;
    int ih = 0x7fffffff;
    for (int i = tid; i < np; i += 256)
        if (f[i] > hv) {
            hv = f[i];
            ih = i;
        }
    pso_block_arg<-1>(hv, ih, sval, sidx);
    if (ih == 0x7fffffff) ih = 0;
    const size_t row = ((size_t) p * np + ih) * ld;
    const bool better = nu < d.fb[(size_t) p * np + ih];
    double ssq = 0.;
    for (int j = tid; j < ld; j += 256) {
        d.X[row + j] = pv[j];
        if (better) d.XB[row + j] = pv[j];
        ssq += pv[j] * pv[j];
    }
    ssq = pso_block_sum(ssq, sval);
    if (tid == 0) {
        f[ih] = nu;
        if (better) d.fb[(size_t) p * np + ih] = nu;
        d.radius[(size_t) p * np + ih] = sqrt(ssq);
        sc->nu = nu;
        sc->fev += 1;
    }
}

// ---------------------------------------------------------------------------
// fused particle update of the particles [i0, i1).  grid (ceil((i1 - i0) / R), P), 16 R threads,
// LDS R * ld doubles.  A generation is a handful of such launches with the swarm's best refreshed
// in between (pso_gbest): the reference refreshes it inside its particle loop (apso.cpp:194-197),
// and at np in the thousands a swarm that only sees the best of the generation START converges
// measurably slower ("APSO: chunked refresh"; tests/test_pop_bands_gpu.py)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pso_update(PsoDev d, PsoConst c, int i0, int i1)
{
    const int p = blockIdx.y;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    extern __shared__ double lds[];
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = i0 + blockIdx.x * (blockDim.x >> 4) + r, ld = c.ld, n = c.n;
    const bool live = i < i1;
    double *row = lds + r * ld;
    const size_t base = ((size_t) p * c.np + (live ? i : i0)) * ld;
    const double *gb = d.xbest + (size_t) p * ld;
    const double w = sc->w, c1 = sc->c1, c2 = sc->c2;
    const int it = sc->it;
    double ssq = 0.;
    if (live) {
        for (int pj = g; pj < ld / 2; pj += 16) {
            const int j = 2 * pj;
            double2 x = *reinterpret_cast<const double2*>(&d.X[base + j]);
            double2 v = *reinterpret_cast<const double2*>(&d.V[base + j]);
            const double2 xb = *reinterpret_cast<const double2*>(&d.XB[base + j]);
            const double2 gg = *reinterpret_cast<const double2*>(&gb[j]);
            if (j < n) {
                const u32x4 q = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) j, (uint32_t) it,
                        stream_word(STREAM_PSO_R, (uint32_t) p));
                const double r1 = u01(q.x, q.y), r2 = u01(q.z, q.w);
                v.x = v.x * w + c1 * r1 * (xb.x - x.x) + c2 * r2 * (gg.x - x.x);
                const double vmax = 0.2 * (d.upper[j] - d.lower[j]);
                v.x = fmax(-vmax, fmin(v.x, vmax));
                x.x += v.x;
                if (c.correct) x.x = fmax(d.lower[j], fmin(x.x, d.upper[j]));
            }
            if (j + 1 < n) {
                const u32x4 q = philox4x32_10(c.seed, (uint32_t) i, (uint32_t) (j + 1),
                        (uint32_t) it, stream_word(STREAM_PSO_R, (uint32_t) p));
                const double r1 = u01(q.x, q.y), r2 = u01(q.z, q.w);
                v.y = v.y * w + c1 * r1 * (xb.y - x.y) + c2 * r2 * (gg.y - x.y);
                const double vmax = 0.2 * (d.upper[j + 1] - d.lower[j + 1]);
                v.y = fmax(-vmax, fmin(v.y, vmax));
                x.y += v.y;
                if (c.correct) x.y = fmax(d.lower[j + 1], fmin(x.y, d.upper[j + 1]));
            }
            *reinterpret_cast<double2*>(&d.X[base + j]) = x;
            *reinterpret_cast<double2*>(&d.V[base + j]) = v;
            *reinterpret_cast<double2*>(&row[j]) = x;
            ssq += x.x * x.x + x.y * x.y;
        }
    }
    __syncthreads();
    ssq = pso_group_sum<16>(ssq);
    if (live && g == 0) d.radius[(size_t) p * c.np + i] = sqrt(ssq);
    if (c.obj < 0) return;   // host objective: pso_pbest runs after the host evaluation
    double f = eval_row_group<16>(c.obj, n, row, d.aux, g);
    if (f != f) f = PSO_INF;
    if (live) {
        const double fb = d.fb[(size_t) p * c.np + i];
        if (f < fb)
            for (int pj = g; pj < ld / 2; pj += 16)
                *reinterpret_cast<double2*>(&d.XB[base + 2 * pj]) =
                        *reinterpret_cast<const double2*>(&row[2 * pj]);
        if (g == 0) {
            d.f[(size_t) p * c.np + i] = f;
            if (f < fb) d.fb[(size_t) p * c.np + i] = f;
        }
    }
}

// host-objective path: pbest update of the particles [i0, i1) from the uploaded fitness
__global__ __launch_bounds__(256) void pso_pbest(PsoDev d, PsoConst c, int i0, int i1)
{
    const int p = blockIdx.y;
    const PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    const int tid = threadIdx.x, r = tid >> 4, g = tid & 15;
    const int i = i0 + blockIdx.x * 16 + r;
    if (i >= i1) return;
    const size_t base = ((size_t) p * c.np + i) * c.ld;
    const double f = d.f[(size_t) p * c.np + i], fb = d.fb[(size_t) p * c.np + i];
    if (f < fb) {
        for (int pj = g; pj < c.ld / 2; pj += 16)
            *reinterpret_cast<double2*>(&d.XB[base + 2 * pj]) =
                    *reinterpret_cast<const double2*>(&d.X[base + 2 * pj]);
        if (g == 0) d.fb[(size_t) p * c.np + i] = f;
    }
}

// the swarm's best after the particles [i0, i1) have moved: arg-min of their new fitness (lowest
// index on ties, as the reference's loop meets them), taken over if it beats the best so far.
// One workgroup of 256 per population.  (pso_finish repeats the scan over the whole swarm at the
// end of the generation; after these refreshes it finds nothing new.)
__global__ __launch_bounds__(256) void pso_gbest(PsoDev d, PsoConst c, int i0, int i1)
{
    const int p = blockIdx.x;
    PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    __shared__ double sval[4];
    __shared__ int sidx[4];
    const int tid = threadIdx.x, np = c.np, ld = c.ld;
    const double *f = d.f + (size_t) p * np;
    double fv = PSO_INF;
    int ib = 0x7fffffff;
    for (int i = i0 + tid; i < i1; i += 256)
        if (f[i] < fv) {
            fv = f[i];
            ib = i;
        }
    pso_block_arg<1>(fv, ib, sval, sidx);
    const bool improved = ib != 0x7fffffff && fv < sc->fbest;
    if (improved)
        for (int j = tid; j < ld; j += 256)
            d.xbest[(size_t) p * ld + j] = d.X[((size_t) p * np + ib) * ld + j];
    if (improved && tid == 0) sc->fbest = fv;
}

// gbest arg-min over the new fitness, stop test, counters.  One workgroup of 256 per population
__global__ __launch_bounds__(256) void pso_finish(PsoDev d, PsoConst c)
{
    const int p = blockIdx.x;
    PsoScal *sc = d.scal + p;
    if (pso_frozen(c, sc)) return;
    __shared__ double sval[4];
    __shared__ int sidx[4];
    const int tid = threadIdx.x, np = c.np, ld = c.ld;
    const double *f = d.f + (size_t) p * np, *rad = d.radius + (size_t) p * np;
    double fv = PSO_INF;
    int ib = 0x7fffffff;
    for (int i = tid; i < np; i += 256)
        if (f[i] < fv) {
            fv = f[i];
            ib = i;
        }
    pso_block_arg<1>(fv, ib, sval, sidx);
    const bool improved = ib != 0x7fffffff && fv < sc->fbest;
    if (improved)
        for (int j = tid; j < ld; j += 256)
            d.xbest[(size_t) p * ld + j] = d.X[((size_t) p * np + ib) * ld + j];
    double s = 0.;
    for (int i = tid; i < np; i += 256) s += rad[i];
    const double mean = pso_block_sum(s, sval) / np;
    double m2 = 0.;
    for (int i = tid; i < np; i += 256) {
        const double dd = rad[i] - mean;
        m2 += dd * dd;
    }
    m2 = pso_block_sum(m2, sval);
    if (tid == 0) {
        if (improved) sc->fbest = fv;
        sc->it += 1;
        sc->fev += np;
        sc->m2 = m2;
        const int conv = m2 <= (np - 1) * c.tol * c.tol ? 1 : 0;
        sc->conv = conv;
        if (conv) sc->stop = 1;
        else if (sc->it >= sc->maxit || sc->fev >= c.mfev) sc->stop = 2;
    }
}

} // namespace bbo
