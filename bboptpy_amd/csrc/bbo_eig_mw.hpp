// bbo_eig_mw.hpp -- the Householder reduction of ONE matrix spread over several compute units
// (round 4; 128 < n <= 256 with few matrices in flight: C5's BIPOP, one population per GPU).
//
// The single-workgroup reduction of bbo_eig.hpp is a chain of n - 1 dependent steps on one CU
// (5.4 us per step with the whole active matrix on chip at n = 256: 0.8 ms of a 1.2 ms
// decomposition) while 255 CUs idle.  Here MW_G workgroups hold the matrix by rows, CYCLICALLY
// (row r lives in workgroup r mod MW_G, so the shrinking active block stays spread evenly), full
// rows in registers: a thread owns 32 entries of one row, eight lanes share a row.  Because the
// rows are whole, p = A u needs no transposed product and no cross-workgroup reduction: every
// workgroup forms the entries of p for ITS rows.  What a step has to exchange:
//   (1) the pieces of p (all-gather: every workgroup needs all of p for the scalar u^T p and w),
//   (2) the next pivot row.
// (2) is folded into (1): the owner of row i - 1 publishes that row AS IT STANDS together with
// its piece of p, and every wavefront of every workgroup forms the updated row itself once it has
// w (row' = row - u_{i-1} w - w_{i-1} u: three vector operations on four entries per lane).  ONE
// exchange per step: every wavefront stores its piece with relaxed agent-scope atomics and waits for
// the stores; the last one of a workgroup to do so raises the workgroup's flag (a monotone epoch:
// launch * 1024 + step; an LDS arrival count, no barrier); every wavefront polls the MW_G flags and
// then loads the vectors -- no fence, no cache invalidate.  Measured (scripts/hiptests/
// allgather.hip, pingpong.hip): 1.05 us per such exchange with the workgroups on ONE XCD, 1.7 us
// on eight; a release / acquire pair costs 2.2 us one way.  All O(n) vector work (norm, root, u,
// u^T p, w) is done redundantly by every wavefront on a 4-entries-per-lane copy, as in the
// one-workgroup code; LDS only turns that layout into the row / column layout of the products (a
// wavefront's own copy).  There is NO workgroup barrier in a step: wavefronts never wait for each
// other inside a workgroup, so one that gives up (below) cannot leave the others at a barrier.
//
// What it buys (round 4, one matrix).  A spread step takes ~2.8 us at n <= 256 -- ~1.3 us of it the
// wavefront's own chain (two wavefront reductions, root and reciprocal, two LDS hand-overs, 96
// FMAs), the rest the exchange -- whatever the size of the active block; the one-workgroup steps
// cost 5.4 us with the whole active matrix on chip (n = 256) and 1.15 us once it is 128 x 128 and
// lives in registers.  With ALL steps spread the n = 256 decomposition took 1.10 against 1.18 ms
// (and lost below n = 240).  Hence istop: the spread steps stop at the leading 128 x 128 block,
// which goes to ONE workgroup (cma_tred_tail, the register-resident reduction): 0.83 against
// 1.04 ms at n = 256, ahead for every n > 128.  For 256 < n <= 512 (16 workgroups, rows of 512
// entries, 5.5 us per step) it replaces a one-workgroup reduction that streams the matrix from L2
// and accumulates Q_house: 41 -> 5 ms at n = 512.  The floor of the design is the exchange,
// (n - 128) x 1.05 us.  Tried on top and slower, each measured at n = 256 with all steps spread:
// the flags side by side in one 256-byte line (1.59 ms: 32 pollers and 8 writers on one channel);
// eight unconditional loads per lane instead of only the active block's (1.17); one wavefront
// per workgroup polling and loading for all four through LDS and a barrier (1.19); every double
// as two self-validating {half, tag} words, no flags and no wait for the stores (1.17: twice
// the requests).  Agent-scope accesses are served on the memory side, and what decides is how
// few of them a step makes.
//
// Same-XCD placement: workgroups are dealt to the 8 XCDs round-robin by their linear index; the
// grid is (8 MW_G, P) and only the blocks with blockIdx.x % 8 == (p + xcd0) % 8 work, the others
// return (xcd0: per engine, so that engines sharing a GPU do not all sit on XCD 0).
//
// Every wait is bounded by the constant-rate wall clock (MW_TIMEOUT_TICKS of wall_clock64(), 100 MHz
// on gfx950: 50 ms whatever the shader clock does; the clock is only read every 1024th poll, so a
// step that gets its pieces pays nothing): on a time-out the wavefront raises the sticky
// CmaScal::eig_mw_fail and the engine's pinned host flag, skips the hand-over (eig_stage stays 0:
// this generation keeps its basis, like a generation the lazy schedule skips), and every later
// launch of these kernels returns at entry (nobody spins twice).  The host reads the flag after
// every synchronisation (iterate / phase / run's poll), stops using this path and runs the
// one-workgroup decomposition for the generation that lost its own.  The host only launches it when
// all workgroups of all engines of this process on the device fit the chip at once
// (MwBudget, bbo_cma.hip).
//
// Ordering.  A piece is published as: data stores (relaxed, agent scope) -> s_waitcnt 0 -> LDS
// arrival count -> flag store; it is consumed as: flag loads until all are there -> data loads.
// The hardware keeps that order for one wavefront (the stores are acknowledged before the count
// goes up; a load issued after the poll's result is used cannot be served before it), what could
// move them is the compiler: __builtin_amdgcn_s_waitcnt carries no memory semantics.  So the two
// places are pinned with __atomic_signal_fence(SEQ_CST) -- a compiler-only fence, no cache
// operation, no instruction -- and tests/test_isa_mw.py disassembles the shipped library and checks
// the order of the instructions themselves.
//
// Output (what cma_eigen_g1 leaves): eig_work[3] = [d | e (shifted down) | h], the reflectors
// V (row i = u_i, zero from column i on) dense n x n in eig_work[1], CmaScal::eig_stage = 1.
// Conventions of eig_tred_accum_reg128: unscaled reflectors, H(i) = I - u_i u_i^T / h_i.
#pragma once

#include "bbo_eig.hpp"

namespace bbo {

constexpr int MW_G = 8;                 // workgroups per matrix of n <= 256 (NMAX / 32 in general: 16 to n = 512)
constexpr int MW_T = 256;               // threads per workgroup: 32 rows x 8 lanes
constexpr long long MW_TIMEOUT_TICKS = 5000000;     // of wall_clock64() (100 MHz): 50 ms
constexpr int MW_WAVES = MW_T / 64;
// (variant builds for A/B timing: scripts/build_variant.sh)
#ifdef MW_NO_FENCE
#define MW_CFENCE() do { } while (0)
#else
#define MW_CFENCE() __atomic_signal_fence(__ATOMIC_SEQ_CST)
#endif
#ifndef MW_POLL_SLEEP
#define MW_POLL_SLEEP 1
#endif
constexpr int MW_FLAG_STRIDE = 16;      // 64-bit words between two flags (128 bytes: side by side, the polls of 32
                                        // wavefronts and the flag stores queue up on one channel -- measured, +40 %)
// per population: flags | e[2][NMAX] | row[2][NMAX]
constexpr int mw_buf_doubles(int nmax) { return (nmax / 32) * MW_FLAG_STRIDE + 4 * nmax; }
constexpr int MW_BUF_DOUBLES = mw_buf_doubles(256);

__device__ inline unsigned long long mw_load(const void *p)
{
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void mw_store(void *p, unsigned long long v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED,
            __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double mw_load_d(const double *p) { return __longlong_as_double((long long) mw_load(p)); }
__device__ inline void mw_store_d(double *p, double v) { mw_store(p, (unsigned long long) __double_as_longlong(v)); }

// entry `idx` (wavefront-uniform) of a vector held four entries per lane (entry lane + 64 v)
template<int NV>
__device__ inline double mw_entry(const double (&x)[NV], int idx)
{
    const int l = idx & 63, v = idx >> 6;
    double e = eig_readlane(x[0], l);
#pragma unroll
    for (int u = 1; u < NV; u++) {
        const double eu = eig_readlane(x[u], l);
        e = v == u ? eu : e;
    }
    return e;
}

// sum_v x_v y_v over a lane's entries, four at a time in the association ((0 + 1) + (2 + 3))
template<int NV>
__device__ inline double mw_dot(const double (&x)[NV], const double (&y)[NV])
{
    double s = (x[0] * y[0] + x[1] * y[1]) + (x[2] * y[2] + x[3] * y[3]);
#pragma unroll
    for (int u = 4; u < NV; u += 4)
        s += (x[u] * y[u] + x[u + 1] * y[u + 1]) + (x[u + 2] * y[u + 2] + x[u + 3] * y[u + 3]);
    return s;
}

// sum over the 8 lanes of a row (two quads of one half of a DPP row): quad butterfly, then the
// mirror image inside the half row brings the other quad's sum
__device__ inline double mw_row8_sum(double v)
{
    v = eig_quad_sum(v);
    v += eig_dpp<0x141>(v);      // row_half_mirror
    return v;
}

// grid (8 * MW_G, P), MW_T threads; mwbuf: MW_BUF_DOUBLES per population, zero at allocation;
// launch: a counter the host increments per launch (the flags are never reset)
// istop: the last pivot row this kernel takes (1: the whole reduction; 128: the leading 128 x 128
// block, updated, goes to eig_work[0] (row stride 128) and cma_tred_tail finishes it on one
// workgroup, whose steps cost 1.15 us where the steps here cost 2.8)
// xcd0: the engine's offset into the XCDs (engines that share a GPU start at different ones).
// (277 registers per lane: ONE workgroup per CU.  Capped at 256 for two -- __launch_bounds__(MW_T,
// 2), 3 spilled -- the decomposition took 867 instead of 835 us, and with u's column copy read
// again from LDS instead of held, 888: the speed of one matrix was kept.  A launch therefore needs
// a free CU per workgroup; engines that share a GPU start on different XCDs, and whoever does not
// get its partners in time falls back, see above.)
// chained (nact > 0): the SECOND spread kernel of a 256 < n <= 512 reduction -- cma_tred_mw512 has
// taken the steps down to pivot row `nact` (= 256: its steps cost 5.5 us, 64 wavefronts exchanging
// rows of 512), left the leading nact x nact block in eig_work[0] (row stride nact) and the value 1
// in the hand-over flag; this kernel (rows of 256, 2.9 us per step) takes that block down to istop
// and leaves 2 there.  The flag is only read at the start (workgroups start at different times:
// nobody may reset it under a late one's feet).
template<int NMAX>
__device__ __forceinline__ void tred_mw_body(const CmaDev &d, const CmaConst &c, int force, double *mwbuf,
        unsigned long long launch, int istop, int xcd0, int nact = 0)
{
    const bool chained = nact > 0;
    constexpr int G = NMAX / 32, NT = NMAX / 16, NV = NMAX / 64;
    const int p = blockIdx.y;
    if ((int) (blockIdx.x & 7) != ((p + xcd0) & 7)) return;
    const int g = blockIdx.x >> 3;
    CmaScal *sc = d.scal + p;
    if (c.honor_stop && sc->stop != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // a wavefront of an earlier launch gave up (the host has not looked yet): nobody waits again, and
    // the kernels behind this one see a generation without a decomposition
    const bool gave_up = sc->eig_mw_fail != 0;
    if (chained) {
        if (gave_up || d.eig_work[(size_t) (4 * p + 3) * eig_slab(c.ld) + 4 * c.n + 1] != 1.) return;
    } else
    // cmaes.cpp:233: skip until enough evaluations have passed
    if (gave_up || (!force && !((double) (sc->fev - sc->eigenlastev) > c.eigenfreq))) {
        if (g == 0 && tid == 0) {
            sc->eigen_done = 0;
            sc->eig_stage = 0;
            // (cma_tred_tail must not take what an earlier generation, or the products that have
            // used this slab since, left in the hand-over flag)
            d.eig_work[(size_t) (4 * p + 3) * eig_slab(c.ld) + 4 * c.n + 1] = 0.;
        }
        return;
    }
    // (fault injection, BBO_MW_FAULT_STEP in the environment of the process that creates the engine
    // -- not reachable through bbo_set: workgroup 3 walks away, at entry (0) or after taking part in
    // k steps (k > 0) -- what the others do about a partner that stops publishing is tested, not
    // assumed: tests/test_cma_gpu.py)
#ifdef MW_NO_FAULT
    const int fault_step = -1;
#else
    const int fault_step = g == 3 ? d.mw_fault : -1;
#endif
    if (fault_step == 0) return;
    __shared__ __attribute__((aligned(16))) double ubuf[4][NMAX];
    __shared__ __attribute__((aligned(16))) double wbuf[4][NMAX];
    __shared__ unsigned arrived;          // wavefronts that have published, over all steps so far
    if (tid == 0) arrived = 0u;
    __syncthreads();
    const int n = c.n;
    // (the matrix this kernel starts from: the covariance, or the block the kernel before it left)
    const int na = chained ? nact : n, ld = chained ? nact : c.ld;
    const double *C = chained ? d.eig_work + (size_t) (4 * p) * eig_slab(c.ld) : d.C + (size_t) p * c.ld * c.ld;
    double *tri = d.eig_work + (size_t) (4 * p + 3) * eig_slab(c.ld);
    double *Vout = d.eig_work + (size_t) (4 * p + 1) * eig_slab(c.ld);
    double *mb = mwbuf + (size_t) p * mw_buf_doubles(NMAX);
    unsigned long long *flags = reinterpret_cast<unsigned long long*>(mb);
    double *ebuf = mb + G * MW_FLAG_STRIDE;             // [2][NMAX], entry of row r at (r % G) * 32 + r / G
    double *rbuf = ebuf + 2 * NMAX;                     // [2][NMAX]

    // this thread's row and columns
    const int q = 8 * wave + (lane >> 3), s = lane & 7;
    const int r = G * q + g;
    double2 a2[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int col = 16 * t + 2 * s;
        a2[t].x = (r < na && col < na) ? C[(size_t) r * ld + col] : 0.;
        a2[t].y = (r < na && col + 1 < na) ? C[(size_t) r * ld + col + 1] : 0.;
    }
    // the pivot row of the first step, four entries per lane (zero from the pivot column on)
    double av[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
        const int idx = lane + 64 * v;
        av[v] = idx < na - 1 ? C[(size_t) (na - 1) * ld + idx] : 0.;
    }
    const bool recorder = g == 0 && wave == 0;
    if (recorder && lane == 0 && !chained) {
        tri[n - 1] = C[(size_t) (n - 1) * ld + n - 1];
        tri[n + n - 1] = 0.;           // (the sub-diagonal is handed over shifted down by one)
        tri[2 * n] = 0.;
        tri[4 * n] = 0.;               // (T factors: not built yet, cma_eig_halves' third workgroup)
        tri[4 * n + 1] = 0.;           // (this kernel's part of the reduction: not done yet)
    }
    if (recorder && !chained)
        for (int idx = lane; idx < n; idx += 64) Vout[idx] = 0.;      // row 0: no reflector
    double *ub = ubuf[wave], *wb = wbuf[wave];
    bool failed = false;
#ifdef BBO_MW_CLOCKS
    long long ck[6] = { 0, 0, 0, 0, 0, 0 }, ct = clock64();
#define MW_CK(k) do { const long long t_ = clock64(); ck[k] += t_ - ct; ct = t_; } while (0)
#else
#define MW_CK(k) do { } while (0)
#endif

    // the reflector of step i from its pivot row, by every wavefront: h, 1 / h, the corner value
    // f, the root gg, u in the four-entries-per-lane layout.  Formed at the END of the step before
    // (software-pipelined): its chain of dependent operations -- a wavefront reduction, a root, a
    // reciprocal -- then shares the instruction stream with that step's rank-2 update, 64 independent
    // FMAs, instead of standing alone in front of the product.
    double f = 0., gg = 0., h = 0., rh = 0., uvv[NV];
    bool none = true;
    auto reflector = [&](int i, const double (&a)[NV], double &f_, double &gg_, double &h_, double &rh_,
            bool &none_, double (&u)[NV]) {
        const double h0 = eig_wave_sum_bf(mw_dot<NV>(a, a));
        f_ = mw_entry(a, i - 1);
        none_ = h0 == 0.;
        gg_ = 0.;
        if (!none_) {
            double y = __builtin_amdgcn_rsq(h0);
            const double err = fma(-h0 * y, y, 1.);
            y = fma(y * err, fma(err, 0.375, 0.5), y);
            gg_ = h0 * y;
            gg_ = fma(fma(-gg_, gg_, h0), 0.5 * y, gg_);
            if (f_ > 0) gg_ = -gg_;
        }
        h_ = none_ ? 0. : h0 - f_ * gg_;
        rh_ = none_ ? 0. : dc_rcp(h_);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const int idx = lane + 64 * v;
            u[v] = (idx < i && !none_) ? (idx == i - 1 ? f_ - gg_ : a[v]) : 0.;
        }
    };
    reflector(na - 1, av, f, gg, h, rh, none, uvv);
#pragma unroll
    for (int v = 0; v < NV; v++) ub[lane + 64 * v] = uvv[v];
    dc_wave_sync();

    for (int i = na - 1; i >= istop && !failed; i--) {
        const unsigned long long epoch = launch * 1024ull + (unsigned long long) (na - i);
        const int par = i & 1;
        if (fault_step > 0 && na - 1 - i == fault_step) return;
        MW_CK(0);
        // ---- p = A u for this thread's row; its piece of e = p / h goes out ----------------------
        double2 uc[NT];
        double acc0 = 0., acc1 = 0.;
#pragma unroll
        for (int t = 0; t < NT; t++) uc[t] = *reinterpret_cast<const double2*>(&ub[16 * t + 2 * s]);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            acc0 = __builtin_fma(a2[t].x, uc[t].x, acc0);
            acc1 = __builtin_fma(a2[t].y, uc[t].y, acc1);
        }
        const double pr = mw_row8_sum(acc0 + acc1);
        // (a wavefront's eight entries lie side by side: one 64-byte piece per wavefront)
        if (s == 0 && r < n) mw_store_d(ebuf + NMAX * par + 32 * g + q, r < i ? pr * rh : 0.);
        // the row that becomes the next pivot, as it stands (its owner: row i - 1) -- turned into
        // the four-entries-per-lane layout through this wavefront's w buffer (free here), so that
        // it leaves as four 512-byte stores, not 64 scattered ones
        if ((i - 1) % G == g && ((i - 1) / G) >> 3 == wave) {
            if (r == i - 1) {
#pragma unroll
                for (int t = 0; t < NT; t++) *reinterpret_cast<double2*>(&wb[16 * t + 2 * s]) = a2[t];
            }
            dc_wave_sync();
#pragma unroll
            for (int v = 0; v < NV; v++) mw_store_d(rbuf + NMAX * par + lane + 64 * v, wb[lane + 64 * v]);
        }
        MW_CFENCE();
        __builtin_amdgcn_s_waitcnt(0);          // this wavefront's stores are out (and its loads of
                                                // the step before: the buffers alternate)
        MW_CFENCE();
        // the LAST wavefront of this workgroup to get here raises the workgroup's flag (an LDS
        // count, no barrier: nobody waits inside the workgroup)
        if (lane == 0) {
            const unsigned before = atomicAdd(&arrived, 1u);
            if (before + 1u == (unsigned) (MW_WAVES * (na - i))) mw_store(flags + MW_FLAG_STRIDE * g, epoch);
        }
        MW_CK(1);
        // ---- every wavefront waits for all pieces ------------------------------------------------
        {
            int spins = 0;
            long long t_wait = 0;
            while (true) {
                const unsigned long long fl = lane < G ? mw_load(flags + MW_FLAG_STRIDE * lane) : epoch;
                if (__ballot(fl < epoch) == 0ull) break;
#ifdef MW_OLD_SPIN
                if (++spins >= (1 << 21)) {
                    failed = true;
                    break;
                }
                if (false) {
#else
                if ((++spins & 1023) == 0) {
#endif
                    const long long now = (long long) wall_clock64();
                    if (t_wait == 0) t_wait = now;
                    else if (now - t_wait > MW_TIMEOUT_TICKS) {
                        failed = true;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(MW_POLL_SLEEP);
            }
        }
        if (failed) break;
        MW_CFENCE();                                  // (the data loads stay behind the poll)
        MW_CK(2);
        // (only the lanes inside the active block ask: measured against eight unconditional loads
        // per lane, 1071 against 1170 us per decomposition -- the requests are served on the memory
        // side, and fewer of them come back sooner)
        double ev_[NV], ro[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const int idx = lane + 64 * v;
            ev_[v] = idx < i ? mw_load_d(ebuf + NMAX * par + 32 * (idx % G) + idx / G) : 0.;
            ro[v] = idx < i ? mw_load_d(rbuf + NMAX * par + idx) : 0.;
        }
        MW_CK(3);
        // ---- w = e - (u^T e / 2h) u ----------------------------------------------------------------
        const double hh = eig_wave_sum_bf(mw_dot<NV>(ev_, uvv)) * (0.5 * rh);
        double wvv[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const int idx = lane + 64 * v;
            wvv[v] = idx < i ? ev_[v] - hh * uvv[v] : 0.;
            wb[idx] = wvv[v];
        }
        dc_wave_sync();
        // ---- the next pivot row, by every wavefront: row' = row - u_{i-1} w - w_{i-1} u ------------
        const double um = none ? 0. : f - gg, wm = mw_entry(wvv, i - 1);
        double an[NV];
#pragma unroll
        for (int v = 0; v < NV; v++) an[v] = ro[v] - (um * wvv[v] + wm * uvv[v]);
        // ---- what this step leaves behind ------------------------------------------------------------
        if (recorder) {
            if (lane == 0) {
                tri[n + i - 1] = none ? f : gg;      // e[i], shifted down by one
                tri[2 * n + i] = h;
            }
#pragma unroll
            for (int v = 0; v < NV; v++) {
                const int idx = lane + 64 * v;
                if (idx < n) Vout[(size_t) i * n + idx] = uvv[v];
                if (idx == i - 1) tri[i - 1] = an[v];           // diagonal entry of row i - 1: final
            }
            // (a row of V is n entries: beyond this kernel's NMAX they are zero)
            for (int idx = NMAX + lane; idx < n; idx += 64) Vout[(size_t) i * n + idx] = 0.;
        }
#pragma unroll
        for (int v = 0; v < NV; v++) av[v] = lane + 64 * v < i - 1 ? an[v] : 0.;
        // ---- the reflector of the NEXT step (its dependent chain runs under the update below) ----------
        // (unconditionally -- behind a test it would be a basic block of its own and the scheduler
        // could not mix it with the update; after the last step it is formed once for nothing)
        double f2, gg2, h2, rh2, u2[NV];
        bool none2;
        reflector(max(i - 1, 1), av, f2, gg2, h2, rh2, none2, u2);
        // ---- A -= u w^T + w u^T on this thread's row (rows and columns >= i see zeros) -------------
        {
            const int rc = r < NMAX ? r : 0;
            const double ur = r < n ? ub[rc] : 0., wr = r < n ? wb[rc] : 0.;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const double2 wc = *reinterpret_cast<const double2*>(&wb[16 * t + 2 * s]);
                a2[t].x -= ur * wc.x + wr * uc[t].x;
                a2[t].y -= ur * wc.y + wr * uc[t].y;
            }
        }
        MW_CK(4);
        // (ub / wb are this wavefront's own: u of the next step replaces this step's, whose reads --
        // u_r above, the columns in registers since the product -- precede the write in program order)
        f = f2; gg = gg2; h = h2; rh = rh2; none = none2;
#pragma unroll
        for (int v = 0; v < NV; v++) {
            uvv[v] = u2[v];
            ub[lane + 64 * v] = u2[v];
        }
        dc_wave_sync();
        MW_CK(5);
    }
#ifdef BBO_MW_CLOCKS
    if (d.stamps && recorder && lane == 0 && p == 0)
        for (int k = 0; k < 6; k++) d.stamps[40 + k] = ck[k] / (n - 1);
#endif
#undef MW_CK
    if (failed) {
        if (lane == 0) {
            sc->eig_mw_fail = 1;
            sc->eigen_done = 0;       // (the products that follow must not run on what is half there)
            if (d.mw_fail_host) __hip_atomic_store(d.mw_fail_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (istop > 1) {
        // the rest is one workgroup's: this thread's part of the leading block, as it stands
        double *L11 = d.eig_work + (size_t) (4 * p) * eig_slab(c.ld);
        if (r < istop) {
#pragma unroll
            for (int t = 0; t < NT; t++)
                if (16 * t + 2 * s < istop)
                    *reinterpret_cast<double2*>(&L11[(size_t) r * istop + 16 * t + 2 * s]) = a2[t];
        }
        if (recorder && lane == 0) tri[4 * n + 1] = chained ? 2. : 1.;
        return;
    }
    if (g == 0 && tid == 0) sc->eig_stage = 1;
}

__global__ __launch_bounds__(MW_T) void cma_tred_mw(CmaDev d, CmaConst c, int force, double *mwbuf,
        unsigned long long launch, int istop, int xcd0)
{
    tred_mw_body<256>(d, c, force, mwbuf, launch, istop, xcd0);
}
// the second spread kernel of a 256 < n <= 512 reduction (see `chained` above)
__global__ __launch_bounds__(MW_T) void cma_tred_mw_chain(CmaDev d, CmaConst c, double *mwbuf,
        unsigned long long launch, int istop, int xcd0)
{
    tred_mw_body<256>(d, c, 0, mwbuf, launch, istop, xcd0, 256);
}
// 256 < n <= 512: 16 workgroups, 64 entries of a row per thread (grid (8 * 16, P))
__global__ __launch_bounds__(MW_T) void cma_tred_mw512(CmaDev d, CmaConst c, int force, double *mwbuf,
        unsigned long long launch, int istop, int xcd0)
{
    tred_mw_body<512>(d, c, force, mwbuf, launch, istop, xcd0);
}

} // namespace bbo
