// bbo_cma.hip -- host side of the CMA-ES / active CMA-ES engine: strategy constants,
// HBM state, and the kernel sequence of one generation.
//
// Reference behaviour restated on the host: BaseCmaes::init (base_cmaes.cpp:54-134),
// Cmaes::init (cmaes.cpp:44-63), ActiveCmaes::init (active_cmaes.cpp:42-69),
// BaseCmaes::setParams (:136-148), optimize (:162-174), solution (:158-160).
#include "bbo_cma_kernels.hpp"
#include "bbo_sep_kernels.hpp"
#include "bbo_eig_mw.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <limits>

namespace bbo {

namespace {

// kernel slots of the profile report (bbo_get "profile"), in launch order
enum { K_SAMPLE = 0, K_RANK, K_WHITEN, K_GRAM, K_PATHS, K_COV, K_EIGEN, K_POST, K_STOP, K_COUNT };
static const char *const K_NAMES[K_COUNT] = { "bbo:cma_sample_eval", "bbo:cma_rank", "bbo:cma_whiten", "bbo:cma_gram", "bbo:cma_paths", "bbo:cma_cov", "bbo:cma_eigen", "bbo:cma_post", "bbo:cma_history_stop" };   // roctx ranges, bench.py's slot names

int pick_maxt(int ld)
{
    const int per_wave = ((ld >> 4) + 3) / 4;
    if (per_wave <= 1) return 1;
    if (per_wave <= 2) return 2;
    if (per_wave <= 4) return 4;
    return 8;
}

int gram_ldy(int ld)
{
    // rows k, k+1 of the slab must land 32 banks apart for the k-major fragment reads
    return (ld % 32 == 0) ? ld + 16 : ld;
}

// dynamic LDS of cma_gram: the slab, its two coefficient columns, the mean's partial sums
constexpr size_t GRAM_LDS_MAX = 160 * 1024 - 64;
size_t gram_lds_bytes(int ld, int rps)
{
    const int rpp = 256 / (ld / 4) > 0 ? 256 / (ld / 4) : 1;
    return ((size_t) rps * gram_ldy(ld) + 2 * (size_t) rps + (size_t) rpp * ld) * sizeof(double);
}

} // namespace

CmaEngine::CmaEngine(const bbo_params &p) :
        params_(p)
{
    BBO_REQUIRE(p.algo == BBO_ALGO_CMAES || p.algo == BBO_ALGO_ACTIVE_CMAES
            || p.algo == BBO_ALGO_SEP_CMAES, "CmaEngine: algo must be CMAES, ACTIVE_CMAES or SEP_CMAES");
    BBO_REQUIRE(p.np >= 4, "CMA-ES needs np >= 4 (mu >= 2, best/worst pairs)");
    BBO_REQUIRE(p.populations >= 1, "populations must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(BBO_ERR_NO_DEVICE, "no HIP device visible: libbbopt_hip has no CPU path");
    BBO_REQUIRE(p.device >= 0 && p.device < ndev, "device ordinal out of range");
    BBO_HIP(hipSetDevice(p.device));
    BBO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    BBO_HIP(hipHostMalloc((void**) &mw_fail_host_, sizeof(int)));
    *mw_fail_host_ = 0;
}

CmaEngine::~CmaEngine()
{
    mw_release();
    if (stream_) (void) hipStreamDestroy(stream_);
    if (mw_fail_host_) (void) hipHostFree(mw_fail_host_);
}

// ---- the spread reduction's share of the device (bbo_eig_mw.hpp) -----------------------------------
// Its workgroups wait for each other, so all of them -- of every engine of this process that may
// have such a kernel in flight on the device -- must be resident at once.  Capacity: compute units
// (hipDeviceProp) x workgroups of cma_tred_mw512 a compute unit holds (the occupancy API; 1 on
// gfx950: 277 / 512 registers per lane), minus a margin of a sixteenth for whatever else runs.  An
// engine reserves its count before its first spread launch and gives it back when it dies, changes
// shape or falls back; who gets no reservation uses the one-workgroup reduction.  Other PROCESSES
// on the device are not seen: against them stands the bounded wait and the fallback.
namespace {
struct MwBudget {
    static constexpr int MAXDEV = 64;
    std::mutex m;
    long cap[MAXDEV] = {};
    long used[MAXDEV] = {};
    bool known[MAXDEV] = {};
    static MwBudget &get()
    {
        // (never destroyed: an engine that outlives the library's static objects -- a global in the
        // caller's program -- still gives its share back through a live object)
        static MwBudget *b = new MwBudget;
        return *b;
    }
    long capacity(int dev)
    {
        if (!known[dev]) {
            hipDeviceProp_t prop;
            int per_cu = 0;
            long cus = 0;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*) cma_tred_mw512, MW_T, 0)
                    != hipSuccess || per_cu < 1)
                per_cu = 1;
            cap[dev] = cus * per_cu - std::max(1L, cus / 16);
            known[dev] = true;
        }
        return cap[dev];
    }
    bool reserve(int dev, long wgs)
    {
        if (dev < 0 || dev >= MAXDEV) return false;
        std::lock_guard<std::mutex> lock(m);
        if (used[dev] + wgs > capacity(dev)) return false;
        used[dev] += wgs;
        return true;
    }
    void release(int dev, long wgs)
    {
        if (dev < 0 || dev >= MAXDEV) return;
        std::lock_guard<std::mutex> lock(m);
        used[dev] -= wgs;
    }
};
} // namespace

// fault injection for the tests of the spread reduction: the ENVIRONMENT of this process only
// (read when an engine is initialised and on the phase-by-phase path the tests drive; bbo_set
// cannot reach it)
static int mw_fault_from_env()
{
    const char *e = std::getenv("BBO_MW_FAULT_STEP");
    return e && *e ? std::atoi(e) : -1;
}

bool CmaEngine::mw_reserve(long workgroups)
{
    if (mw_reserved_ == workgroups) return true;
    mw_release();
    if (!MwBudget::get().reserve(params_.device, workgroups)) return false;
    mw_reserved_ = workgroups;
    return true;
}

void CmaEngine::mw_release()
{
    if (mw_reserved_ > 0) MwBudget::get().release(params_.device, mw_reserved_);
    mw_reserved_ = 0;
}

// after a synchronisation of the stream: did a spread reduction launched since the last look give up?
bool CmaEngine::mw_check_failed()
{
    if (!mw_launched_) return false;
    mw_launched_ = false;
    if (!*mw_fail_host_) return false;
    *mw_fail_host_ = 0;
    mw_disabled_ = true;
    mw_release();
    return true;
}

void CmaEngine::set_params(int np, double sigma, int mfev)
{
    params_.np = np;
    params_.sigma0 = sigma;
    params_.mfev = mfev;
    if (params_.bound) {
        params_.bound = 0;
        fprintf(stderr, "Warning [CMA]: box bounding is no longer enabled.\n");
    }
}

void CmaEngine::init(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj)
{
    const bool sep = params_.algo == BBO_ALGO_SEP_CMAES;
    if (sep)
        BBO_REQUIRE(n >= 1 && n <= 4096, "SepCMAES: dimension must be in [1, 4096]");
    else
        BBO_REQUIRE(n >= 1 && n <= EIG_NMAX, "dimension must be in [1, 512]");
    BBO_HIP(hipSetDevice(params_.device));
    obj_ = obj;
    const int P = params_.populations;
    const int lambda = params_.np;
    CmaConst &c = c_;
    c = CmaConst {};
    c.n = n;
    c.ld = round_up(n, 16);
    c.lambda = lambda;
    c.lambda_pad = round_up(lambda, 16);
    c.mu = lambda / 2;
    c.mu_pad = round_up(c.mu, 16);
    c.variant = sep ? 2 : params_.algo == BBO_ALGO_ACTIVE_CMAES ? 1 : 0;
    c.bound = params_.bound ? 1 : 0;
    // (16 < ld <= 128: the samplers there always hand down ||z||^2, so without a box nothing in
    // a generation reads C^-1/2 but cma_paths, which can work from B and D)
    c.lazy_isc = (!sep && !c.bound && c.ld > 16 && c.ld <= 256) ? 1 : 0;
    c.obj = obj.on_device() ? obj.builtin : OBJ_HOST;
    c.mfev = params_.mfev;
    c.mit = params_.mfev / lambda;
    c.npop = P;
    c.seed = params_.seed;
    c.tol = params_.tol;
    c.sigma0 = params_.sigma0;
    c.stop_off = 0;
    c.ftarget = -std::numeric_limits<double>::infinity();

    // recombination weights, base_cmaes.cpp:92-105
    std::vector<double> w(c.mu);
    double sum = 0.;
    for (int i = 0; i < c.mu; i++) {
        w[i] = std::log(0.5 * (lambda + 1.)) - std::log(i + 1.);
        sum += w[i];
    }
    const double inv = 1. / sum;
    for (int i = 0; i < c.mu; i++) w[i] *= inv;
    double lenw = 0.;
    for (int i = 0; i < c.mu; i++) lenw = lenw + w[i] * w[i];
    c.mueff = 1. / lenw;

    // base_cmaes.cpp:108-117
    c.chi = std::sqrt(n) * (1. - 1. / (4. * n) + 1. / (21. * n * n));
    c.cc = (4. + c.mueff / n) / (n + 4. + 2. * c.mueff / n);
    c.cs = (c.mueff + 2.) / (5. + n + c.mueff);
    c.c1 = 2. / ((1.3 + n) * (1.3 + n) + c.mueff);
    c.cmu = std::min(1. - c.c1,
            2. * (c.mueff - 2. + 1. / c.mueff) / ((2. + n) * (2. + n) + c.mueff));
    c.damps = 1. + c.cs + 2. * std::max(0., std::sqrt((c.mueff - 1.) / (n + 1.)) - 1.);
    c.hlen = 10 + (int) std::ceil((30. * n) / lambda);
    c.ik = (int) std::ceil(0.1 + lambda / 4.);
    // cmaes.cpp:48
    c.eigenfreq = params_.eigenrate * lambda / (c.c1 + c.cmu) / n;
    c.cm = 1.;
    c.alphaold = 0.5;
    c.cneg = 0.;
    if (c.variant == 1) {
        // active_cmaes.cpp:48-64
        const double ac = params_.alphacov;
        c.cc = (4. + 0. * c.mueff / n) / (n + 4. + 0. * 2. * c.mueff / n);
        c.cs = (c.mueff + 2.) / (3. + n + c.mueff);
        c.c1 = ac * std::min(1., lambda / 6.) / ((n + 1.3) * (n + 1.3) + c.mueff);
        c.cmu = 1. - c.c1;
        c.cmu = std::min(c.cmu,
                ac * (c.mueff - 2. + 1. / c.mueff) / ((2. + n) * (2. + n) + ac * c.mueff / 2.));
        c.cneg = (1. - c.cmu) * (ac / 8.) * c.mueff / (std::pow(n + 2., 1.5) + 2. * c.mueff);
        c.damps = 1. + c.cs + 2. * std::max(0., std::sqrt((c.mueff - 1.) / (n + 1.)) - 1.);
        c.eigenfreq = params_.eigenrate * (1. / (c.c1 + c.cmu + c.cneg)) / n;
    }

    if (sep) {
        // sep_cmaes.cpp:46-62
        c.cc = 4. / (n + 4.);
        c.cs = (c.mueff + 2.) / (3. + n + c.mueff);
        c.damps = 1. + c.cs + 2. * std::max(0., std::sqrt((c.mueff - 1.) / (n + 1.)) - 1.);
        c.ccov = 2. / ((n + std::sqrt(2.)) * (n + std::sqrt(2.)) * c.mueff);
        c.ccov += std::min(1., (2. * c.mueff - 1.) / ((n + 2.) * (n + 2.) + c.mueff))
                * (1. - 1. / c.mueff);
        if (params_.adjustlr) c.ccov *= ((n + 2.) / 3.);
    }

    // Gram split-K geometry: cma_gram holds its slab (rps rows of ldy doubles) in LDS
    c.rps = 64;
    while (c.rps > 16 && gram_lds_bytes(c.ld, c.rps) > GRAM_LDS_MAX) c.rps >>= 1;
    if (c.ld == 128) {
        // cma_gram128 streams its slab: size the slabs for ~512 workgroups over all populations
        // (two resident per CU: one round; 1024 cost 1 % more in the Gram kernel and a quarter more
        // in cma_cov, which sums the slabs)
        int cap = 32;
        if (const char *e = std::getenv("BBO_GRAM_WANT")) cap = std::max(1, std::atoi(e));     // (tuning)
        int want = std::max(1, std::min(cap, (512 + P - 1) / P));
        want = std::min(want, (c.lambda_pad + G128_CH - 1) / G128_CH);
        c.rps = ((c.lambda_pad + want - 1) / want + G128_CH - 1) / G128_CH * G128_CH;
    }
    c.splits = (c.lambda_pad + c.rps - 1) / c.rps;
    if (sep)   // slabs of the mu selected ranks: enough workgroups to stream at HBM rate
        c.splits = std::max(1, std::min(c.mu / 16, (1024 + P - 1) / P));

    // ---- HBM state ------------------------------------------------------------
    const size_t ld = c.ld, ld2 = ld * ld;
    const bool same_shape = !sep && keep_bc_ && last_n_ == n && C_.count == P * ld2;
    X_.alloc((size_t) P * c.lambda_pad * ld);
    f_.alloc((size_t) P * c.lambda_pad);
    zn2_.alloc((size_t) P * c.lambda_pad);
    rank_.alloc((size_t) P * c.lambda_pad);
    order_.alloc((size_t) P * c.lambda_pad);
    xmean_.alloc(P * ld);
    xold_.alloc(P * ld);
    pc_.alloc(P * ld);
    ps_.alloc(P * ld);
    D_.alloc(P * ld);
    csep_.alloc(P * ld);
    if (!sep) {
        isc_.alloc(P * ld2);
        BDp_.alloc(P * ld2);
        ISp_.alloc(P * ld2);
    }
    S_.alloc((size_t) P * c.mu_pad);
    gram_part_.alloc((size_t) P * c.splits * (sep ? ld : ld2));   // sep: second moments, [ld]
    mean_part_.alloc((size_t) P * c.splits * ld);
    hist_best_.alloc((size_t) P * c.hlen);
    hist_kth_.alloc((size_t) P * c.hlen);
    weights_.alloc(c.mu);
    weights_.upload(w.data(), c.mu);
    lower_.alloc(ld);
    upper_.alloc(ld);
    aux_.alloc(ld);
    scal_.alloc(P);
    zinject_.release();
    zrecord_.release();

    lower_h_.assign(ld, 0.);
    upper_h_.assign(ld, 0.);
    aux_h_.assign(ld, 0.);
    std::copy(lower, lower + n, lower_h_.begin());
    std::copy(upper, upper + n, upper_h_.begin());
    fill_objective_aux(obj.on_device() ? obj.builtin : -1, n, aux_h_.data());
    lower_.upload(lower_h_.data(), ld);
    upper_.upload(upper_h_.data(), ld);
    aux_.upload(aux_h_.data(), ld);

    // B = C = C^-1/2 = I, D = 1.  The reference resize()s _b/_c, so on a re-init of the
    // SAME object with the same n the old off-diagonals survive and only the diagonals are
    // reset (cmaes.cpp:53-59); restart drivers depend on that, so it is kept.
    std::vector<double> eye(sep ? 0 : P * ld2, 0.), ones(P * ld, 1.);
    csep_.upload(ones.data(), P * ld);
    if (!sep) {
    for (int p = 0; p < P; p++)
        for (int i = 0; i < n; i++) eye[p * ld2 + (size_t) i * ld + i] = 1.;
    if (same_shape) {
        std::vector<double> bm(P * ld2), cm(P * ld2);
        B_.download(bm.data(), P * ld2);
        C_.download(cm.data(), P * ld2);
        for (int p = 0; p < P; p++)
            for (int i = 0; i < n; i++) {
                bm[p * ld2 + (size_t) i * ld + i] = 1.;
                cm[p * ld2 + (size_t) i * ld + i] = 1.;
            }
        B_.upload(bm.data(), P * ld2);
        C_.upload(cm.data(), P * ld2);
    } else {
        B_.alloc(P * ld2);
        C_.alloc(P * ld2);
        B_.upload(eye.data(), P * ld2);
        C_.upload(eye.data(), P * ld2);
    }
    isc_.upload(eye.data(), P * ld2);
    }   // !sep
    D_.upload(ones.data(), P * ld);
    keep_bc_ = !sep;
    last_n_ = n;

    std::vector<double> xm(P * ld, 0.);
    for (int p = 0; p < P; p++) std::copy(guess + (size_t) p * n, guess + (size_t) (p + 1) * n,
            xm.begin() + p * ld);
    xmean_.upload(xm.data(), P * ld);

    std::vector<CmaScal> sc(P);
    for (auto &s : sc) {
        std::memset(&s, 0, sizeof(s));
        s.sigma = params_.sigma0;
        s.fbest = -std::numeric_limits<double>::infinity();
        s.fworst = std::numeric_limits<double>::infinity();
        s.hist_head = -1;
        // a re-initialised object keeps the off-diagonals of B (cmaes.cpp:53-59) while C^-1/2
        // restarts from I: the two disagree until the first decomposition
        s.basis_ok = same_shape ? 0 : 1;
    }
    scal_.upload(sc.data(), P);
    basis_maybe_stale_ = same_shape;

    // the eigensolver keeps its matrix in LDS when it fits
    // global scratch of the eigensolver: the work matrix when it does not fit LDS, or
    // (divide and conquer) the Householder matrix and the merge factor
    if (!sep) eig_work_.alloc((size_t) P * 4 * eig_slab((int) ld));

    CmaDev &d = d_;
    d = CmaDev {};
    d.X = X_.p; d.f = f_.p; d.rank = rank_.p; d.order = order_.p;
    d.xmean = xmean_.p; d.xold = xold_.p; d.pc = pc_.p; d.ps = ps_.p;
    d.C = C_.p; d.B = B_.p; d.D = D_.p; d.isc = isc_.p; d.BDp = BDp_.p; d.ISp = ISp_.p;
    d.S = S_.p; d.zn2 = zn2_.p; d.csep = csep_.p; d.gram_part = gram_part_.p; d.mean_part = mean_part_.p;
    d.hist_best = hist_best_.p; d.hist_kth = hist_kth_.p; d.eig_work = eig_work_.p;
    d.weights = weights_.p; d.lower = lower_.p; d.upper = upper_.p; d.aux = aux_.p;
    d.zinject = nullptr; d.zrecord = nullptr; d.scal = scal_.p;
    d.stamps = stamps_.p;
    d.mw_fail_host = mw_fail_host_;
    d.mw_fault = mw_fault_from_env();
    mw_release();          // (the shape may have changed: reserved again at the first spread launch)

    // packed operands of the initial B, D, C^-1/2
    c.honor_stop = 0;
    inited_ = true;
    if (!sep) {
        launch_post(2);
        BBO_HIP(hipGetLastError());
        BBO_HIP(hipStreamSynchronize(stream_));
    }
}

// ---- kernel launches ---------------------------------------------------------------
void CmaEngine::launch_post(int mode)
{
    const CmaConst &c = c_;
    if (c.ld <= 128) {
        const size_t lds = (size_t) (c.ld * (c.ld + 2) + c.ld) * sizeof(double);
        allow_lds((const void*) cma_post_mfma<1>, 140 * 1024);
        allow_lds((const void*) cma_post_mfma<4>, 140 * 1024);
        // few populations: four workgroups each (latency); many: one (no redundant staging)
        if (c.npop < 32)
            hipLaunchKernelGGL(cma_post_mfma<4>, dim3(c.npop, 4), dim3(256), lds, stream_, d_, c_,
                    mode);
        else
            hipLaunchKernelGGL(cma_post_mfma<1>, dim3(c.npop, 1), dim3(256), lds, stream_, d_, c_,
                    mode);
    } else {
        dim3 grid(c.ld / 16, c.ld / 16, c.npop);
        hipLaunchKernelGGL(cma_post, grid, dim3(256), 0, stream_, d_, c_, mode);
    }
}

void CmaEngine::launch_sample_eval()
{
    const CmaConst &c = c_;
    bool zn_valid = false;
    timer_.begin(stream_, K_SAMPLE);
    if (c.variant == 2) {
        // separable: 16 lanes per candidate for short rows, one wavefront per candidate beyond
        // (a workgroup stages the generator's table once and then walks chunks of candidates:
        // ~4096 workgroups in all, 16 per CU)
        const int per_pop = std::max(1, 4096 / c.npop);
        if (c.ld <= 256) {   // (beyond that the 16-row LDS tile would leave one workgroup per CU)
            allow_lds((const void*) sep_sample_eval<16>, 128 * 1024);
            const size_t lds = (size_t) 16 * c.ld * sizeof(double);
            hipLaunchKernelGGL(sep_sample_eval<16>,
                    dim3(std::min((c.lambda_pad + 15) / 16, per_pop), c.npop), dim3(256), lds,
                    stream_, d_, c_);
        } else if (c.ld <= 2048) {
            // 8 rows per workgroup of 512: rows + table leave room for two workgroups (16
            // wavefronts) per CU up to ld = 1024
            const size_t lds = (size_t) 8 * c.ld * sizeof(double);
            const dim3 grid(std::min((c.lambda_pad + 7) / 8, per_pop), c.npop);
            if (c.n == c.ld && !c.bound && c.lambda == c.lambda_pad && !d_.zinject
                    && !d_.zrecord && !(d_.dbg & 256)) {   // nothing to guard (the benchmark's SEP)
                if (sep_sum_objective(c.obj) && !(d_.dbg & 1048576)) {
                    // sums of per-coordinate terms: no row in LDS (sep_sample_sum); 256-thread
                    // workgroups, as many rows in flight as the registers allow
                    const dim3 sgrid(std::min(c.lambda_pad / (4 * SEP_K), per_pop), c.npop);
#define BBO_SEP_SUM(NCV, OBJV) hipLaunchKernelGGL((sep_sample_sum<256, NCV, OBJV>), sgrid, dim3(256), 0, stream_, d_, c_)
#define BBO_SEP_SUM_OBJ(NCV) \
                    switch (c.obj) { \
                    case OBJ_SPHERE: BBO_SEP_SUM(NCV, OBJ_SPHERE); break; \
                    case OBJ_ELLIPSOID: BBO_SEP_SUM(NCV, OBJ_ELLIPSOID); break; \
                    case OBJ_RASTRIGIN: BBO_SEP_SUM(NCV, OBJ_RASTRIGIN); break; \
                    case OBJ_CIGAR: BBO_SEP_SUM(NCV, OBJ_CIGAR); break; \
                    case OBJ_DISCUS: BBO_SEP_SUM(NCV, OBJ_DISCUS); break; \
                    default: BBO_SEP_SUM(NCV, OBJ_DIFFPOW); break; \
                    }
                    // (the cosine / pow objectives inline a long body per coordinate: with the
                    // lane's 32 constants resident as well they spill or fall to one wavefront
                    // per SIMD -- Rastrigin took 512 registers and ran slower than from LDS rows)
                    if (c.ld == 1024 && c.obj != OBJ_RASTRIGIN && c.obj != OBJ_DIFFPOW) { BBO_SEP_SUM_OBJ(4) }
                    else { BBO_SEP_SUM_OBJ(0) }
#undef BBO_SEP_SUM_OBJ
#undef BBO_SEP_SUM
                } else {
                    allow_lds((const void*) sep_sample_eval<64, 512, true>, 140 * 1024);
                    hipLaunchKernelGGL((sep_sample_eval<64, 512, true>), grid, dim3(512), lds,
                            stream_, d_, c_);
                }
            } else {
                allow_lds((const void*) sep_sample_eval<64, 512>, 140 * 1024);
                hipLaunchKernelGGL((sep_sample_eval<64, 512>), grid, dim3(512), lds, stream_,
                        d_, c_);
            }
        } else {
            allow_lds((const void*) sep_sample_eval<64>, 140 * 1024);
            const size_t lds = (size_t) 4 * c.ld * sizeof(double);
            hipLaunchKernelGGL(sep_sample_eval<64>,
                    dim3(std::min((c.lambda_pad + 3) / 4, per_pop), c.npop), dim3(256), lds,
                    stream_, d_, c_);
        }
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
        c_.use_zn = 0;
        return;
    }
    if (c.ld == 128 && (long) c.npop * c.lambda_pad >= sample128_min_rows_
            && (c.obj < 0 || frag_objective_ok(c.obj))) {
        // whole populations in flight: packed operand in LDS, normals drawn into the A fragments
        // one workgroup per CU when the populations allow it: long tile loops amortise the fill
        int rw = (int) (((long) c.npop * c.lambda_pad / 256 + 127) / 128) * 128;
        rw = std::max(128, std::min(4096, rw));
        allow_lds((const void*) cma_sample_eval128, 128 * 1024);
        dim3 grid((c.lambda_pad + rw - 1) / rw, c.npop);
        // the lean build of the tile loop where nothing needs guarding (M, C3)
        const int full = (c.n == 128 && !c.bound && c.lambda == c.lambda_pad && !d_.zinject
                && !d_.zrecord && !(d_.dbg & 256)) ? 1 : 0;
        hipLaunchKernelGGL(cma_sample_eval128, grid, dim3(512), 128 * 1024, stream_, d_, c_, rw,
                full);
        zn_valid = true;
    } else if (c.ld == 128 && (long) c.npop * (c.lambda_pad / 16) <= sample_wide_max_tiles_) {
        // one population at a time: a 16-row tile per WORKGROUP, one column tile per wavefront (the
        // form C5's handful of candidates takes at n = 256) -- a wavefront's chain is 32 MFMAs and
        // one Philox call instead of 256 and eight
        const size_t lds = (size_t) 16 * (c.ld + 2) * sizeof(double);
        hipLaunchKernelGGL((cma_sample_eval<1, 8>), dim3(c.lambda_pad / 16, c.npop), dim3(512), lds, stream_,
                d_, c_);
        zn_valid = true;
    } else if (c.ld <= 128) {
        // 64 candidates per workgroup, packed operand held in registers
        dim3 grid((c.lambda_pad + 63) / 64, c.npop);
        const size_t lds = (size_t) 64 * (c.ld + 2) * sizeof(double);
        allow_lds((const void*) cma_sample_eval64<1>, 80 * 1024);
        allow_lds((const void*) cma_sample_eval64<2>, 80 * 1024);
        if (c.ld <= 32)      // (8 k-steps of operand in registers instead of 32: twice the occupancy)
            hipLaunchKernelGGL((cma_sample_eval64<1, 8>), grid, dim3(256), lds, stream_, d_, c_);
        else if (c.ld <= 64)
            hipLaunchKernelGGL(cma_sample_eval64<1>, grid, dim3(256), lds, stream_, d_, c_);
        else
            hipLaunchKernelGGL(cma_sample_eval64<2>, grid, dim3(256), lds, stream_, d_, c_);
        zn_valid = true;
    } else {
        dim3 grid(c.lambda_pad / 16, c.npop);
        const size_t lds = (size_t) 16 * (c.ld + 2) * sizeof(double);
        allow_lds((const void*) cma_sample_eval<8>, 80 * 1024);   // ld = 512: 65 792 bytes
        // a handful of row tiles on the whole chip (C5: two): one column tile per wavefront
        if ((long) grid.x * grid.y <= 32 && c.ld <= 256 && !(d_.dbg & 268435456))
            hipLaunchKernelGGL((cma_sample_eval<1, 16>), grid, dim3(1024), lds, stream_, d_, c_);
        else
        switch (pick_maxt(c.ld)) {
        case 1: hipLaunchKernelGGL(cma_sample_eval<1>, grid, dim3(256), lds, stream_, d_, c_); break;
        case 2: hipLaunchKernelGGL(cma_sample_eval<2>, grid, dim3(256), lds, stream_, d_, c_); break;
        case 4: hipLaunchKernelGGL(cma_sample_eval<4>, grid, dim3(256), lds, stream_, d_, c_); break;
        default: hipLaunchKernelGGL(cma_sample_eval<8>, grid, dim3(256), lds, stream_, d_, c_); break;
        }
        zn_valid = true;
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    // ||z||^2 stands in for the whitened norm only if this launch wrote it and no x was clamped
    c_.use_zn = (zn_valid && !c.bound) ? 1 : 0;
}

void CmaEngine::launch_rank()
{
    const CmaConst &c = c_;
    rank_wrote_norms_ = false;
    timer_.begin(stream_, K_RANK);
    // few populations: the counting kernel spreads one ranking over many CUs; many
    // populations: one in-LDS sort per population is far less work in total
    if (c.lambda <= 64 && c.lambda >= 2 && c.npop >= 4) {
        hipLaunchKernelGGL(cma_rank_wave, dim3((c.npop + 3) / 4), dim3(256), 0, stream_, d_, c_);
    } else if (c.lambda <= SORT_LDS_MAX && c.npop >= 4) {
        int m = 2;
        while (m < c.lambda) m <<= 1;
        allow_lds((const void*) cma_rank_sort, 128 * 1024);
        const size_t lds = rank_sort_merges(m, d_.dbg) ? (size_t) m * 24 : (size_t) std::max(m, 1024) * 12;
        hipLaunchKernelGGL(cma_rank_sort, dim3(c.npop), dim3(sort_threads(m)), lds, stream_, d_,
                c_, m);
        // the whitened norms of the worst mu, where they are the sampler's sigma^2 ||z||^2 handed
        // round through the ranking (cma_whiten128's shortcut): written by the sort itself, which
        // has the ranking in LDS -- a launch less per generation (10 us of the M step)
        rank_wrote_norms_ = c.variant == 1 && c.use_zn && !basis_maybe_stale_;
    } else {
        // few candidates in flight: 8 per workgroup, 32 slices each (a quarter of the 64-bit compares
        // per thread; diagnostic bit 128 keeps the 32-candidate form -- the same counts)
        if ((long) c.npop * ((c.lambda + 31) / 32) <= 256 && c.lambda >= 2048 && !(d_.dbg & (128 | 4096)))
            hipLaunchKernelGGL(cma_rank64, dim3((c.lambda + 3) / 4, c.npop), dim3(256), 0, stream_, d_, c_);
        else if ((long) c.npop * ((c.lambda + 31) / 32) <= 512 && c.lambda >= 512 && !(d_.dbg & 128))
            hipLaunchKernelGGL(cma_rank32, dim3((c.lambda + 7) / 8, c.npop), dim3(256), 0, stream_, d_, c_);
        else
            hipLaunchKernelGGL(cma_rank, dim3((c.lambda + 31) / 32, c.npop), dim3(256), 0, stream_, d_, c_);
        // (the whitened norms leave with the ranking here too: see cma_rank_body)
        rank_wrote_norms_ = c.variant == 1 && c.use_zn && !basis_maybe_stale_;
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

void CmaEngine::launch_update()
{
    const CmaConst &c = c_;
    if (c.variant == 2) {
        timer_.begin(stream_, K_GRAM);
        hipLaunchKernelGGL(sep_moments, dim3(c.splits, (c.ld + 511) / 512, c.npop), dim3(256), 0,
                stream_, d_, c_);
        timer_.end(stream_);
        timer_.begin(stream_, K_PATHS);
        hipLaunchKernelGGL(sep_paths, dim3(c.npop), dim3(256), 0, stream_, d_, c_);
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
        return;
    }
    const bool norms_done = rank_wrote_norms_;
    rank_wrote_norms_ = false;             // (one ranking serves one update)
    if (c.variant == 1 && !norms_done) {
        timer_.begin(stream_, K_WHITEN);
        if (c.ld == 128 && (long) c.npop * c.mu_pad >= 256 * 128) {
            int rw = (int) (((long) c.npop * c.mu_pad / 256 + 127) / 128) * 128;
            rw = std::max(128, std::min(2048, rw));
            const size_t lds = (size_t) (128 * 128 + 128) * sizeof(double);
            allow_lds((const void*) cma_whiten128, 132 * 1024);
            dim3 grid128((c.mu_pad + rw - 1) / rw, c.npop);
            hipLaunchKernelGGL(cma_whiten128, grid128, dim3(512), lds, stream_, d_, c_, rw);
        } else {
        dim3 grid(c.mu_pad / 16, c.npop);
        const size_t lds = (size_t) (16 * (c.ld + 2) + 64) * sizeof(double);
        allow_lds((const void*) cma_whiten<8>, 80 * 1024);        // ld = 512: 66 304 bytes
        switch (pick_maxt(c.ld)) {
        case 1: hipLaunchKernelGGL(cma_whiten<1>, grid, dim3(256), lds, stream_, d_, c_); break;
        case 2: hipLaunchKernelGGL(cma_whiten<2>, grid, dim3(256), lds, stream_, d_, c_); break;
        case 4: hipLaunchKernelGGL(cma_whiten<4>, grid, dim3(256), lds, stream_, d_, c_); break;
        default: hipLaunchKernelGGL(cma_whiten<8>, grid, dim3(256), lds, stream_, d_, c_); break;
        }
        }
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
    }
    if (c.ld == 128) {
        const size_t lds = (size_t) (2 * G128_CH * G128_LDY + 4 * G128_CH) * sizeof(double);
        allow_lds((const void*) cma_gram128, 80 * 1024);
        timer_.begin(stream_, K_GRAM);
        if (d_.dbg & 512)   // (diagnostic: the LDS-staged form, same bits)
            hipLaunchKernelGGL(cma_gram128, dim3(c.splits, c.npop), dim3(256), lds, stream_, d_, c_);
        else
            hipLaunchKernelGGL(cma_gram128s, dim3(c.splits, c.npop), dim3(256), 0, stream_, d_, c_);
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
    } else {
        const int NT = c.ld / 16, LT = NT * (NT + 1) / 2;
        const int ldy = gram_ldy(c.ld);
        dim3 grid(c.splits, (LT + 4 * GRAM_TPW - 1) / (4 * GRAM_TPW), c.npop);
        const size_t lds = gram_lds_bytes(c.ld, c.rps);
        allow_lds((const void*) cma_gram, (int) GRAM_LDS_MAX);
        timer_.begin(stream_, K_GRAM);
        hipLaunchKernelGGL(cma_gram, grid, dim3(256), lds, stream_, d_, c_, ldy);
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
    }
    timer_.begin(stream_, K_PATHS);
    if (c.lazy_isc && c.n >= 64 && !(d_.dbg & 131072))
        // (1024 threads: a quarter of the dependent round trips of the two passes over B)
        hipLaunchKernelGGL(cma_paths_lazy1k, dim3(c.npop), dim3(1024), 0, stream_, d_, c_);
    else if (c.lazy_isc)
        hipLaunchKernelGGL(cma_paths_lazy, dim3(c.npop), dim3(256), 0, stream_, d_, c_);
    else
        hipLaunchKernelGGL(cma_paths, dim3(c.npop), dim3(256), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    {
        const int total = c.n * (c.n + 1) / 2;
        dim3 grid((total + 255) / 256, c.npop);
        timer_.begin(stream_, K_COV);
        hipLaunchKernelGGL(cma_cov, grid, dim3(256), 0, stream_, d_, c_);
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
    }
}

int CmaEngine::next_mw_xcd()
{
    static std::atomic<int> counter { 0 };
    return counter.fetch_add(1) & 7;
}

void CmaEngine::launch_eigen()
{
    const CmaConst &c = c_;
    if (c.variant == 2) return;        // diagonal covariance: d = sqrt(c) is part of sep_paths
    // 64 < n <= 128 with few matrices in flight (one optimisation run at a time): the reduction on
    // one workgroup, the divide and conquer and the reflectors over many -- the structure of
    // 128 < n <= 256 (diagnostic bit 4194304: everything on the reducing workgroup, as for a batch)
    const EigPlan pl_lds = eig_plan(c.n, c.ld);
    const bool split128 = pl_lds.use_lds && pl_lds.threads == 512 && pl_lds.dc && c.npop <= split_maxp_
            && !(d_.dbg & (2 | 4 | 8 | 1024 | 4194304));
    const EigPlan pl = split128 ? eig_plan_split(c.n, c.ld) : pl_lds;
    bool wy4_packs = false, fcols_closed = false;
    allow_lds((const void*) cma_eigen, 160 * 1024 - 768);
    allow_lds((const void*) cma_eigen_g, 160 * 1024 - 768);
    allow_lds((const void*) cma_eigen_b, 160 * 1024 - 768);
    allow_lds((const void*) cma_eigen_256, 160 * 1024 - 768);
    allow_lds((const void*) cma_eigen_128, 160 * 1024 - 768);
    timer_.begin(stream_, K_EIGEN);
    // (256 < n <= 512 by the spread reduction: while its 16 workgroups per matrix fit the chip at once)
    const bool big_spread = c.n > 256 && pl.dc && !pl.hybrid && !mw_disabled_
            && !(d_.dbg & (2 | 16777216)) && mw_reserve((long) c.npop * 16);
    // n <= 16: a wavefront per matrix (dbg bit 4 keeps the big kernel)
    const bool small = c.n <= 16 && c.n >= 2 && c.ld == 16 && !(d_.dbg & 16);
    if (small)    // (does cma_post's work too: one launch less where launches are what costs)
        hipLaunchKernelGGL(cma_eigen_small, dim3((c.npop + 3) / 4), dim3(256), 0, stream_, d_, c_, 0,
                1);
    else if (pl.threads == 128)   // four lanes per row: smaller matrices, smaller workgroups,
        hipLaunchKernelGGL(cma_eigen_128, dim3(c.npop), dim3(128), pl.lds_bytes, stream_, d_, c_,
                pl, 0);             // several of them per CU
    else if (pl.threads == 256)
        hipLaunchKernelGGL(cma_eigen_256, dim3(c.npop), dim3(256), pl.lds_bytes, stream_, d_, c_,
                pl, 0);
    else if (pl.use_lds)
        hipLaunchKernelGGL(cma_eigen, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_,
                pl, 0);
    else if (pl.hybrid && !(d_.dbg & (2 | 1024 | 4194304))) {
        // 128 < n <= 256: reduction, the two halves side by side (eigenvector blocks in LDS), top
        // merge -- three launches (diagnostic bit 4194304: everything in one workgroup, round 3)
        const EigPlan plh = eig_plan(128, 128);
        allow_lds((const void*) cma_eigen_g1, 160 * 1024 - 768);
        allow_lds((const void*) cma_eigen_g2, 160 * 1024 - 768);
        allow_lds((const void*) cma_eig_halves, 160 * 1024 - 768);
        // the reduction: its first n - 128 steps spread over MW_G workgroups per matrix (2.8 us per
        // step, an exchange between compute units each, where the one-workgroup step with the whole
        // active matrix on chip costs ~5), the leading 128 x 128 block then on one workgroup
        // (1.15 us per step): 0.35 against 0.36 ms per decomposition at n = 132, 0.63 / 0.73 at 200,
        // 0.83 / 1.04 at 256 -- while all of a launch's workgroups (256 threads, 277 registers per
        // lane: ONE per compute unit) can be resident at once next to those of the process's other
        // engines (they wait for each other: MwBudget above, bbo_eig_mw.hpp; diagnostic bit 16777216
        // keeps the reduction on one workgroup)
        const bool use_mw = !split128 && !mw_disabled_ && !(d_.dbg & 16777216) && mw_reserve((long) c.npop * MW_G);
        if (split128) {
            allow_lds((const void*) cma_eigen_r1, 160 * 1024 - 768);
            hipLaunchKernelGGL(cma_eigen_r1, dim3(c.npop), dim3(512), pl_lds.lds_bytes, stream_, d_, c_,
                    pl_lds, 0);
        } else if (use_mw) {
            mw_launched_ = true;
            if (mw_buf_.count != (size_t) c.npop * MW_BUF_DOUBLES) mw_buf_.alloc((size_t) c.npop * MW_BUF_DOUBLES);
            // (its steps down to the leading 128 x 128 block; that block on one workgroup: diagnostic
            // bit 536870912 keeps all steps spread)
            const int istop = (d_.dbg & 536870912) ? 1 : 128;
            hipLaunchKernelGGL(cma_tred_mw, dim3(8 * MW_G, c.npop), dim3(MW_T), 0, stream_, d_, c_, 0,
                    mw_buf_.p, ++mw_launch_, istop, mw_xcd_);
            if (istop > 1) {
                allow_lds((const void*) cma_tred_tail, 160 * 1024 - 768);
                hipLaunchKernelGGL(cma_tred_tail, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_, pl, 0);
            }
        } else
        hipLaunchKernelGGL(cma_eigen_g1, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_, pl, 0);
        hipLaunchKernelGGL(cma_eig_halves, dim3(3, c.npop), dim3(512), plh.lds_bytes, stream_, d_, c_,
                plh, pl.lda);
        // the top merge: with few matrices in flight its secular equation goes to ceil(n / 32)
        // workgroups of its own between the two parts (diagnostic bit 67108864: one kernel)
        if ((long) c.npop * 8 <= 256 && !(d_.dbg & 67108864)) {
            hipLaunchKernelGGL(cma_eigen_g2, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_, pl, 1);
            hipLaunchKernelGGL(cma_eig_secular, dim3((c.n + 31) / 32, c.npop), dim3(512), 0, stream_, d_, c_);
            // behind the secular equation: the Loewner vector and the columns of F on n / 32 workgroups
            // each, the closing repair / root with them (even n: the T factors were built beside the
            // halves; diagnostic bit 33554432: cma_eigen_g2 part 2, one workgroup, as in round 4)
            fcols_closed = !(c.n & 1) && c.lazy_isc && (long) c.npop * ((c.n + 15) / 16) <= 256
                    && !(d_.dbg & (33554432 | 134217728));
            if (fcols_closed) {
                hipLaunchKernelGGL(cma_eig_lowner, dim3((c.n + 31) / 32, c.npop), dim3(512), 0, stream_, d_, c_);
                hipLaunchKernelGGL(cma_eig_fcols, dim3((c.n + 31) / 32, c.npop), dim3(512), 0, stream_, d_, c_);
            } else
            hipLaunchKernelGGL(cma_eigen_g2, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_, pl, 2);
        } else
        hipLaunchKernelGGL(cma_eigen_g2, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_, pl, 0);
    } else if (pl.hybrid)
        hipLaunchKernelGGL(cma_eigen_g, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_,
                pl, 0);
    else if (big_spread) {
        // 256 < n <= 512, few matrices: the structure of 128 < n <= 256 -- the reduction's first
        // n - 128 steps spread over 16 workgroups (cma_tred_mw512), the leading block on one
        // (cma_tred_tail), the reflectors stashed -- in front of the same divide and conquer
        // (two spread kernels: rows of 512 on 16 workgroups down to pivot row 256, 5.5 us per step,
        // then rows of 256 on 8, 2.9 us per step -- each with exchange buffers of its own: a value
        // one of them publishes must never sit where the other looks for a flag; diagnostic bit
        // 536870912 keeps the first one down to row 128)
        const bool chain = !(d_.dbg & 536870912);
        mw_launched_ = true;
        const size_t need = (size_t) c.npop * (mw_buf_doubles(512) + mw_buf_doubles(256));
        if (mw_buf_.count != need) mw_buf_.alloc(need);
        hipLaunchKernelGGL(cma_tred_mw512, dim3(8 * 16, c.npop), dim3(MW_T), 0, stream_, d_, c_, 0,
                mw_buf_.p, ++mw_launch_, chain ? 256 : 128, mw_xcd_);
        if (chain)
            hipLaunchKernelGGL(cma_tred_mw_chain, dim3(8 * MW_G, c.npop), dim3(MW_T), 0, stream_, d_, c_,
                    mw_buf_.p + (size_t) c.npop * mw_buf_doubles(512), ++mw_launch_, 128, mw_xcd_);
        const EigPlan plt = eig_plan(256, 256);        // (the tail's LDS: vectors + a 128 x 130 matrix)
        allow_lds((const void*) cma_tred_tail, 160 * 1024 - 768);
        hipLaunchKernelGGL(cma_tred_tail, dim3(c.npop), dim3(512), plt.lds_bytes, stream_, d_, c_, plt,
                chain ? 1 : 0);
        allow_lds((const void*) cma_eigen_b4, 160 * 1024 - 768);
        hipLaunchKernelGGL(cma_eigen_b4, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_, pl, 0);
    } else
        hipLaunchKernelGGL(cma_eigen_b, dim3(c.npop), dim3(512), pl.lds_bytes, stream_, d_, c_,
                pl, 0);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    if (pl.dc && !pl.reg_path && (pl.hybrid || !(d_.dbg & 2))) {
        // n > 128: the top merge's two products as whole-GPU kernels
        dim3 grid((c.n + 63) / 64, (c.n + 63) / 64, c.npop);
        // few populations: 64 x 16 blocks, four times the workgroups (same sums, same order)
        const bool narrow = (long) grid.x * grid.y * grid.z < 128;
        dim3 grid1((c.n + 15) / 16, (c.n + 63) / 64, c.npop);
        if (narrow)
            hipLaunchKernelGGL(cma_eig_gemm1, grid1, dim3(256), 0, stream_, d_, c_, pl.lda, 0);
        else
            hipLaunchKernelGGL(cma_eig_gemm, grid, dim3(256), 0, stream_, d_, c_, pl.lda, 0);
        // second product: the stashed reflectors applied in blocked form (the QL fallback of
        // the diagnostic switch has accumulated Q_house instead)
        if (big_spread)
            hipLaunchKernelGGL(cma_eig_wy4_512, dim3((c.n + 15) / 16, c.npop), dim3(256), 0, stream_, d_, c_);
        else if ((d_.dbg & 2) || !pl.hybrid)     // (n > 256: Q_house was accumulated by the reduction)
            hipLaunchKernelGGL(cma_eig_gemm, grid, dim3(256), 0, stream_, d_, c_, pl.lda, 1);
        else
        {
            // few matrices: a 16-column tile per WORKGROUP, its rows dealt to the four wavefronts
            // (diagnostic bit 134217728 keeps a tile per wavefront)
            if ((long) c.npop * ((c.n + 15) / 16) <= 256 && !(d_.dbg & 134217728)) {
                // (under lazy_isc the packed operand B D leaves with B: no cma_post launch)
                wy4_packs = c.lazy_isc != 0;
                hipLaunchKernelGGL(cma_eig_wy4, dim3((c.n + 15) / 16, c.npop), dim3(256), 0, stream_, d_, c_,
                        wy4_packs ? (fcols_closed ? 2 : 1) : 0);
            } else
            hipLaunchKernelGGL(cma_eig_wy, dim3((c.n + 63) / 64, c.npop), dim3(256), 0, stream_, d_,
                    c_);
        }
        BBO_HIP(hipGetLastError());
    }
    timer_.begin(stream_, K_POST);
    // (lazy_isc: the eigensolver has written the packed B D itself and C^-1/2 is not formed)
    const bool packed_by_eigen = c.lazy_isc && pl.dc && pl.reg_path && !(d_.dbg & 2);
    if (!small && !packed_by_eigen && !wy4_packs) launch_post(0);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

void CmaEngine::launch_history_stop()
{
    timer_.begin(stream_, K_STOP);
    hipLaunchKernelGGL(cma_history_stop, dim3(c_.npop), dim3(64), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

// the compatibility path for arbitrary host objectives: X leaves HBM once per generation
void CmaEngine::host_evaluate()
{
    const CmaConst &c = c_;
    const size_t rows = (size_t) c.npop * c.lambda_pad;
    std::vector<double> xh(rows * c.ld), fh(rows, std::numeric_limits<double>::infinity());
    BBO_HIP(hipStreamSynchronize(stream_));
    X_.download(xh.data(), xh.size());
    std::vector<CmaScal> sc;
    fetch_scal(sc);
    for (int p = 0; p < c.npop; p++) {
        if (c.honor_stop && sc[p].stop) continue;
        const size_t r0 = (size_t) p * c.lambda_pad;
        obj_.eval_host(xh.data() + r0 * c.ld, c.lambda, c.n, c.ld, fh.data() + r0);
        for (int r = 0; r < c.lambda; r++)
            if (fh[r0 + r] != fh[r0 + r]) fh[r0 + r] = std::numeric_limits<double>::infinity();
    }
    f_.upload(fh.data(), rows);
}

// n <= 16, lambda <= 64, on-device objective: whole generations in one launch
// (cma_small_generations).  The per-kernel timers and the diagnostic bit 64 keep the nine-kernel
// path, which computes the same bits.
bool CmaEngine::small_fused_ok() const
{
    const CmaConst &c = c_;
    return c.variant != 2 && c.ld == 16 && c.n >= 2 && c.lambda_pad <= 64 && obj_.on_device()
            && c.npop <= SMALL_FUSED_MAXP && !timer_.on() && !(d_.dbg & (16 | 64));
}

void CmaEngine::launch_small(int gens, bool honor_stop)
{
    c_.honor_stop = honor_stop ? 1 : 0;
    c_.use_zn = c_.bound ? 0 : 1;          // cma_sample_eval64 always hands down ||z||^2
    const int ldy = gram_ldy(c_.ld);
    size_t lds = gram_lds_bytes(c_.ld, c_.rps);
    lds = std::max(lds, (size_t) 64 * (c_.ld + 2) * sizeof(double));
    lds = std::max(lds, (size_t) (16 * (c_.ld + 2) + 64) * sizeof(double));
    const int scratch = (int) ((lds + 15) / 16 * 2);                 // doubles, 16-byte aligned
    const size_t total = ((size_t) scratch + small_state_doubles(c_)) * sizeof(double);
    allow_lds((const void*) cma_small_generations, 120 * 1024);
    BBO_REQUIRE(total <= 120 * 1024, "small fused path: state does not fit LDS");
    hipLaunchKernelGGL(cma_small_generations, dim3(c_.npop), dim3(256), total, stream_, d_, c_,
            gens, ldy, scratch);
    BBO_HIP(hipGetLastError());
}

void CmaEngine::generation(bool honor_stop)
{
    if (small_fused_ok()) {
        launch_small(1, honor_stop);
        return;
    }
    c_.honor_stop = honor_stop ? 1 : 0;
    launch_sample_eval();
    if (!obj_.on_device()) host_evaluate();
    launch_rank();
    launch_update();
    launch_eigen();
    launch_history_stop();
}

void CmaEngine::phase(int which)
{
    BBO_REQUIRE(inited_, "phase before init");
    BBO_HIP(hipSetDevice(params_.device));
    c_.honor_stop = 0;
    d_.mw_fault = mw_fault_from_env();
    switch (which) {
    case BBO_PHASE_SAMPLE_EVALUATE:
        launch_sample_eval();
        if (!obj_.on_device()) host_evaluate();
        break;
    case BBO_PHASE_RANK: launch_rank(); break;
    case BBO_PHASE_UPDATE: launch_update(); break;
    case BBO_PHASE_EIGEN: launch_eigen(); break;
    case BBO_PHASE_HISTORY_STOP: launch_history_stop(); break;
    default: throw Error(BBO_ERR_ARG, "unknown CMA phase");
    }
    BBO_HIP(hipStreamSynchronize(stream_));
    if (mw_check_failed() && which == BBO_PHASE_EIGEN) {
        // the spread reduction gave up: this generation's decomposition by the one-workgroup path
        launch_eigen();
        BBO_HIP(hipStreamSynchronize(stream_));
    }
    timer_.collect();
}

void CmaEngine::inject_normals(const double *z, int count)
{
    BBO_REQUIRE(inited_, "inject_normals before init");
    rank_wrote_norms_ = false;
    if (!z) {
        d_.zinject = nullptr;
        return;
    }
    const size_t want = (size_t) c_.npop * c_.lambda * c_.n;
    BBO_REQUIRE((size_t) count == want, "inject_normals: count must be populations*lambda*n");
    if (zinject_.count != want) zinject_.alloc(want);
    zinject_.upload(z, want);
    d_.zinject = zinject_.p;
}

void CmaEngine::iterate()
{
    if (!inited_) throw Error(BBO_ERR_STATE, "iterate() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    generation(false);
    BBO_HIP(hipStreamSynchronize(stream_));
    if (mw_check_failed()) {
        // the spread reduction gave up (bbo_eig_mw.hpp): the decomposition this generation was due
        // is not lost -- eigenlastev has not moved, the one-workgroup kernels take it now
        launch_eigen();
        BBO_HIP(hipStreamSynchronize(stream_));
    }
    timer_.collect();
}

void CmaEngine::fetch_scal(std::vector<CmaScal> &out)
{
    out.resize(c_.npop);
    scal_.download(out.data(), c_.npop);
}

bool CmaEngine::all_stopped()
{
    std::vector<CmaScal> sc;
    fetch_scal(sc);
    // (the one host-side copy of "every population's C^-1/2 matches its (B, D)": see launch_rank)
    bool stale = false;
    for (const auto &s : sc) stale = stale || !s.basis_ok;
    basis_maybe_stale_ = stale;
    // (bbo_eig_mw.hpp: back to the one-workgroup reduction; the generations since the time-out
    // returned from the spread kernel at entry, the next one decomposes -- eigenlastev stood still)
    mw_check_failed();
    for (const auto &s : sc)
        if (s.eig_mw_fail && !mw_disabled_) {
            mw_disabled_ = true;
            mw_release();
        }
    for (const auto &s : sc)
        if (!s.stop) return false;
    return true;
}

int CmaEngine::run(int max_generations)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "run() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    const int poll = params_.poll_every > 0 ? params_.poll_every : 8;
    int done = 0;
    // a population whose budget is already spent must not take another generation
    // (the reference's loop is `while (_fev < _mfev)`, base_cmaes.cpp:166)
    {
        std::vector<CmaScal> sc;
        fetch_scal(sc);
        bool touched = false;
        for (auto &s : sc)
            if (!s.stop && s.fev >= c_.mfev) {
                s.stop = 2;
                touched = true;
            }
        if (touched) scal_.upload(sc.data(), c_.npop);
    }
    while (done < max_generations) {
        if (all_stopped()) break;
        int chunk = obj_.on_device() ? std::min(poll, max_generations - done) : 1;
        // (one launch = `chunk` generations on the fused path: keep a launch to tens of
        // milliseconds whatever poll_every says)
        if (small_fused_ok()) chunk = std::min(chunk, 512);
        if (small_fused_ok()) launch_small(chunk, true);
        else
            for (int g = 0; g < chunk; g++) generation(true);
        BBO_HIP(hipStreamSynchronize(stream_));
            timer_.collect();
        done += chunk;
    }
    return done;
}

void CmaEngine::solution(int population, double *x_out, int *n_evals, int *converged)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "solution() before initialize()");
    BBO_REQUIRE(population >= 0 && population < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    CmaScal s;
    scal_.download(&s, 1, population);
    std::vector<double> x(c_.ld);
    // bestSolution(), base_cmaes.cpp:232-238: the mean before the first generation, else
    // the best candidate of the LAST generation
    if (s.it <= 0) xmean_.download(x.data(), c_.ld, (size_t) population * c_.ld);
    else X_.download(x.data(), c_.ld, ((size_t) population * c_.lambda_pad + s.ibw[0]) * c_.ld);
    std::copy(x.begin(), x.begin() + c_.n, x_out);
    *n_evals = s.fev;
    // solution() re-runs converged() (base_cmaes.cpp:158-160); before any generation the
    // tests see it = 0 and the initial state
    if (s.it <= 0) {
        *converged = (0 >= c_.mit) ? 1 : 0;
    } else {
        *converged = s.flag != 0 ? 1 : 0;
    }
}

void CmaEngine::optimize(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged)
{
    init(n, lower, upper, guess, obj);
    // while (_fev < _mfev) { iterate(); if (converged()) break; }   base_cmaes.cpp:166-172
    const int max_gen = c_.mfev / c_.lambda + 2;
    run(max_gen);
    CmaScal s;
    scal_.download(&s, 1, 0);
    std::vector<double> x(c_.ld);
    if (s.it <= 0) xmean_.download(x.data(), c_.ld, 0);
    else X_.download(x.data(), c_.ld, (size_t) s.ibw[0] * c_.ld);
    std::copy(x.begin(), x.begin() + c_.n, x_out);
    *n_evals = s.fev;
    *converged = (s.stop == 1) ? 1 : 0;
}

double CmaEngine::evaluate_point(const double *x)
{
    if (!obj_.on_device()) {
        double f = 0.;
        obj_.eval_host(x, 1, c_.n, c_.n, &f);
        return f;
    }
    // restart drivers re-evaluate one point per restart (bipop_cmaes.cpp:86): host
    // arithmetic with the same definition and the same per-coordinate table
    return builtin_objective_host(obj_.builtin, c_.n, x, aux_h_.data());
}

// ---- named state access ---------------------------------------------------------------
int CmaEngine::get(const std::string &k, int p, double *out, int cap)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "get() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const CmaConst &c = c_;
    const size_t ld = c.ld, n = c.n;
    auto vec = [&](const DevBuf<double> &b) {   // [P][ld] -> n
        if (out && cap >= (int) n) b.download(out, n, p * ld);
        return (int) n;
    };
    auto mat = [&](const DevBuf<double> &b, size_t rows, size_t rstride, size_t cols,
            size_t base) {
        if (out && cap >= (int) (rows * cols)) {
            std::vector<double> tmp(rows * rstride);
            b.download(tmp.data(), rows * rstride, base);
            for (size_t i = 0; i < rows; i++)
                std::copy(tmp.begin() + i * rstride, tmp.begin() + i * rstride + cols,
                        out + i * cols);
        }
        return (int) (rows * cols);
    };
    auto one = [&](double v) {
        if (out && cap >= 1) out[0] = v;
        return 1;
    };
    if (c.variant == 2 && (k == "B" || k == "C" || k == "invsqrtC" || k == "ycoeff"))
        throw Error(BBO_ERR_KEY, "SepCMAES keeps a diagonal covariance: read 'csep' and 'D'");
    if (k == "xmean") return vec(xmean_);
    if (k == "xold") return vec(xold_);
    if (k == "pc") return vec(pc_);
    if (k == "ps") return vec(ps_);
    if (k == "D") return vec(D_);
    if (k == "csep") return vec(csep_);
    if (k == "B") return mat(B_, n, ld, n, p * ld * ld);
    if (k == "C") return mat(C_, n, ld, n, p * ld * ld);
    if (k == "invsqrtC") {
        if (c.lazy_isc && out) {         // not kept current by the generations: form it now
            const int hs = c_.honor_stop;
            c_.honor_stop = 0;
            launch_post(3);
            c_.honor_stop = hs;
            BBO_HIP(hipGetLastError());
            BBO_HIP(hipStreamSynchronize(stream_));
        }
        return mat(isc_, n, ld, n, p * ld * ld);
    }
    if (k == "BD") {
        // the sampler's operand B diag(D) as the kernels hold it (MFMA B-fragment order), unpacked to
        // n x n row-major: element (i, j) sits at tile i >> 4, k-step j >> 2, lane (j & 3, i & 15)
        if (out && cap >= (int) (n * n)) {
            std::vector<double> pk(ld * ld);
            BDp_.download(pk.data(), ld * ld, p * ld * ld);
            const size_t KS = ld >> 2;
            for (size_t i = 0; i < n; i++)
                for (size_t j = 0; j < n; j++)
                    out[i * n + j] = pk[((i >> 4) * KS + (j >> 2)) * 64 + ((j & 3) << 4) + (i & 15)];
        }
        return (int) (n * n);
    }
    if (k == "arx") return mat(X_, c.lambda, ld, n, (size_t) p * c.lambda_pad * ld);
    if (k == "weights") {
        if (out && cap >= c.mu) weights_.download(out, c.mu);
        return c.mu;
    }
    if (k == "fitness") {
        if (out && cap >= c.lambda) f_.download(out, c.lambda, (size_t) p * c.lambda_pad);
        return c.lambda;
    }
    if (k == "fit_idx" || k == "fit_val" || k == "rank") {
        if (out && cap >= c.lambda) {
            std::vector<int> ord(c.lambda);
            (k == "rank" ? rank_ : order_).download(ord.data(), c.lambda,
                    (size_t) p * c.lambda_pad);
            if (k == "fit_val") {
                std::vector<double> f(c.lambda);
                f_.download(f.data(), c.lambda, (size_t) p * c.lambda_pad);
                for (int i = 0; i < c.lambda; i++) out[i] = f[ord[i]];
            } else {
                for (int i = 0; i < c.lambda; i++) out[i] = ord[i];
            }
        }
        return c.lambda;
    }
    if (k == "ycoeff") {
        if (out && cap >= c.mu) {
            std::vector<double> S(c.mu_pad);
            S_.download(S.data(), c.mu_pad, (size_t) p * c.mu_pad);
            for (int i = 0; i < c.mu; i++) out[i] = S[i] / std::max(S[c.mu - 1 - i], 1e-8);
        }
        return c.mu;
    }
    if (k == "zlast") {
        const size_t cnt = (size_t) c.lambda * n;
        if (!zrecord_.p) return 0;
        if (out && cap >= (int) cnt) zrecord_.download(out, cnt, p * cnt);
        return (int) cnt;
    }
    if (k == "profile") return timer_.report(out, cap);
    if (k == "eig_work") {   // diagnostic: the eigensolver's global scratch of population p
        const size_t cnt = std::min((size_t) 4 * eig_slab(c.ld), eig_work_.count);
        if (out && (size_t) cap >= cnt) eig_work_.download(out, cnt, (size_t) p * 4 * eig_slab(c.ld));
        return (int) cnt;
    }
    if (k == "eig_stamps") {
        if (!stamps_.p) return 0;
        if (out && cap >= 48) {
            long long t[48];
            stamps_.download(t, 48);
            for (int i = 0; i < 48; i++) out[i] = (double) t[i];
        }
        return 48;
    }
    if (k == "best_hist" || k == "kth_hist") {
        if (out && cap >= c.hlen)
            (k == "best_hist" ? hist_best_ : hist_kth_).download(out, c.hlen, (size_t) p * c.hlen);
        return c.hlen;
    }
    CmaScal s;
    scal_.download(&s, 1, p);
    if (k == "sigma") return one(s.sigma);
    if (k == "it") return one(s.it);
    if (k == "fev") return one(s.fev);
    if (k == "flag") return one(s.flag);
    if (k == "stop") return one(s.stop);
    if (k == "hsig") return one(s.hsig);
    if (k == "pslen") return one(s.pslen);
    if (k == "fbest") return one(s.fbest);
    if (k == "fworst") return one(s.fworst);
    if (k == "eigenlastev") return one(s.eigenlastev);
    if (k == "eigen_done") return one(s.eigen_done);
    if (k == "basis_ok") return one(s.basis_ok);
    if (k == "eig_stage") return one(s.eig_stage);
    if (k == "eig_mw_fail") return one(s.eig_mw_fail);     // (bbo_eig_mw.hpp: sticky)
    if (k == "eig_mw_off") return one(mw_disabled_ ? 1 : 0);
    if (k == "eig_split_maxp") return one(split_maxp_);
    if (k == "eig_mw_reserved") return one((double) mw_reserved_);       // this engine's share of the device's ...
    if (k == "eig_mw_capacity") {                                        // ... budget of spread workgroups
        MwBudget &b = MwBudget::get();
        std::lock_guard<std::mutex> lock(b.m);
        return one((double) b.capacity(params_.device));
    }
    if (k == "best_len") return one(s.hist_len);
    if (k == "best_buffer") return one(s.hist_head);
    if (k == "ibest") return one(s.ibw[0]);
    if (k == "n") return one(c.n);
    if (k == "lambda") return one(c.lambda);
    if (k == "mu") return one(c.mu);
    if (k == "mueff") return one(c.mueff);
    if (k == "cc") return one(c.cc);
    if (k == "cs") return one(c.cs);
    if (k == "c1") return one(c.c1);
    if (k == "cmu") return one(c.cmu);
    if (k == "cneg") return one(c.cneg);
    if (k == "alphaold") return one(c.alphaold);
    if (k == "cm") return one(c.cm);
    if (k == "ccov") return one(c.ccov);
    if (k == "damps") return one(c.damps);
    if (k == "chi") return one(c.chi);
    if (k == "eigenfreq") return one(c.eigenfreq);
    if (k == "hlen") return one(c.hlen);
    if (k == "ik") return one(c.ik);
    if (k == "mit") return one(c.mit);
    if (k == "mfev") return one(c.mfev);
    if (k == "sigma0") return one(c.sigma0);
    throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
}

int CmaEngine::set(const std::string &k, int p, const double *in, int count)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "set() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    rank_wrote_norms_ = false;      // (S of a ranking before this call may not match the new state)
    const CmaConst &c = c_;
    const size_t ld = c.ld, n = c.n;
    auto vec = [&](DevBuf<double> &b) {
        BBO_REQUIRE(count == (int) n, "set: wrong element count");
        std::vector<double> tmp(ld, 0.);
        std::copy(in, in + n, tmp.begin());
        b.upload(tmp.data(), ld, p * ld);
        return count;
    };
    auto mat = [&](DevBuf<double> &b) {
        BBO_REQUIRE(count == (int) (n * n), "set: wrong element count");
        std::vector<double> tmp(ld * ld, 0.);
        for (size_t i = 0; i < n; i++) std::copy(in + i * n, in + (i + 1) * n, tmp.begin() + i * ld);
        b.upload(tmp.data(), ld * ld, p * ld * ld);
        return count;
    };
    if (c.variant == 2 && (k == "B" || k == "C" || k == "D"))
        throw Error(BBO_ERR_KEY, "SepCMAES: set 'csep' (D is its square root)");
    if (k == "xmean") return vec(xmean_);
    if (k == "xold") return vec(xold_);
    if (k == "pc") return vec(pc_);
    if (k == "ps") return vec(ps_);
    if (k == "csep") {
        BBO_REQUIRE(c.variant == 2, "csep belongs to SepCMAES");
        const int r = vec(csep_);
        std::vector<double> dd(ld, 1.);
        for (size_t i = 0; i < n; i++) dd[i] = std::sqrt(in[i]);
        D_.upload(dd.data(), ld, p * ld);
        return r;
    }
    if (k == "C") return mat(C_);
    if (k == "best_hist" || k == "kth_hist") {   // cmaes_history rings (crafted stop states)
        BBO_REQUIRE(count == c.hlen, "set: wrong element count");
        (k == "best_hist" ? hist_best_ : hist_kth_).upload(in, c.hlen, (size_t) p * c.hlen);
        return count;
    }
    if (k == "invsqrtC") throw Error(BBO_ERR_KEY, "invsqrtC is derived from B and D: set those");
    if (k == "B" || k == "D") {
        int r;
        if (k == "D") {
            BBO_REQUIRE(count == (int) n, "set: wrong element count");
            std::vector<double> tmp(ld, 1.);
            std::copy(in, in + n, tmp.begin());
            D_.upload(tmp.data(), ld, p * ld);
            r = count;
        } else {
            r = mat(B_);
        }
        // refresh C^-1/2 and the packed MFMA operands
        c_.honor_stop = 0;
        launch_post(1);
        BBO_HIP(hipGetLastError());
        BBO_HIP(hipStreamSynchronize(stream_));
        return r;
    }
    if (k == "profile") {
        timer_.enable(in[0] != 0., K_COUNT, K_NAMES);
        return 1;
    }
    if (k == "dbg") {
        d_.dbg = (int) in[0];
        return 1;
    }
    if (k == "sample_wide_max") {  // (tuning: at most this many 16-row tiles take the tile-per-workgroup sampler)
        sample_wide_max_tiles_ = (long) in[0];
        return 1;
    }
    if (k == "sample128_min") {    // (tuning: candidates in flight from which cma_sample_eval128 draws)
        sample128_min_rows_ = (long) in[0];
        return 1;
    }
    if (k == "eig_split_maxp") {   // (tuning: at most this many populations take the split 64 < n <= 128 decomposition)
        split_maxp_ = (int) in[0];
        return 1;
    }
    if (k == "stop_off") {     // (extension) bit k silences the stop test with flag k
        c_.stop_off = (int) in[0];
        return 1;
    }
    if (k == "ftarget") {      // (extension) f_best <= ftarget stops with flag 10
        c_.ftarget = in[0];
        return 1;
    }
    if (k == "eig_stamps") {
        if (stamps_.count != 48) stamps_.alloc(48);    // [32..47]: step clocks of diagnostic builds
        BBO_HIP(hipStreamSynchronize(stream_));
        BBO_HIP(hipMemset(stamps_.p, 0, 48 * sizeof(long long)));     // (set again = clear)
        d_.stamps = stamps_.p;
        return 1;
    }
    if (k == "record_normals") {
        BBO_REQUIRE(count == 1, "set: wrong element count");
        if (in[0] != 0.) {
            const size_t want = (size_t) c.npop * c.lambda * n;
            if (zrecord_.count != want) zrecord_.alloc(want);
            d_.zrecord = zrecord_.p;
        } else {
            d_.zrecord = nullptr;
        }
        return 1;
    }
    BBO_REQUIRE(count == 1, "set: wrong element count");
    CmaScal s;
    scal_.download(&s, 1, p);
    if (k == "sigma") s.sigma = in[0];
    else if (k == "it") s.it = (int) in[0];
    else if (k == "fev") s.fev = (int) in[0];
    else if (k == "eigenlastev") s.eigenlastev = (int) in[0];
    else if (k == "fbest") s.fbest = in[0];
    else if (k == "fworst") s.fworst = in[0];
    else if (k == "stop") s.stop = (int) in[0];
    else if (k == "best_len") s.hist_len = (int) in[0];
    else if (k == "best_buffer") s.hist_head = (int) in[0];
    else throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
    scal_.upload(&s, 1, p);
    return 1;
}

} // namespace bbo
