// bbo_cso.hip -- host side of the CSO engine.  Reference behaviour restated on the host:
// CSOSearch::CSOSearch / init / optimize / solution (cso.cpp:46-112, :160-175).
#include "bbo_cso_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>

namespace bbo {

namespace {
enum { K_MEAN = 0, K_SHUFFLE, K_GROUPS, K_COMPETE, K_FINISH, K_COUNT };
static const char *const K_NAMES[K_COUNT] = { "bbo:cso_mean", "bbo:cso_shuffle", "bbo:cso_groups", "bbo:cso_compete", "bbo:cso_finish" };   // roctx ranges, bench.py's slot names
}

CsoEngine::CsoEngine(const bbo_params &p) :
        params_(p)
{
    BBO_REQUIRE(p.algo == BBO_ALGO_CSO, "CsoEngine: bad algo");
    BBO_REQUIRE(p.np >= 2, "CSO needs at least 2 particles");
    BBO_REQUIRE(p.populations >= 1, "populations must be >= 1");
    // cso.cpp:53-64: at least two particles per competition, np rounded up to a multiple
    if (params_.pcompete < 2) {
        params_.pcompete = 2;
        fprintf(stderr, "Warning [CSO]: particles per competition is too small - adjusted.\n");
    }
    while (params_.np % params_.pcompete != 0) params_.np++;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(BBO_ERR_NO_DEVICE, "no HIP device visible: libbbopt_hip has no CPU path");
    BBO_REQUIRE(p.device >= 0 && p.device < ndev, "device ordinal out of range");
    BBO_HIP(hipSetDevice(p.device));
    BBO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

CsoEngine::~CsoEngine()
{
    if (stream_) (void) hipStreamDestroy(stream_);
}

void CsoEngine::init(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj)
{
    (void) guess;   // CSO never reads it (cso.cpp:67-112)
    BBO_REQUIRE(n >= 1 && n <= 1024, "CSO: dimension must be in [1, 1024]");
    for (int j = 0; j < n; j++)
        BBO_REQUIRE(std::isfinite(lower[j]) && std::isfinite(upper[j]),
                "CSO draws its swarm from [lower, upper]: the bounds must be finite");
    BBO_HIP(hipSetDevice(params_.device));
    obj_ = obj;
    const int P = params_.populations;
    CsoConst &c = c_;
    c = CsoConst {};
    c.n = n;
    c.ld = round_up(n, 2);
    c.np = params_.np;
    c.pc = params_.pcompete;
    c.ngroup = c.np / c.pc;
    c.ring = params_.ring ? 1 : 0;
    c.correct = params_.correct ? 1 : 0;
    c.obj = obj.on_device() ? obj.builtin : OBJ_HOST;
    c.mfev = params_.mfev;
    c.npop = P;
    c.stol = params_.tol;
    c.vmax = params_.vmax;
    c.seed = params_.seed;
    c.parts = std::max(1, std::min(256, c.np / 64));
    c.fparts = std::max(1, std::min(64, c.np / 1024));
    // cso.cpp:196-217
    if (c.pc == 2) {
        if (c.np <= 100) {
            c.phil = c.phih = 0.;
        } else {
            c.phil = std::max(0., 0.14 * std::log(c.np) - 0.3);
            c.phih = std::max(0., 0.27 * std::log(c.np) - 0.51);
        }
    } else {
        c.phil = 0.;
        c.phih = 0.3;
    }

    const size_t rows = (size_t) P * c.np, ld = c.ld;
    X_.alloc(rows * ld);
    V_.alloc(rows * ld);
    PM_.alloc(c.ring ? rows * ld : 1);
    f_.alloc(rows);
    radius_.alloc(rows);
    occ_.alloc(rows);
    occ2_.alloc(rows);
    mean_.alloc(P * ld);
    meanw_.alloc(P * ld);
    colpart_.alloc((size_t) P * c.parts * ld);
    // the swarm mean from cso_compete's own sums where a lane's share of a row is at most four
    // column pairs (ld <= 512 with up to 64 lanes per group) and the global mean is what is used
    fuse_g_ = (c.ring || c.ld > 512) ? 0 : c.ld <= 128 ? 16 : c.ld <= 256 ? 32 : 64;
    c.nwg = fuse_g_ ? (c.ngroup + 256 / fuse_g_ - 1) / (256 / fuse_g_) : 1;
    wgpart_.alloc(fuse_g_ ? (size_t) P * c.nwg * ld : 1);
    fpart_.alloc((size_t) P * c.fparts * CSO_FPART);
    lower_.alloc(ld);
    upper_.alloc(ld);
    aux_.alloc(ld);
    scal_.alloc(P);
    std::vector<double> lo(ld, 0.), up(ld, 0.);
    aux_h_.assign(ld, 0.);
    std::copy(lower, lower + n, lo.begin());
    std::copy(upper, upper + n, up.begin());
    fill_objective_aux(obj.on_device() ? obj.builtin : -1, n, aux_h_.data());
    lower_.upload(lo.data(), ld);
    upper_.upload(up.data(), ld);
    aux_.upload(aux_h_.data(), ld);
    std::vector<CsoScal> sc(P);
    for (auto &s : sc) {
        std::memset(&s, 0, sizeof(s));
        s.fev = c.np;          // the initial swarm is evaluated (cso.cpp:99)
        s.fbest = std::numeric_limits<double>::infinity();
    }
    scal_.upload(sc.data(), P);

    CsoDev &d = d_;
    d = CsoDev {};
    d.X = X_.p; d.V = V_.p; d.PM = PM_.p; d.f = f_.p; d.radius = radius_.p;
    d.occ = occ_.p; d.occ2 = occ2_.p;
    d.mean = mean_.p; d.meanw = meanw_.p; d.colpart = colpart_.p; d.fpart = fpart_.p;
    d.wgpart = wgpart_.p;
    d.lower = lower_.p; d.upper = upper_.p; d.aux = aux_.p; d.scal = scal_.p;
    c.honor_stop = 0;
    inited_ = true;

    const int R = rows_per_wg16(c.ld);
    hipLaunchKernelGGL(cso_init, dim3((c.np + R - 1) / R, P), dim3(16 * R),
            (size_t) R * c.ld * sizeof(double), stream_, d_, c_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) host_evaluate(false);
    hipLaunchKernelGGL(cso_finish_part, dim3(c_.fparts, P), dim3(256), 0, stream_, d_, c_);
    hipLaunchKernelGGL(cso_finish, dim3(P), dim3(64), 0, stream_, d_, c_, 1);
    if (fuse_g_) {      // the sums cso_compete maintains from now on, of the initial swarm
        const size_t lds = (size_t) (256 / fuse_g_ + 2) * c.ld * sizeof(double);   // rows + the box
        const dim3 grid(c.nwg, P);
        if (fuse_g_ == 16) hipLaunchKernelGGL(cso_team_colsum<16>, grid, dim3(256), lds, stream_, d_, c_);
        else if (fuse_g_ == 32) hipLaunchKernelGGL(cso_team_colsum<32>, grid, dim3(256), lds, stream_, d_, c_);
        else hipLaunchKernelGGL(cso_team_colsum<64>, grid, dim3(256), lds, stream_, d_, c_);
    }
    BBO_HIP(hipGetLastError());
    BBO_HIP(hipStreamSynchronize(stream_));
}

// host objective: every particle (init) or the losers of this generation (slots that are not
// the first of their group), so that the callable is called exactly `fev` times
void CsoEngine::host_evaluate(bool losers_only)
{
    const CsoConst &c = c_;
    BBO_HIP(hipStreamSynchronize(stream_));
    std::vector<CsoScal> sc(c.npop);
    scal_.download(sc.data(), c.npop);
    std::vector<double> xh((size_t) c.np * c.ld), fh(c.np);
    std::vector<int> occ(c.np);
    for (int p = 0; p < c.npop; p++) {
        if (c.honor_stop && sc[p].stop) continue;
        X_.download(xh.data(), xh.size(), (size_t) p * c.np * c.ld);
        f_.download(fh.data(), c.np, (size_t) p * c.np);
        occ_.download(occ.data(), c.np, (size_t) p * c.np);
        for (int s = 0; s < c.np; s++) {
            if (losers_only && s % c.pc == 0) continue;   // the winner of a group does not move
            const int row = occ[s];
            double f = 0.;
            obj_.eval_host(xh.data() + (size_t) row * c.ld, 1, c.n, c.ld, &f);
            fh[row] = f != f ? std::numeric_limits<double>::infinity() : f;
        }
        f_.upload(fh.data(), c.np, (size_t) p * c.np);
    }
}

void CsoEngine::generation(bool honor_stop)
{
    CsoConst &c = c_;
    c.honor_stop = honor_stop ? 1 : 0;
    const int P = c.npop;
    timer_.begin(stream_, K_MEAN);
    if (c.ring) {
        hipLaunchKernelGGL(cso_ring_mean, dim3((c.np + 15) / 16, P), dim3(256), 0, stream_, d_, c_);
    } else if (fuse_g_) {
        hipLaunchKernelGGL(cso_wgsum, dim3(c.parts, P), dim3(256), 0, stream_, d_, c_);
        hipLaunchKernelGGL(cso_mean, dim3(P), dim3(256), 0, stream_, d_, c_, 0, c.np);
    } else {
        hipLaunchKernelGGL(cso_colsum, dim3(c.parts, P), dim3(256), 0, stream_, d_, c_, 1, c.np);
        hipLaunchKernelGGL(cso_mean, dim3(P), dim3(256), 0, stream_, d_, c_, 0, c.np);
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_SHUFFLE);
    {
        int bits = 1;
        while ((1u << bits) < (unsigned) c.np) bits++;
        hipLaunchKernelGGL(cso_shuffle, dim3((c.np + 255) / 256, P), dim3(256), 0, stream_, d_, c_,
                (bits + 1) / 2);
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_GROUPS);
    hipLaunchKernelGGL(cso_groups, dim3((c.ngroup + 255) / 256, P), dim3(256), 0, stream_, d_, c_);
    hipLaunchKernelGGL(cso_colsum, dim3(c.parts, P), dim3(256), 0, stream_, d_, c_, c.pc,
            c.ngroup);
    hipLaunchKernelGGL(cso_mean, dim3(P), dim3(256), 0, stream_, d_, c_, 1, c.ngroup);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_COMPETE);
    if (fuse_g_) {
        const size_t lds = (size_t) (256 / fuse_g_ + 2) * c.ld * sizeof(double);   // rows + the box
        const dim3 grid(c.nwg, P);
        if (fuse_g_ == 16)
            hipLaunchKernelGGL((cso_compete<16, true>), grid, dim3(256), lds, stream_, d_, c_);
        else if (fuse_g_ == 32)
            hipLaunchKernelGGL((cso_compete<32, true>), grid, dim3(256), lds, stream_, d_, c_);
        else
            hipLaunchKernelGGL((cso_compete<64, true>), grid, dim3(256), lds, stream_, d_, c_);
    } else {
        const int R = rows_per_wg16(c.ld);     // groups staged in LDS per workgroup
        allow_lds((const void*) cso_compete<16, false>, 128 * 1024);
        hipLaunchKernelGGL((cso_compete<16, false>), dim3((c.ngroup + R - 1) / R, P), dim3(16 * R),
                (size_t) (R + 2) * c.ld * sizeof(double), stream_, d_, c_);
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) host_evaluate(true);
    timer_.begin(stream_, K_FINISH);
    hipLaunchKernelGGL(cso_finish_part, dim3(c_.fparts, P), dim3(256), 0, stream_, d_, c_);
    hipLaunchKernelGGL(cso_finish, dim3(P), dim3(64), 0, stream_, d_, c_, 0);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

void CsoEngine::iterate()
{
    if (!inited_) throw Error(BBO_ERR_STATE, "iterate() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    generation(false);
    BBO_HIP(hipStreamSynchronize(stream_));
    timer_.collect();
}

bool CsoEngine::all_stopped()
{
    std::vector<CsoScal> sc(c_.npop);
    scal_.download(sc.data(), c_.npop);
    for (const auto &s : sc)
        if (!s.stop) return false;
    return true;
}

int CsoEngine::run(int max_generations)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "run() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    {
        // `while (_fev < _mfev)`, cso.cpp:167
        std::vector<CsoScal> sc(c_.npop);
        scal_.download(sc.data(), c_.npop);
        bool touched = false;
        for (auto &s : sc)
            if (!s.stop && s.fev >= c_.mfev) {
                s.stop = 2;
                touched = true;
            }
        if (touched) scal_.upload(sc.data(), c_.npop);
    }
    const int poll = params_.poll_every > 0 ? params_.poll_every : 8;
    int done = 0;
    while (done < max_generations) {
        if (all_stopped()) break;
        const int chunk = obj_.on_device() ? std::min(poll, max_generations - done) : 1;
        for (int g = 0; g < chunk; g++) generation(true);
        BBO_HIP(hipStreamSynchronize(stream_));
        timer_.collect();
        done += chunk;
    }
    return done;
}

void CsoEngine::solution(int population, double *x_out, int *n_evals, int *converged)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "solution() before initialize()");
    BBO_REQUIRE(population >= 0 && population < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    CsoScal s;
    scal_.download(&s, 1, population);
    std::vector<double> x(c_.ld);
    X_.download(x.data(), c_.ld, ((size_t) population * c_.np + s.ibest) * c_.ld);
    std::copy(x.begin(), x.begin() + c_.n, x_out);
    *n_evals = s.fev;
    *converged = s.conv;
}

void CsoEngine::optimize(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged)
{
    init(n, lower, upper, guess, obj);
    run(std::numeric_limits<int>::max());
    int conv = 0;
    solution(0, x_out, n_evals, &conv);
    CsoScal s;
    scal_.download(&s, 1, 0);
    *converged = s.stop == 1 ? 1 : 0;
}

int CsoEngine::get(const std::string &k, int p, double *out, int cap)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "get() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const CsoConst &c = c_;
    CsoScal s;
    scal_.download(&s, 1, p);
    const size_t pb = (size_t) p * c.np;
    auto one = [&](double v) {
        if (out && cap >= 1) out[0] = v;
        return 1;
    };
    if (k == "profile") return timer_.report(out, cap);
    // per-particle arrays are reported in SLOT order, like the reference's _swarm
    if (k == "x" || k == "v" || k == "pmean") {
        const int cnt = c.np * c.n;
        if (k == "pmean" && !c.ring) return 0;
        if (out && cap >= cnt) {
            std::vector<int> occ(c.np);
            occ_.download(occ.data(), c.np, pb);
            std::vector<double> M((size_t) c.np * c.ld);
            (k == "x" ? X_ : k == "v" ? V_ : PM_).download(M.data(), M.size(), pb * c.ld);
            for (int sl = 0; sl < c.np; sl++)
                std::copy(M.begin() + (size_t) occ[sl] * c.ld,
                        M.begin() + (size_t) occ[sl] * c.ld + c.n, out + (size_t) sl * c.n);
        }
        return cnt;
    }
    if (k == "f" || k == "home") {
        if (out && cap >= c.np) {
            std::vector<int> occ(c.np);
            occ_.download(occ.data(), c.np, pb);
            std::vector<double> f(c.np);
            f_.download(f.data(), c.np, pb);
            for (int sl = 0; sl < c.np; sl++) out[sl] = k == "f" ? f[occ[sl]] : (double) occ[sl];
        }
        return c.np;
    }
    if (k == "mean" || k == "meanw" || k == "xbest") {
        if (out && cap >= c.n) {
            std::vector<double> v(c.ld);
            if (k == "xbest") X_.download(v.data(), c.ld, (pb + s.ibest) * c.ld);
            else (k == "mean" ? mean_ : meanw_).download(v.data(), c.ld, (size_t) p * c.ld);
            std::copy(v.begin(), v.begin() + c.n, out);
        }
        return c.n;
    }
    if (k == "fbest") return one(s.fbest);
    if (k == "np") return one(c.np);
    if (k == "fev") return one(s.fev);
    if (k == "it") return one(s.gen);
    if (k == "stop") return one(s.stop);
    if (k == "conv") return one(s.conv);
    if (k == "m2") return one(s.m2);
    if (k == "phil") return one(c.phil);
    if (k == "phih") return one(c.phih);
    if (k == "n") return one(c.n);
    throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
}

int CsoEngine::set(const std::string &k, int p, const double *in, int count)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "set() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    (void) count;
    if (k == "profile") {
        timer_.enable(in[0] != 0., K_COUNT, K_NAMES);
        return 1;
    }
    throw Error(BBO_ERR_KEY, "unknown or read-only state key '" + k + "'");
}

Optimizer* make_cso_engine(const bbo_params &p)
{
    return new CsoEngine(p);
}

} // namespace bbo
