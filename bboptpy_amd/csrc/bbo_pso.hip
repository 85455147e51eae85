// bbo_pso.hip -- host side of the APSO engine.
// Reference behaviour restated on the host: APSOSearch::init/optimize/solution
// (apso.cpp:48-127).
#include "bbo_pso_kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <cmath>
#include <limits>

namespace bbo {

namespace {
enum { K_CENTER = 0, K_ESE, K_CTRL, K_UPDATE, K_FINISH, K_COUNT };
static const char *const K_NAMES[K_COUNT] = { "bbo:pso_center", "bbo:pso_ese", "bbo:pso_control", "bbo:pso_update", "bbo:pso_finish" };   // roctx ranges, bench.py's slot names
}

PsoEngine::PsoEngine(const bbo_params &p) :
        params_(p)
{
    BBO_REQUIRE(p.algo == BBO_ALGO_APSO, "PsoEngine: bad algo");
    BBO_REQUIRE(p.np >= 2, "APSO needs at least 2 particles");
    BBO_REQUIRE(p.populations >= 1, "populations must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(BBO_ERR_NO_DEVICE, "no HIP device visible: libbbopt_hip has no CPU path");
    BBO_REQUIRE(p.device >= 0 && p.device < ndev, "device ordinal out of range");
    BBO_HIP(hipSetDevice(p.device));
    BBO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

PsoEngine::~PsoEngine()
{
    if (stream_) (void) hipStreamDestroy(stream_);
}

void PsoEngine::init(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj)
{
    (void) guess;   // APSO never reads it (apso.cpp:48-103)
    BBO_REQUIRE(n >= 1 && n <= 2048, "APSO: dimension must be in [1, 2048]");
    for (int j = 0; j < n; j++)
        BBO_REQUIRE(std::isfinite(lower[j]) && std::isfinite(upper[j]),
                "APSO draws its swarm from [lower, upper]: the bounds must be finite");
    BBO_HIP(hipSetDevice(params_.device));
    obj_ = obj;
    const int P = params_.populations;
    PsoConst &c = c_;
    c = PsoConst {};
    c.n = n;
    c.ld = round_up(n, 2);
    c.np = params_.np;
    c.ldc = round_up(n, 16);
    c.npad = round_up(params_.np, 128);
    c.correct = params_.correct ? 1 : 0;
    c.obj = obj.on_device() ? obj.builtin : OBJ_HOST;
    c.mfev = params_.mfev;
    c.npop = P;
    c.tol = params_.tol;
    c.seed = params_.seed;
    parts_ = std::max(1, std::min(256, c.np / 64));
    {   // up to 16 refreshes of the swarm's best per generation, chunks of at least 64 particles
        // (multiples of 16: whole workgroups); a swarm of up to 64 moves in one piece
        // (beyond 32768 particles eight: a launch over fewer than ~8000 particles is one round of
        // workgroups and leaves HBM half idle -- C4's update ran at 0.37 of the roof in 16 chunks)
        const int nchunks = c.np > 32768 ? 8 : std::min(16, (c.np + 63) / 64);
        chunk_ = ((c.np + nchunks - 1) / nchunks + 15) / 16 * 16;
        if (nchunks <= 1) chunk_ = c.np;
    }

    const size_t rows = (size_t) P * c.np, ld = c.ld;
    X_.alloc(rows * ld);
    V_.alloc(rows * ld);
    XB_.alloc(rows * ld);
    f_.alloc(rows);
    fb_.alloc(rows);
    xbest_.alloc(P * ld);
    ws_.alloc(rows);
    mean_.alloc(P * ld);
    nrm_.alloc(rows);
    Xc_.alloc((size_t) P * c.npad * c.ldc);     // zeroed: the padding is never written
    pvec_.alloc(P * ld);
    radius_.alloc(rows);
    colpart_.alloc((size_t) P * parts_ * ld);
    colpart2_.alloc((size_t) P * ((c.np + 127) / 128) * c.np);
    rowpart2_.alloc(rows);
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
    std::vector<PsoScal> sc(P);
    for (auto &s : sc) {
        std::memset(&s, 0, sizeof(s));
        s.w = 0.9;
        s.c1 = s.c2 = 2.;
        s.fbest = std::numeric_limits<double>::infinity();
        s.maxit = (int) std::round(params_.mfev / (1. + c.np));   // apso.cpp:68
        s.fev = c.np;
    }
    scal_.upload(sc.data(), P);

    PsoDev &d = d_;
    d = PsoDev {};
    d.X = X_.p; d.V = V_.p; d.XB = XB_.p; d.f = f_.p; d.fb = fb_.p; d.xbest = xbest_.p;
    d.ws = ws_.p; d.mean = mean_.p; d.nrm = nrm_.p; d.Xc = Xc_.p; d.pvec = pvec_.p; d.radius = radius_.p;
    d.colpart = colpart_.p; d.colpart2 = colpart2_.p; d.rowpart2 = rowpart2_.p; d.lower = lower_.p; d.upper = upper_.p; d.aux = aux_.p;
    d.scal = scal_.p;
    c.honor_stop = 0;
    inited_ = true;

    dim3 g16((c.np + 15) / 16, P);
    const int R = rows_per_wg16(c.ld);
    hipLaunchKernelGGL(pso_init, dim3((c.np + R - 1) / R, P), dim3(16 * R),
            (size_t) R * c.ld * sizeof(double), stream_, d_, c_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) {
        host_evaluate_swarm();
        hipLaunchKernelGGL(pso_copy_fb, dim3((c.np + 255) / 256, P), dim3(256), 0, stream_, d_,
                c_);
        BBO_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(pso_gbest_init, dim3(P), dim3(256), 0, stream_, d_, c_);
    BBO_HIP(hipGetLastError());
    BBO_HIP(hipStreamSynchronize(stream_));
}

// the particles [i0, i1) of every population through the host objective (i1 < 0: the whole swarm)
void PsoEngine::host_evaluate_swarm(int i0, int i1)
{
    const PsoConst &c = c_;
    if (i1 < 0) i1 = c.np;
    const int cnt = i1 - i0;
    BBO_HIP(hipStreamSynchronize(stream_));
    std::vector<PsoScal> sc(c.npop);
    scal_.download(sc.data(), c.npop);
    std::vector<double> xh((size_t) cnt * c.ld), fh(cnt);
    for (int p = 0; p < c.npop; p++) {
        if (c.honor_stop && sc[p].stop) continue;
        X_.download(xh.data(), xh.size(), ((size_t) p * c.np + i0) * c.ld);
        obj_.eval_host(xh.data(), cnt, c.n, c.ld, fh.data());
        for (auto &v : fh)
            if (v != v) v = std::numeric_limits<double>::infinity();
        f_.upload(fh.data(), cnt, (size_t) p * c.np + i0);
    }
}

void PsoEngine::host_evaluate_elite()
{
    const PsoConst &c = c_;
    BBO_HIP(hipStreamSynchronize(stream_));
    std::vector<PsoScal> sc(c.npop);
    scal_.download(sc.data(), c.npop);
    std::vector<double> pv(c.ld);
    bool touched = false;
    for (int p = 0; p < c.npop; p++) {
        if ((c.honor_stop && sc[p].stop) || !sc[p].need_elite) continue;
        pvec_.download(pv.data(), c.ld, (size_t) p * c.ld);
        double f = 0.;
        obj_.eval_host(pv.data(), 1, c.n, c.ld, &f);
        sc[p].nu = f != f ? std::numeric_limits<double>::infinity() : f;
        touched = true;
    }
    if (touched) scal_.upload(sc.data(), c.npop);
}

void PsoEngine::generation(bool honor_stop)
{
    PsoConst &c = c_;
    c.honor_stop = honor_stop ? 1 : 0;
    const int P = c.npop;
    dim3 g16((c.np + 15) / 16, P);
    const int R = rows_per_wg16(c.ld);     // rows staged in LDS per workgroup
    const size_t ldsR = (size_t) R * c.ld * sizeof(double);
    timer_.begin(stream_, K_CENTER);
    hipLaunchKernelGGL(pso_center, dim3(parts_, P), dim3(256), 0, stream_, d_, c_, parts_);
    hipLaunchKernelGGL(pso_mean, dim3(P), dim3(256), 0, stream_, d_, c_, parts_);
    hipLaunchKernelGGL(pso_nrm, g16, dim3(256), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_ESE);
    {
        const size_t lds = (size_t) ESE2_LDS_DOUBLES * sizeof(double);
        allow_lds((const void*) pso_ese_sym, (int) lds);
        hipLaunchKernelGGL(pso_ese_sym, dim3((c.np + 127) / 128, P), dim3(256), lds, stream_, d_,
                c_);
        hipLaunchKernelGGL(pso_ese_finish, dim3((c.np + 255) / 256, P), dim3(256), 0, stream_, d_,
                c_);
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_CTRL);
    hipLaunchKernelGGL(pso_control_a, dim3(P), dim3(256), 0, stream_, d_, c_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) host_evaluate_elite();
    hipLaunchKernelGGL(pso_control_b, dim3(P), dim3(256), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    // the swarm moves in chunks of chunk_ particles, the best refreshed between them: what the
    // reference's in-loop refresh (apso.cpp:194-197) buys at large np (one chunk = the generation-
    // synchronous form of rounds 1-4: every particle sees the best of the generation start)
    const int step = chunk_ > 0 && chunk_ < c.np ? chunk_ : c.np;
    // (the timer slots: pso_update brackets each update launch, pso_finish the refreshes of the best
    // and the closing kernel)
    for (int i0 = 0; i0 < c.np; i0 += step) {
        const int i1 = std::min(c.np, i0 + step);
        timer_.begin(stream_, K_UPDATE);
        hipLaunchKernelGGL(pso_update, dim3((i1 - i0 + R - 1) / R, P), dim3(16 * R), ldsR, stream_, d_,
                c_, i0, i1);
        timer_.end(stream_);
        if (!obj_.on_device()) {
            host_evaluate_swarm(i0, i1);
            hipLaunchKernelGGL(pso_pbest, dim3((i1 - i0 + 15) / 16, P), dim3(256), 0, stream_, d_, c_, i0,
                    i1);
        }
        if (i1 < c.np) {
            timer_.begin(stream_, K_FINISH);
            hipLaunchKernelGGL(pso_gbest, dim3(P), dim3(256), 0, stream_, d_, c_, i0, i1);
            timer_.end(stream_);
        }
    }
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_FINISH);
    hipLaunchKernelGGL(pso_finish, dim3(P), dim3(256), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

void PsoEngine::iterate()
{
    if (!inited_) throw Error(BBO_ERR_STATE, "iterate() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    generation(false);
    BBO_HIP(hipStreamSynchronize(stream_));
    timer_.collect();
    std::vector<PsoScal> sc(c_.npop);
    scal_.download(sc.data(), c_.npop);
    for (const auto &s : sc)
        if (s.bad_rule)
            throw Error(BBO_ERR_ARG,
                    "Error [PSO]: Invalid rule base. Please report this issue on Github.");
}

bool PsoEngine::all_stopped()
{
    std::vector<PsoScal> sc(c_.npop);
    scal_.download(sc.data(), c_.npop);
    bool all = true;
    for (const auto &s : sc) {
        if (s.bad_rule)
            throw Error(BBO_ERR_ARG,
                    "Error [PSO]: Invalid rule base. Please report this issue on Github.");
        if (!s.stop) all = false;
    }
    return all;
}

int PsoEngine::run(int max_generations)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "run() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    {
        // `while (_it < _maxit && _fev < _mfev)`, apso.cpp:118
        std::vector<PsoScal> sc(c_.npop);
        scal_.download(sc.data(), c_.npop);
        bool touched = false;
        for (auto &s : sc)
            if (!s.stop && (s.it >= s.maxit || s.fev >= c_.mfev)) {
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

void PsoEngine::solution(int population, double *x_out, int *n_evals, int *converged)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "solution() before initialize()");
    BBO_REQUIRE(population >= 0 && population < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    PsoScal s;
    scal_.download(&s, 1, population);
    std::vector<double> x(c_.ld);
    xbest_.download(x.data(), c_.ld, (size_t) population * c_.ld);
    std::copy(x.begin(), x.begin() + c_.n, x_out);
    *n_evals = s.fev;
    if (s.it == 0) {
        std::vector<double> rad(c_.np);
        radius_.download(rad.data(), c_.np, (size_t) population * c_.np);
        double mean = 0.;
        for (double r : rad) mean += r;
        mean /= c_.np;
        double m2 = 0.;
        for (double r : rad) m2 += (r - mean) * (r - mean);
        *converged = m2 <= (c_.np - 1) * c_.tol * c_.tol ? 1 : 0;
    } else {
        *converged = s.conv;
    }
}

void PsoEngine::optimize(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged)
{
    init(n, lower, upper, guess, obj);
    run(std::numeric_limits<int>::max());
    int conv = 0;
    solution(0, x_out, n_evals, &conv);
    PsoScal s;
    scal_.download(&s, 1, 0);
    *converged = s.stop == 1 ? 1 : 0;
}

int PsoEngine::get(const std::string &k, int p, double *out, int cap)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "get() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const PsoConst &c = c_;
    PsoScal s;
    scal_.download(&s, 1, p);
    auto one = [&](double v) {
        if (out && cap >= 1) out[0] = v;
        return 1;
    };
    auto rowsof = [&](const DevBuf<double> &b) {
        const int cnt = c.np * c.n;
        if (out && cap >= cnt) {
            std::vector<double> tmp((size_t) c.np * c.ld);
            b.download(tmp.data(), tmp.size(), (size_t) p * c.np * c.ld);
            for (int i = 0; i < c.np; i++)
                std::copy(tmp.begin() + (size_t) i * c.ld, tmp.begin() + (size_t) i * c.ld + c.n,
                        out + (size_t) i * c.n);
        }
        return cnt;
    };
    auto vecof = [&](const DevBuf<double> &b) {
        if (out && cap >= c.np) b.download(out, c.np, (size_t) p * c.np);
        return c.np;
    };
    if (k == "profile") return timer_.report(out, cap);
    if (k == "x") return rowsof(X_);
    if (k == "v") return rowsof(V_);
    if (k == "xb") return rowsof(XB_);
    if (k == "f") return vecof(f_);
    if (k == "fb") return vecof(fb_);
    if (k == "ws") return vecof(ws_);
    if (k == "xbest") {
        if (out && cap >= c.n) xbest_.download(out, c.n, (size_t) p * c.ld);
        return c.n;
    }
    if (k == "fbest") return one(s.fbest);
    if (k == "w") return one(s.w);
    if (k == "c1") return one(s.c1);
    if (k == "c2") return one(s.c2);
    if (k == "state") return one(s.state);
    if (k == "it") return one(s.it);
    if (k == "maxit") return one(s.maxit);
    if (k == "fev") return one(s.fev);
    if (k == "np") return one(c.np);
    if (k == "chunk") return one(chunk_);      // particles between two refreshes of the swarm's best
    if (k == "evof") return one(s.evof);
    if (k == "stop") return one(s.stop);
    if (k == "conv") return one(s.conv);
    if (k == "m2") return one(s.m2);
    if (k == "n") return one(c.n);
    throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
}

int PsoEngine::set(const std::string &k, int p, const double *in, int count)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "set() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const PsoConst &c = c_;
    if (k == "chunk") {        // 0 or >= np: the whole swarm sees the best of the generation start
        chunk_ = (int) in[0] <= 0 ? c.np : (int) in[0];
        return 1;
    }
    if (k == "profile") {
        timer_.enable(in[0] != 0., K_COUNT, K_NAMES);
        return 1;
    }
    auto rows_in = [&](DevBuf<double> &b) {
        BBO_REQUIRE(count == c.np * c.n, "set: wrong element count");
        std::vector<double> tmp((size_t) c.np * c.ld, 0.);
        for (int i = 0; i < c.np; i++)
            std::copy(in + (size_t) i * c.n, in + (size_t) (i + 1) * c.n, tmp.begin() + (size_t) i * c.ld);
        b.upload(tmp.data(), tmp.size(), (size_t) p * c.np * c.ld);
        return count;
    };
    if (k == "x") return rows_in(X_);
    if (k == "v") return rows_in(V_);
    if (k == "xb") return rows_in(XB_);
    if (k == "f" || k == "fb") {
        BBO_REQUIRE(count == c.np, "set: wrong element count");
        (k == "f" ? f_ : fb_).upload(in, c.np, (size_t) p * c.np);
        return count;
    }
    if (k == "xbest") {
        BBO_REQUIRE(count == c.n, "set: wrong element count");
        xbest_.upload(in, c.n, (size_t) p * c.ld);
        return count;
    }
    BBO_REQUIRE(count == 1, "set: wrong element count");
    PsoScal s;
    scal_.download(&s, 1, p);
    if (k == "fbest") s.fbest = in[0];
    else if (k == "w") s.w = in[0];
    else if (k == "c1") s.c1 = in[0];
    else if (k == "c2") s.c2 = in[0];
    else if (k == "state") s.state = (int) in[0];
    else if (k == "it") s.it = (int) in[0];
    else if (k == "fev") s.fev = (int) in[0];
    else if (k == "stop") s.stop = (int) in[0];
    else throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
    scal_.upload(&s, 1, p);
    return 1;
}

Optimizer* make_pso_engine(const bbo_params &p)
{
    return new PsoEngine(p);
}

} // namespace bbo
