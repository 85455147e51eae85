// bbo_de.hip -- host side of the L-SHADE / JADE / SaNSDE engine.
// Reference behaviour restated on the host: ShadeSearch::init/optimize/solution
// (shade.cpp:56-94, :238-256), JadeSearch likewise (jade.cpp:64-96, :208-226).
#include "bbo_de_kernels.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace bbo {

namespace {
enum { K_GEN = 0, K_BOOK, K_ARCH, K_RANK, K_FINISH, K_SELECT, K_COUNT };
static const char *const K_NAMES[K_COUNT] = { "bbo:de_generation", "bbo:de_bookkeep", "bbo:de_archive_copy", "bbo:de_rank", "bbo:de_finish", "bbo:de_select" };   // roctx ranges, bench.py's slot names
}

DeEngine::DeEngine(const bbo_params &p) :
        params_(p)
{
    BBO_REQUIRE(p.algo == BBO_ALGO_SHADE || p.algo == BBO_ALGO_JADE || p.algo == BBO_ALGO_SANSDE,
            "DeEngine: bad algo");
    if (p.algo == BBO_ALGO_SANSDE)
        BBO_REQUIRE(p.crref >= 1 && p.pupdate >= 1 && p.crupdate >= 1,
                "SANSDE: crref, pupdate, crupdate must be >= 1");
    BBO_REQUIRE(p.np >= 4, "DE needs a population of at least 4");
    BBO_REQUIRE(p.populations >= 1, "populations must be >= 1");
    if (p.algo == BBO_ALGO_SHADE) {
        BBO_REQUIRE(p.h >= 1, "SHADE: h must be >= 1");
        BBO_REQUIRE(p.npmin >= 4 && p.npmin <= p.np, "SHADE: need 4 <= npmin <= npinit");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(BBO_ERR_NO_DEVICE, "no HIP device visible: libbbopt_hip has no CPU path");
    BBO_REQUIRE(p.device >= 0 && p.device < ndev, "device ordinal out of range");
    BBO_HIP(hipSetDevice(p.device));
    BBO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

DeEngine::~DeEngine()
{
    if (stream_) (void) hipStreamDestroy(stream_);
}

void DeEngine::init(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj)
{
    (void) guess;   // JADE / SHADE never read it (jade.cpp:64-96, shade.cpp:56-94)
    BBO_REQUIRE(n >= 1 && n <= 2048, "DE: dimension must be in [1, 2048]");
    for (int j = 0; j < n; j++)
        BBO_REQUIRE(std::isfinite(lower[j]) && std::isfinite(upper[j]),
                "DE draws its population from [lower, upper]: the bounds must be finite");
    BBO_HIP(hipSetDevice(params_.device));
    obj_ = obj;
    const int P = params_.populations;
    DeConst &c = c_;
    c = DeConst {};
    c.variant = params_.algo == BBO_ALGO_SANSDE ? 2 : params_.algo == BBO_ALGO_JADE ? 1 : 0;
    c.ncrref = params_.crref;
    c.npup = params_.pupdate;
    c.ncrup = params_.crupdate;
    c.n = n;
    c.ld = round_up(n, 2);
    c.npinit = params_.np;
    c.npmin = c.variant == 0 ? params_.npmin : params_.np;
    c.h = c.variant == 0 ? params_.h : 1;
    c.archive = (params_.archive && c.variant != 2) ? 1 : 0;   // SaNSDE has no archive
    c.repaircr = params_.repaircr ? 1 : 0;
    c.obj = obj.on_device() ? obj.builtin : OBJ_HOST;
    c.mfev = params_.mfev;
    c.npop = P;
    c.tol = params_.tol;
    c.pelite = params_.pelite;
    c.cdamp = params_.cdamp;
    c.jsigma = params_.jade_sigma;
    c.seed = params_.seed;
    c.np_launch = c.npinit;

    const size_t rows = (size_t) P * c.npinit, ld = c.ld;
    Xa_.alloc(rows * ld);
    Xb_.alloc(rows * ld);
    fa_.alloc(rows);
    fb_.alloc(rows);
    arch_.alloc(c.variant == 2 ? 1 : rows * ld);
    cra_.alloc(rows);
    crb_.alloc(rows);
    MCR_.alloc((size_t) P * c.h);
    MF_.alloc((size_t) P * c.h);
    rec_cr_.alloc(rows);
    rec_f_.alloc(rows);
    rec_df_.alloc(rows);
    radius_.alloc(rows);
    order_.alloc(rows);
    rank_.alloc(rows);
    rec_flag_.alloc(rows);
    claim_.alloc(rows);
    slot_of_.alloc(rows);
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
    std::vector<double> half((size_t) P * c.h, 0.5);
    MCR_.upload(half.data(), half.size());
    MF_.upload(half.data(), half.size());
    std::vector<DeScal> sc(P);
    for (auto &s : sc) {
        std::memset(&s, 0, sizeof(s));
        s.mucr = s.muf = 0.5;
        s.np = c.npinit;
        s.k = 1;
        s.fev = c.npinit;   // the initial population is evaluated (shade.cpp:88-89)
        s.sp = s.sfp = s.crm = 0.5;   // sansde.cpp:72-76
    }
    {
        std::vector<double> half_rows(rows, 0.5);   // sansde.cpp:90: every _cr starts at 0.5
        cra_.upload(half_rows.data(), rows);
        crb_.upload(half_rows.data(), rows);
    }
    scal_.upload(sc.data(), P);

    DeDev &d = d_;
    d = DeDev {};
    d.X[0] = Xa_.p; d.X[1] = Xb_.p; d.f[0] = fa_.p; d.f[1] = fb_.p;
    d.order = order_.p; d.rank = rank_.p; d.arch = arch_.p; d.MCR = MCR_.p; d.MF = MF_.p;
    d.rec_cr = rec_cr_.p; d.rec_f = rec_f_.p; d.rec_df = rec_df_.p; d.radius = radius_.p;
    d.rec_flag = rec_flag_.p; d.claim = claim_.p; d.slot_of = slot_of_.p;
    d.crow[0] = cra_.p; d.crow[1] = crb_.p;
    d.lower = lower_.p; d.upper = upper_.p; d.aux = aux_.p; d.scal = scal_.p;

    np_host_ = c.npinit;
    fev_host_ = c.npinit;
    c.honor_stop = 0;
    inited_ = true;

    const int R = rows_per_wg16(c.ld);
    dim3 grid((c.npinit + R - 1) / R, P);
    hipLaunchKernelGGL(de_init, grid, dim3(16 * R), (size_t) R * c.ld * sizeof(double), stream_,
            d_, c_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) host_evaluate(0, c.npinit);
    launch_rank(0, c.npinit);
    BBO_HIP(hipStreamSynchronize(stream_));
}

// ranks f[cur] (which_next = 0) or f[cur ^ 1] (1) of the first np individuals
void DeEngine::launch_rank(int which_next, int np_bound)
{
    const DeConst &c = c_;
    if (np_bound <= 64 && np_bound >= 2 && c.npop >= 4) {
        hipLaunchKernelGGL(de_rank_wave, dim3((c.npop + 3) / 4), dim3(256), 0, stream_, d_, c_,
                which_next);
    } else if (np_bound <= SORT_LDS_MAX && c.npop >= 4) {
        int m = 2;
        while (m < np_bound) m <<= 1;
        allow_lds((const void*) de_rank_sort, SORT_LDS_MAX * 12);
        // (2048 / 4096 keys: merge sort by merge path, two buffers; bbo_rank.hpp)
        hipLaunchKernelGGL(de_rank_sort, dim3(c.npop), dim3(sort_threads(m)),
                (m == 2048 || m == 4096) ? (size_t) m * 24 : (size_t) std::max(m, 1024) * 12,
                stream_, d_,
                c_, which_next, m);
    } else {
        dim3 rgrid((np_bound + 31) / 32, c.npop);
        hipLaunchKernelGGL(de_rank, rgrid, dim3(256), 0, stream_, d_, c_, which_next);
    }
    BBO_HIP(hipGetLastError());
}

// host objective: fitness of rows [0, rows) of buffer `which` of every population
void DeEngine::host_evaluate(int which, int rows)
{
    const DeConst &c = c_;
    BBO_HIP(hipStreamSynchronize(stream_));
    std::vector<DeScal> sc(c.npop);
    scal_.download(sc.data(), c.npop);
    std::vector<double> xh((size_t) rows * c.ld), fh(rows);
    for (int p = 0; p < c.npop; p++) {
        if (c.honor_stop && sc[p].stop) continue;
        const int buf = which < 0 ? (sc[p].cur ^ 1) : which;
        const int cnt = std::min(rows, sc[p].np);
        DevBuf<double> &X = buf == 0 ? Xa_ : Xb_;
        DevBuf<double> &F = buf == 0 ? fa_ : fb_;
        X.download(xh.data(), (size_t) cnt * c.ld, (size_t) p * c.npinit * c.ld);
        obj_.eval_host(xh.data(), cnt, c.n, c.ld, fh.data());
        for (int r = 0; r < cnt; r++)
            if (fh[r] != fh[r]) fh[r] = std::numeric_limits<double>::infinity();
        F.upload(fh.data(), cnt, (size_t) p * c.npinit);
    }
}

void DeEngine::generation(bool honor_stop)
{
    DeConst &c = c_;
    c.honor_stop = honor_stop ? 1 : 0;
    c.np_launch = np_host_;
    const int P = c.npop;
    dim3 g16((np_host_ + 15) / 16, P);
    const int R = rows_per_wg16(c.ld);     // rows staged in LDS per workgroup
    dim3 gR((np_host_ + R - 1) / R, P);
    const size_t lds = (size_t) R * c.ld * sizeof(double);
    timer_.begin(stream_, K_GEN);
    const size_t lds_box = lds + (size_t) 2 * c.ld * sizeof(double);   // + lower, upper
    if (c.variant == 2) {
        allow_lds((const void*) sansde_generation, 128 * 1024);
        hipLaunchKernelGGL(sansde_generation, gR, dim3(16 * R), lds_box, stream_, d_, c_);
    } else {
        allow_lds((const void*) de_generation, 128 * 1024);
        hipLaunchKernelGGL(de_generation, gR, dim3(16 * R), lds_box, stream_, d_, c_);
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) {
        host_evaluate(-1, np_host_);
        timer_.begin(stream_, K_SELECT);
        hipLaunchKernelGGL(de_select, g16, dim3(256), 0, stream_, d_, c_);
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
    }
    timer_.begin(stream_, K_BOOK);
    if (c.variant == 2)
        hipLaunchKernelGGL(sansde_bookkeep, dim3(P), dim3(pop_threads()), 0, stream_, d_, c_);
    else
        hipLaunchKernelGGL(de_bookkeep, dim3(P), dim3(pop_threads()), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    if (c.archive) {
        timer_.begin(stream_, K_ARCH);
        hipLaunchKernelGGL(de_archive_copy, g16, dim3(256), 0, stream_, d_, c_);
        timer_.end(stream_);
        BBO_HIP(hipGetLastError());
    }
    timer_.begin(stream_, K_RANK);
    launch_rank(1, np_host_);
    timer_.end(stream_);
    timer_.begin(stream_, K_FINISH);
    hipLaunchKernelGGL(de_finish, dim3(P), dim3(pop_threads()), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    // the same population-size schedule on the host (shade.cpp:218-225), for the grid only
    fev_host_ += np_host_;
    if (c.variant == 0) {
        const int npnew = (int) std::round((c.npmin - c.npinit) * ((1. * fev_host_) / c.mfev)
                + c.npinit);
        if (npnew < np_host_) np_host_ = std::max(npnew, c.npmin);
    }
}

void DeEngine::iterate()
{
    if (!inited_) throw Error(BBO_ERR_STATE, "iterate() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    generation(false);
    BBO_HIP(hipStreamSynchronize(stream_));
    timer_.collect();
}

bool DeEngine::all_stopped()
{
    std::vector<DeScal> sc(c_.npop);
    scal_.download(sc.data(), c_.npop);
    for (const auto &s : sc)
        if (!s.stop) return false;
    return true;
}

int DeEngine::run(int max_generations)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "run() before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    {
        // `while (_fev < _mfev)` (shade.cpp:247): no generation once the budget is spent
        std::vector<DeScal> sc(c_.npop);
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

void DeEngine::solution(int population, double *x_out, int *n_evals, int *converged)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "solution() before initialize()");
    BBO_REQUIRE(population >= 0 && population < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    DeScal s;
    scal_.download(&s, 1, population);
    int best = 0;
    order_.download(&best, 1, (size_t) population * c_.npinit);
    std::vector<double> x(c_.ld);
    (s.cur == 0 ? Xa_ : Xb_).download(x.data(), c_.ld,
            ((size_t) population * c_.npinit + best) * c_.ld);
    std::copy(x.begin(), x.begin() + c_.n, x_out);
    *n_evals = s.fev;
    if (s.gen == 0) {
        // converged() before any generation: evaluate the radius spread of the initial swarm
        std::vector<double> rad(s.np);
        radius_.download(rad.data(), s.np, (size_t) population * c_.npinit);
        double mean = 0.;
        for (double r : rad) mean += r;
        mean /= s.np;
        double m2 = 0.;
        for (double r : rad) m2 += (r - mean) * (r - mean);
        *converged = m2 <= (s.np - 1) * c_.tol * c_.tol ? 1 : 0;
    } else {
        *converged = s.conv;
    }
}

void DeEngine::optimize(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged)
{
    init(n, lower, upper, guess, obj);
    // while (_fev < _mfev) { iterate(); if (converged()) break; }   shade.cpp:247-253
    run(std::numeric_limits<int>::max());
    int conv = 0;
    solution(0, x_out, n_evals, &conv);
    DeScal s;
    scal_.download(&s, 1, 0);
    *converged = s.stop == 1 ? 1 : 0;
}

int DeEngine::get(const std::string &k, int p, double *out, int cap)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "get() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const DeConst &c = c_;
    DeScal s;
    scal_.download(&s, 1, p);
    const size_t pbase = (size_t) p * c.npinit;
    auto one = [&](double v) {
        if (out && cap >= 1) out[0] = v;
        return 1;
    };
    if (k == "profile") return timer_.report(out, cap);
    if (k == "x" || k == "f") {   // in sorted order, like the reference's _swarm
        const int cnt = k == "x" ? s.np * c.n : s.np;
        if (out && cap >= cnt) {
            std::vector<int> ord(s.np);
            order_.download(ord.data(), s.np, pbase);
            if (k == "f") {
                std::vector<double> f(c.npinit);
                (s.cur == 0 ? fa_ : fb_).download(f.data(), c.npinit, pbase);
                for (int i = 0; i < s.np; i++) out[i] = f[ord[i]];
            } else {
                std::vector<double> X((size_t) c.npinit * c.ld);
                (s.cur == 0 ? Xa_ : Xb_).download(X.data(), X.size(), pbase * c.ld);
                for (int i = 0; i < s.np; i++)
                    std::copy(X.begin() + (size_t) ord[i] * c.ld,
                            X.begin() + (size_t) ord[i] * c.ld + c.n, out + (size_t) i * c.n);
            }
        }
        return cnt;
    }
    if (k == "arch") {
        const int cnt = s.larch * c.n;
        if (out && cap >= cnt && cnt > 0) {
            std::vector<double> A((size_t) s.larch * c.ld);
            arch_.download(A.data(), A.size(), pbase * c.ld);
            for (int i = 0; i < s.larch; i++)
                std::copy(A.begin() + (size_t) i * c.ld, A.begin() + (size_t) i * c.ld + c.n,
                        out + (size_t) i * c.n);
        }
        return cnt;
    }
    if (k == "MCR" || k == "MF") {
        if (out && cap >= c.h) (k == "MCR" ? MCR_ : MF_).download(out, c.h, (size_t) p * c.h);
        return c.h;
    }
    if (k == "rec_flag" || k == "rec_cr" || k == "rec_f" || k == "rec_df") {
        if (out && cap >= s.np) {
            if (k == "rec_flag") {
                std::vector<int> fl(s.np);
                rec_flag_.download(fl.data(), s.np, pbase);
                for (int i = 0; i < s.np; i++) out[i] = fl[i];
            } else {
                (k == "rec_cr" ? rec_cr_ : k == "rec_f" ? rec_f_ : rec_df_).download(out, s.np,
                        pbase);
            }
        }
        return s.np;
    }
    if (k == "cr") {   // SaNSDE: per-individual CR in sorted order
        if (out && cap >= s.np) {
            std::vector<int> ord(s.np);
            order_.download(ord.data(), s.np, pbase);
            std::vector<double> crv(c.npinit);
            (s.cur == 0 ? cra_ : crb_).download(crv.data(), c.npinit, pbase);
            for (int i = 0; i < s.np; i++) out[i] = crv[ord[i]];
        }
        return s.np;
    }
    if (k == "pns" || k == "pnf" || k == "fpns" || k == "fpnf") {
        if (out && cap >= 2)
            for (int q = 0; q < 2; q++)
                out[q] = k == "pns" ? s.pns[q] : k == "pnf" ? s.pnf[q] : k == "fpns" ? s.fpns[q]
                                                                                     : s.fpnf[q];
        return 2;
    }
    if (k == "p") return one(s.sp);
    if (k == "fp") return one(s.sfp);
    if (k == "crm") return one(s.crm);
    if (k == "crrec") return one(s.crrec);
    if (k == "crdeltaf") return one(s.crdeltaf);
    if (k == "it") return one(s.gen);
    if (k == "k") return one(s.k);
    if (k == "np") return one(s.np);
    if (k == "fev") return one(s.fev);
    if (k == "gen") return one(s.gen);
    if (k == "larch") return one(s.larch);
    if (k == "mucr") return one(s.mucr);
    if (k == "muf") return one(s.muf);
    if (k == "stop") return one(s.stop);
    if (k == "conv") return one(s.conv);
    if (k == "m2") return one(s.m2);
    if (k == "nsucc") return one(s.nsucc);
    if (k == "n") return one(c.n);
    throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
}

int DeEngine::set(const std::string &k, int p, const double *in, int count)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "set() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const DeConst &c = c_;
    DeScal s;
    scal_.download(&s, 1, p);
    const size_t pbase = (size_t) p * c.npinit;
    if (k == "profile") {
        timer_.enable(in[0] != 0., K_COUNT, K_NAMES);
        return 1;
    }
    if (k == "x") {   // rows in sorted order; follow with set("f") to re-rank
        BBO_REQUIRE(count % c.n == 0 && count / c.n <= c.npinit, "set x: bad element count");
        const int rows = count / c.n;
        std::vector<double> X((size_t) rows * c.ld, 0.);
        for (int i = 0; i < rows; i++)
            std::copy(in + (size_t) i * c.n, in + (size_t) (i + 1) * c.n, X.begin() + (size_t) i * c.ld);
        (s.cur == 0 ? Xa_ : Xb_).upload(X.data(), X.size(), pbase * c.ld);
        s.np = rows;
        scal_.upload(&s, 1, p);
        np_host_ = std::max(np_host_, rows);
        return count;
    }
    if (k == "f") {
        BBO_REQUIRE(count == s.np, "set f: count must equal np");
        (s.cur == 0 ? fa_ : fb_).upload(in, count, pbase);
        c_.honor_stop = 0;
        c_.np_launch = c.npinit;
        launch_rank(0, c.npinit);
        BBO_HIP(hipStreamSynchronize(stream_));
        return count;
    }
    if (k == "arch") {
        BBO_REQUIRE(count % c.n == 0 && count / c.n <= c.npinit, "set arch: bad element count");
        const int rows = count / c.n;
        if (rows > 0) {
            std::vector<double> A((size_t) rows * c.ld, 0.);
            for (int i = 0; i < rows; i++)
                std::copy(in + (size_t) i * c.n, in + (size_t) (i + 1) * c.n,
                        A.begin() + (size_t) i * c.ld);
            arch_.upload(A.data(), A.size(), pbase * c.ld);
        }
        s.larch = rows;
        scal_.upload(&s, 1, p);
        return count;
    }
    if (k == "MCR" || k == "MF") {
        BBO_REQUIRE(count == c.h, "set MCR/MF: count must equal h");
        (k == "MCR" ? MCR_ : MF_).upload(in, c.h, (size_t) p * c.h);
        return count;
    }
    BBO_REQUIRE(count == 1, "set: wrong element count");
    if (k == "k") s.k = (int) in[0];
    else if (k == "fev") { s.fev = (int) in[0]; fev_host_ = s.fev; }
    else if (k == "gen") s.gen = (int) in[0];
    else if (k == "mucr") s.mucr = in[0];
    else if (k == "muf") s.muf = in[0];
    else if (k == "stop") s.stop = (int) in[0];
    else throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
    scal_.upload(&s, 1, p);
    return 1;
}

Optimizer* make_de_engine(const bbo_params &p)
{
    return new DeEngine(p);
}

} // namespace bbo
