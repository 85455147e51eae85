// bbo_ccpso.hip -- host side of the CCPSO2 engine (ccpso.cpp:51-148; no local optimizer).
#include "bbo_ccpso_kernels.hpp"
#include "bbo_cma.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace bbo {

namespace {
enum { K_REGROUP = 0, K_EVAL, K_UPDATE, K_POSITION, K_FINISH, K_COUNT };
static const char *const K_NAMES[K_COUNT] = { "bbo:ccp_regroup", "bbo:ccp_eval", "bbo:ccp_update", "bbo:ccp_position", "bbo:ccp_finish" };   // roctx ranges, bench.py's slot names
}

CcpsoEngine::CcpsoEngine(const bbo_params &p) :
        params_(p)
{
    BBO_REQUIRE(p.algo == BBO_ALGO_CCPSO, "CcpsoEngine: bad algo");
    BBO_REQUIRE(p.np >= 3, "CCPSO needs at least 3 particles (ring neighbourhood)");
    BBO_REQUIRE(p.npps >= 1 && p.npps <= 16, "CCPSO: between 1 and 16 swarm sizes (pps)");
    BBO_REQUIRE(p.populations >= 1, "populations must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        throw Error(BBO_ERR_NO_DEVICE, "no HIP device visible: libbbopt_hip has no CPU path");
    BBO_REQUIRE(p.device >= 0 && p.device < ndev, "device ordinal out of range");
    BBO_HIP(hipSetDevice(p.device));
    BBO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
}

CcpsoEngine::~CcpsoEngine()
{
    if (stream_) (void) hipStreamDestroy(stream_);
}

void CcpsoEngine::init(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj)
{
    (void) guess;   // never read (ccpso.cpp:66-112)
    BBO_REQUIRE(n >= 1 && n <= 1024, "CCPSO: dimension must be in [1, 1024]");
    for (int j = 0; j < n; j++)
        BBO_REQUIRE(std::isfinite(lower[j]) && std::isfinite(upper[j]),
                "CCPSO draws its swarm from [lower, upper]: the bounds must be finite");
    for (int k = 0; k < params_.npps; k++)   // the reference throws when the size is drawn (:196)
        if (params_.pps[k] <= 0 || params_.pps[k] > n || n % params_.pps[k] != 0)
            throw Error(BBO_ERR_ARG, "Error [CC-PSO]: invalid component size.");
    BBO_HIP(hipSetDevice(params_.device));
    obj_ = obj;
    const int P = params_.populations;
    CcpConst &c = c_;
    c = CcpConst {};
    c.shard_rank = shard_rank_;
    c.shard_world = shard_world_;
    c.n = n;
    c.ld = round_up(n, 2);
    c.np = params_.np;
    c.npps = params_.npps;
    for (int k = 0; k < c.npps; k++) c.pps[k] = params_.pps[k];
    c.correct = params_.correct ? 1 : 0;
    c.phat0 = params_.pcauchy;
    // (the reference reads an uninitialised member in this test, ccpso.cpp:60; the intended
    // reading is taken: adapt unless a probability in (0, 1) was given)
    c.adaptp = !(c.phat0 > 0. && c.phat0 < 1.) ? 1 : 0;
    c.obj = obj.on_device() ? obj.builtin : OBJ_HOST;
    c.mfev = params_.mfev;
    c.npop = P;
    c.stol = params_.tol;
    c.seed = params_.seed;

    const size_t rows = (size_t) P * c.np, ld = c.ld, sw = (size_t) P * n * c.np;
    X_.alloc(rows * ld);
    Y_.alloc(rows * ld);
    yhat_.alloc(P * ld);
    ysave_.alloc(P * ld);
    fX_.alloc(sw);
    fY_.alloc(sw);
    ibest_.alloc(sw);
    strat_.alloc(sw);
    range_.alloc((size_t) P * n);
    grp_of_.alloc((size_t) P * n);
    radius_.alloc(rows);
    rpart_.alloc(rows * ((ld / 2 + 255) / 256));
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
    lower_h_.assign(lower, lower + n);
    upper_h_.assign(upper, upper + n);
    nlocal_ = 0;
    std::vector<int> zi(sw, 0);
    strat_.upload(zi.data(), sw);
    ibest_.upload(zi.data(), sw);
    std::vector<CcpScal> sc(P);
    for (auto &s : sc) {
        std::memset(&s, 0, sizeof(s));
        s.fev = c.np;
        s.is = -1;
        s.phat = c.adaptp ? 0.5 : c.phat0;
        s.fyhat = std::numeric_limits<double>::infinity();
    }
    scal_.upload(sc.data(), P);

    CcpDev &d = d_;
    d = CcpDev {};
    d.X = X_.p; d.Y = Y_.p; d.yhat = yhat_.p; d.ysave = ysave_.p; d.fX = fX_.p; d.fY = fY_.p;
    d.ibest = ibest_.p; d.strat = strat_.p; d.range = range_.p; d.grp_of = grp_of_.p;
    d.radius = radius_.p; d.rpart = rpart_.p; d.lower = lower_.p; d.upper = upper_.p; d.aux = aux_.p; d.scal = scal_.p;
    c.honor_stop = 0;
    inited_ = true;

    allow_lds((const void*) ccp_init, 128 * 1024);
    allow_lds((const void*) ccp_eval<16>, 128 * 1024);
    hipLaunchKernelGGL(ccp_init, dim3((c.np + 15) / 16, P), dim3(256),
            (size_t) 16 * c.ld * sizeof(double), stream_, d_, c_);
    BBO_HIP(hipGetLastError());
    if (!obj_.on_device()) {
        BBO_HIP(hipStreamSynchronize(stream_));
        std::vector<double> xh((size_t) c.np * c.ld), fh(c.np);
        for (int p = 0; p < P; p++) {
            X_.download(xh.data(), xh.size(), (size_t) p * c.np * c.ld);
            obj_.eval_host(xh.data(), c.np, c.n, c.ld, fh.data());
            for (auto &v : fh)
                if (v != v) v = std::numeric_limits<double>::infinity();
            fX_.upload(fh.data(), c.np, (size_t) p * n * c.np);
        }
    }
    hipLaunchKernelGGL(ccp_init_yhat, dim3(P), dim3(256), 0, stream_, d_, c_);
    BBO_HIP(hipGetLastError());
    BBO_HIP(hipStreamSynchronize(stream_));
}

// host objective: the 2 nswarm np context-vector evaluations, in the reference's order
void CcpsoEngine::host_eval_candidates()
{
    const CcpConst &c = c_;
    BBO_HIP(hipStreamSynchronize(stream_));
    std::vector<CcpScal> sc(c.npop);
    scal_.download(sc.data(), c.npop);
    std::vector<double> X((size_t) c.np * c.ld), Y(X.size()), yh(c.ld), work(c.ld);
    std::vector<int> rg(c.n);
    for (int p = 0; p < c.npop; p++) {
        if (c.honor_stop && sc[p].stop) continue;
        const int nswarm = sc[p].nswarm, cp = sc[p].cpswarm;
        X_.download(X.data(), X.size(), (size_t) p * c.np * c.ld);
        Y_.download(Y.data(), Y.size(), (size_t) p * c.np * c.ld);
        yhat_.download(yh.data(), c.ld, (size_t) p * c.ld);
        range_.download(rg.data(), c.n, (size_t) p * c.n);
        std::vector<double> fx((size_t) nswarm * c.np), fy(fx.size());
        const int j0 = (int) ((long) nswarm * c.shard_rank / c.shard_world);
        const int j1 = (int) ((long) nswarm * (c.shard_rank + 1) / c.shard_world);
        for (int j = j0; j < j1; j++)      // (all swarms unless the groups are sharded over ranks)
            for (int i = 0; i < c.np; i++)
                for (int which = 0; which < 2; which++) {
                    work = yh;
                    const double *src = (which ? Y : X).data() + (size_t) i * c.ld;
                    for (int q = 0; q < cp; q++) work[rg[j * cp + q]] = src[rg[j * cp + q]];
                    double f = 0.;
                    obj_.eval_host(work.data(), 1, c.n, c.ld, &f);
                    if (f != f) f = std::numeric_limits<double>::infinity();
                    (which ? fy : fx)[(size_t) j * c.np + i] = f;
                }
        fX_.upload(fx.data(), fx.size(), (size_t) p * c.n * c.np);
        fY_.upload(fy.data(), fy.size(), (size_t) p * c.n * c.np);
    }
}

// host objective: f of a moved yhat, parked in scal.fyhat for ccp_yhat
void CcpsoEngine::host_eval_yhat()
{
    const CcpConst &c = c_;
    BBO_HIP(hipStreamSynchronize(stream_));
    std::vector<CcpScal> sc(c.npop);
    scal_.download(sc.data(), c.npop);
    std::vector<double> yh(c.ld);
    bool touched = false;
    for (int p = 0; p < c.npop; p++) {
        if ((c.honor_stop && sc[p].stop) || !sc[p].yupd) continue;
        yhat_.download(yh.data(), c.ld, (size_t) p * c.ld);
        double f = 0.;
        obj_.eval_host(yh.data(), 1, c.n, c.ld, &f);
        sc[p].fyhat = f != f ? std::numeric_limits<double>::infinity() : f;
        touched = true;
    }
    if (touched) scal_.upload(sc.data(), c.npop);
}

void CcpsoEngine::launch_regroup_eval()
{
    CcpConst &c = c_;
    const int P = c.npop;
    timer_.begin(stream_, K_REGROUP);
    hipLaunchKernelGGL(ccp_regroup, dim3(P), dim3(256), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    // the largest swarm count any subset size can ask for (the device knows the real one)
    int cpmin = c.pps[0];
    for (int k = 1; k < c.npps; k++) cpmin = std::min(cpmin, c.pps[k]);
    const int maxswarm = c.n / cpmin;
    timer_.begin(stream_, K_EVAL);
    if (obj_.on_device()) {
        if (c.ld <= 256)
            hipLaunchKernelGGL(ccp_eval<16>, dim3((CCP_SPLIT * maxswarm + 15) / 16, P), dim3(256),
                    (size_t) 16 * c.ld * sizeof(double), stream_, d_, c_);
        else
            hipLaunchKernelGGL(ccp_eval<64>, dim3((CCP_SPLIT * maxswarm + 3) / 4, P), dim3(256),
                    (size_t) 4 * c.ld * sizeof(double), stream_, d_, c_);
    } else {
        host_eval_candidates();
    }
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

void CcpsoEngine::launch_rest()
{
    CcpConst &c = c_;
    const int P = c.npop;
    int cpmin = c.pps[0];
    for (int k = 1; k < c.npps; k++) cpmin = std::min(cpmin, c.pps[k]);
    timer_.begin(stream_, K_UPDATE);
    hipLaunchKernelGGL(ccp_update, dim3(c.n / cpmin, P), dim3(64), 0, stream_, d_, c_);
    if (!obj_.on_device()) host_eval_yhat();
    hipLaunchKernelGGL(ccp_yhat, dim3(P), dim3(256), (size_t) c.ld * sizeof(double), stream_, d_,
            c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_POSITION);
    hipLaunchKernelGGL(ccp_strategy, dim3(((c.n / cpmin) * c.np + 255) / 256, P), dim3(256), 0,
            stream_, d_, c_);
    const int rparts = (c.ld / 2 + 255) / 256;
    hipLaunchKernelGGL(ccp_position, dim3(rparts, c.np, P), dim3(256), 0, stream_, d_, c_);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
    timer_.begin(stream_, K_FINISH);
    hipLaunchKernelGGL(ccp_finish, dim3(P), dim3(256), 0, stream_, d_, c_, rparts);
    timer_.end(stream_);
    BBO_HIP(hipGetLastError());
}

void CcpsoEngine::generation(bool honor_stop)
{
    c_.honor_stop = honor_stop ? 1 : 0;
    launch_regroup_eval();
    launch_rest();
}

// ---- swarm groups sharded over ranks ---------------------------------------------------
void CcpsoEngine::set_shard(int rank, int world)
{
    BBO_REQUIRE(world >= 1 && rank >= 0 && rank < world, "shard: need 0 <= rank < world");
    BBO_REQUIRE(params_.populations == 1 || world == 1,
            "sharded swarm groups work on one population");
    c_.shard_rank = rank;
    c_.shard_world = world;
    shard_rank_ = rank;
    shard_world_ = world;
}

void CcpsoEngine::phase(int which)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "phase before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    c_.honor_stop = 0;
    if (which == 0) launch_regroup_eval();
    else if (which == 1) launch_rest();
    else throw Error(BBO_ERR_ARG, "unknown CCPSO phase");
    // Phase 0 with the objective on the device returns without waiting: what follows it --
    // bbo_ccpso_export_tables, bbo_get -- is ordered behind it on the engine's stream and
    // synchronises itself (one host wait per exchange instead of two).  Phase 1 ends a generation
    // and waits, like bbo_iterate.
    if (which == 0 && obj_.on_device() && !timer_.on()) return;
    BBO_HIP(hipStreamSynchronize(stream_));
    timer_.collect();
}

// doubles per table in one rank's record: room for the largest block a rank can own
// (ceil(max swarms / world) swarms of np particles; max swarms = n / the smallest swarm size)
int CcpsoEngine::shard_stride() const
{
    int smin = c_.pps[0];
    for (int k = 1; k < c_.npps; k++) smin = std::min(smin, c_.pps[k]);
    const int smax = c_.n / std::max(1, smin);
    return ((smax + c_.shard_world - 1) / c_.shard_world) * c_.np;
}

int CcpsoEngine::table_record() const
{
    return 2 * shard_stride();    // fX block | fY block of this rank's swarms
}

void CcpsoEngine::export_tables(double *dst, bool device_memory)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "export_tables before initialize()");
    BBO_HIP(hipSetDevice(params_.device));
    const int stride = shard_stride();
    double *out = dst;
    if (!device_memory) {
        if (stage_.count != (size_t) 2 * stride) stage_.alloc((size_t) 2 * stride);
        out = stage_.p;
    }
    hipLaunchKernelGGL(ccp_export, dim3((stride + 255) / 256), dim3(256), 0, stream_, d_, c_, out,
            stride);
    BBO_HIP(hipGetLastError());
    if (!device_memory)
        BBO_HIP(hipMemcpyAsync(dst, out, (size_t) 2 * stride * sizeof(double),
                hipMemcpyDeviceToHost, stream_));
    // the caller's collective runs on another stream: the record must be complete when we return
    BBO_HIP(hipStreamSynchronize(stream_));
}

void CcpsoEngine::merge_tables(const double *gathered, int world, bool device_memory)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "merge_tables before initialize()");
    BBO_REQUIRE(world == c_.shard_world, "merge_tables: world differs from the shard setting");
    BBO_HIP(hipSetDevice(params_.device));
    const int stride = shard_stride();
    const double *src = gathered;
    if (!device_memory) {
        const size_t cnt = (size_t) world * 2 * stride;
        BBO_HIP(hipStreamSynchronize(stream_));  // (the upload is not stream-ordered)
        if (gather_.count != cnt) gather_.alloc(cnt);
        gather_.upload(gathered, cnt);
        src = gather_.p;
    }
    const int cap = c_.n * c_.np;
    hipLaunchKernelGGL(ccp_merge, dim3((cap + 255) / 256), dim3(256), 0, stream_, d_, c_, src,
            world, stride);
    BBO_HIP(hipGetLastError());
    // (no host synchronisation: phase(1) is ordered behind the merge on the engine's stream; a
    // device-memory `gathered` must stay untouched until the next export, which synchronises)
    if (!device_memory) BBO_HIP(hipStreamSynchronize(stream_));
}

// Once the swarm groups are sharded, a generation is phase(0) / merge / phase(1): the plain entry
// points would evaluate this rank's block only and update from stale rows for every other swarm.
void CcpsoEngine::require_unsharded(const char *what) const
{
    if (c_.shard_world > 1)
        throw Error(BBO_ERR_STATE, std::string(what) + ": the swarm groups are sharded over " +
                std::to_string(c_.shard_world) + " ranks -- drive generations with "
                "bbo_ccpso_phase(0) / bbo_ccpso_merge_tables / bbo_ccpso_phase(1)");
}

// ---- the local optimizer (CCPSOSearch::localSearch, ccpso.cpp:371-435; adaptive weighting) -----
// Every `localfreq` generations one weight per swarm, scaling that swarm's coordinates of the
// context vector, is optimized by `local` inside the box that keeps the scaled vector in bounds;
// the result replaces yhat if it is better.  The generations run on the device, the search is
// driven from here: its objective is a host callback that evaluates f(yhat * w[group]) (the
// caller's callable, or the built-in's formula on the host).  Each search starts `local` afresh
// (B = C = I) under the seed base + search index -- the reference's CMA-ES objects start from the
// previous search's matrices (cmaes.cpp:53-54), which corrupts the heap when the swarm count
// grows; not reproduced (HISTORY.md section 3, CCPSO2).
void CcpsoEngine::set_local(Optimizer *local, int localfreq)
{
    BBO_REQUIRE(c_.npop <= 1 && params_.populations <= 1,
            "CCPSO: the local optimizer works on one population");
    local_ = local;
    localfreq_ = localfreq;
    nlocal_ = 0;
    if (auto *cma = dynamic_cast<CmaEngine*>(local)) local_seed0_ = cma->seed();
}

double CcpsoEngine::eval_full(const double *x)
{
    if (!obj_.on_device()) {
        double f = 0.;
        obj_.eval_host(x, 1, c_.n, c_.n, &f);
        return f != f ? std::numeric_limits<double>::infinity() : f;
    }
    return builtin_objective_host(obj_.builtin, c_.n, x, aux_h_.data());
}

namespace {
struct LocalCtx {
    CcpsoEngine *eng;
    const double *yhat;
    const int *group;
    int n;
    std::vector<double> x;
    std::string error;
};

double local_objective(const double *w, int nsw, void *user, int *failed)
{
    (void) nsw;
    auto *cx = static_cast<LocalCtx*>(user);
    for (int j = 0; j < cx->n; j++) cx->x[j] = cx->yhat[j] * w[cx->group[j]];
    try {
        return cx->eng->eval_full(cx->x.data());
    } catch (const std::exception &e) {
        cx->error = e.what();
        *failed = 1;
        return 0.;
    }
}
}

void CcpsoEngine::local_search()
{
    const int n = c_.n;
    BBO_HIP(hipStreamSynchronize(stream_));
    CcpScal s;
    scal_.download(&s, 1, 0);
    std::vector<double> yh(c_.ld);
    yhat_.download(yh.data(), c_.ld, 0);
    std::vector<int> k(n);                       // position -> coordinate, swarm-major
    range_.download(k.data(), n, 0);
    const int cps = s.cpswarm, nsw = s.nswarm;
    std::vector<int> group(n);
    for (int pos = 0; pos < n; pos++) group[k[pos]] = pos / cps;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> wlb(nsw, -inf), wub(nsw, inf), wguess(nsw);
    for (int j = 0; j < n; j++) {
        const double y = yh[j];
        const double scale = std::fabs(y) < 1e-3 ? (y > 0. ? 1e-3 : -1e-3) : y;
        double lb = lower_h_[j] / scale, ub = upper_h_[j] / scale;
        if (lb > ub) std::swap(lb, ub);
        wlb[group[j]] = std::max(wlb[group[j]], lb);
        wub[group[j]] = std::min(wub[group[j]], ub);
    }
    for (int g = 0; g < nsw; g++) wguess[g] = std::max(wlb[g], std::min(1., wub[g]));

    LocalCtx cx { this, yh.data(), group.data(), n, std::vector<double>(n), std::string() };
    ObjectiveSpec faux;
    faux.kind = BBO_OBJECTIVE_SCALAR_CALLBACK;
    faux.scalar = local_objective;
    faux.user = &cx;
    if (auto *cma = dynamic_cast<CmaEngine*>(local_)) cma->fresh_start(local_seed0_ + (uint64_t) nlocal_);
    nlocal_++;
    std::vector<double> w(nsw);
    int lfev = 0, lconv = 0;
    local_->optimize(nsw, wlb.data(), wub.data(), wguess.data(), faux, w.data(), &lfev, &lconv);
    BBO_HIP(hipSetDevice(params_.device));       // (the local optimizer may live on another device)
    int fev = s.fev + lfev;
    std::vector<double> trial(c_.ld, 0.);
    bool inside = true;
    for (int j = 0; j < n; j++) {
        trial[j] = yh[j] * w[group[j]];
        if (c_.correct && !(trial[j] >= lower_h_[j] && trial[j] <= upper_h_[j])) inside = false;
    }
    scal_.download(&s, 1, 0);
    if (inside) {
        const double fwy = eval_full(trial.data());
        fev++;
        if (fwy < s.fyhat) {
            yhat_.upload(trial.data(), c_.ld, 0);
            s.fyhat = fwy;
            s.improved = 1;
        }
    }
    s.fev = fev;
    // (the reference tests the budget right after the search, ccpso.cpp:137-141: the next
    // generation must not start once the search has spent it)
    if (fev >= c_.mfev && !s.stop) s.stop = 2;
    scal_.upload(&s, 1, 0);
}

void CcpsoEngine::after_generation(int gen_before)
{
    if (local_ && localfreq_ > 0 && gen_before % localfreq_ == 0) {   // ccpso.cpp:116-118
        CcpScal s;
        BBO_HIP(hipStreamSynchronize(stream_));
        scal_.download(&s, 1, 0);
        // the reference's iterate() searches BEFORE optimize() tests the budget and the spread
        // (ccpso.cpp:112-147): a generation that has just raised the stop flag still gets its
        // search; one that did not run at all (the flag was up before it) does not
        if (s.gen != gen_before) local_search();
    }
}

void CcpsoEngine::iterate()
{
    if (!inited_) throw Error(BBO_ERR_STATE, "iterate() before initialize()");
    require_unsharded("iterate()");
    BBO_HIP(hipSetDevice(params_.device));
    int gen0 = 0;
    if (local_) {
        CcpScal s;
        BBO_HIP(hipStreamSynchronize(stream_));
        scal_.download(&s, 1, 0);
        gen0 = s.gen;
    }
    generation(false);
    BBO_HIP(hipStreamSynchronize(stream_));
    if (local_ && localfreq_ > 0 && gen0 % localfreq_ == 0) local_search();
    timer_.collect();
}

bool CcpsoEngine::all_stopped()
{
    std::vector<CcpScal> sc(c_.npop);
    scal_.download(sc.data(), c_.npop);
    for (const auto &s : sc)
        if (!s.stop) return false;
    return true;
}

int CcpsoEngine::run(int max_generations)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "run() before initialize()");
    require_unsharded("run()");
    BBO_HIP(hipSetDevice(params_.device));
    // (the reference's loop is `while (true) { iterate(); ... }`: at least one generation)
    const int poll = params_.poll_every > 0 ? params_.poll_every : 8;
    int done = 0;
    while (done < max_generations) {
        if (all_stopped()) break;
        const int chunk = (obj_.on_device() && !local_) ? std::min(poll, max_generations - done) : 1;
        int gen0 = 0;
        if (local_) {
            CcpScal s;
            scal_.download(&s, 1, 0);
            gen0 = s.gen;
        }
        for (int g = 0; g < chunk; g++) generation(true);
        BBO_HIP(hipStreamSynchronize(stream_));
        if (local_) after_generation(gen0);
        timer_.collect();
        done += chunk;
    }
    return done;
}

void CcpsoEngine::solution(int population, double *x_out, int *n_evals, int *converged)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "solution() before initialize()");
    BBO_REQUIRE(population >= 0 && population < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    CcpScal s;
    scal_.download(&s, 1, population);
    std::vector<double> x(c_.ld);
    yhat_.download(x.data(), c_.ld, (size_t) population * c_.ld);
    std::copy(x.begin(), x.begin() + c_.n, x_out);
    *n_evals = s.fev;
    if (s.gen == 0) {
        std::vector<double> rad(c_.np);
        radius_.download(rad.data(), c_.np, (size_t) population * c_.np);
        double mean = 0.;
        for (double r : rad) mean += r;
        mean /= c_.np;
        double m2 = 0.;
        for (double r : rad) m2 += (r - mean) * (r - mean);
        *converged = m2 <= (c_.np - 1) * c_.stol * c_.stol ? 1 : 0;
    } else {
        *converged = s.conv;
    }
}

void CcpsoEngine::optimize(int n, const double *lower, const double *upper, const double *guess,
        const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged)
{
    require_unsharded("optimize()");
    init(n, lower, upper, guess, obj);
    run(std::numeric_limits<int>::max());
    int conv = 0;
    solution(0, x_out, n_evals, &conv);
    CcpScal s;
    scal_.download(&s, 1, 0);
    *converged = s.stop == 1 ? 1 : 0;
}

int CcpsoEngine::get(const std::string &k, int p, double *out, int cap)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "get() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));
    const CcpConst &c = c_;
    CcpScal s;
    scal_.download(&s, 1, p);
    auto one = [&](double v) {
        if (out && cap >= 1) out[0] = v;
        return 1;
    };
    if (k == "profile") return timer_.report(out, cap);
    if (k == "x" || k == "y") {
        const int cnt = c.np * c.n;
        if (out && cap >= cnt) {
            std::vector<double> M((size_t) c.np * c.ld);
            (k == "x" ? X_ : Y_).download(M.data(), M.size(), (size_t) p * c.np * c.ld);
            for (int i = 0; i < c.np; i++)
                std::copy(M.begin() + (size_t) i * c.ld, M.begin() + (size_t) i * c.ld + c.n,
                        out + (size_t) i * c.n);
        }
        return cnt;
    }
    if (k == "yhat") {
        if (out && cap >= c.n) {
            std::vector<double> v(c.ld);
            yhat_.download(v.data(), c.ld, (size_t) p * c.ld);
            std::copy(v.begin(), v.begin() + c.n, out);
        }
        return c.n;
    }
    if (k == "fx" || k == "fy") {
        const int cnt = s.nswarm * c.np;
        if (out && cap >= cnt && cnt > 0)
            (k == "fx" ? fX_ : fY_).download(out, cnt, (size_t) p * c.n * c.np);
        return cnt;
    }
    if (k == "ibest" || k == "strat" || k == "k") {
        const int cnt = k == "k" ? c.n : s.nswarm * c.np;
        if (out && cap >= cnt && cnt > 0) {
            std::vector<int> v(cnt);
            if (k == "k") range_.download(v.data(), cnt, (size_t) p * c.n);
            else (k == "ibest" ? ibest_ : strat_).download(v.data(), cnt, (size_t) p * c.n * c.np);
            for (int q = 0; q < cnt; q++) out[q] = v[q];
        }
        return cnt;
    }
    if (k == "fyhat") return one(s.fyhat);
    if (k == "phat") return one(s.phat);
    if (k == "fev") return one(s.fev);
    if (k == "it") return one(s.gen);
    if (k == "is") return one(s.is);
    if (k == "nswarm") return one(s.nswarm);
    if (k == "cpswarm") return one(s.cpswarm);
    if (k == "improved") return one(s.improved);
    if (k == "np") return one(c.np);
    if (k == "stop") return one(s.stop);
    if (k == "conv") return one(s.conv);
    if (k == "m2") return one(s.m2);
    if (k == "n") return one(c.n);
    throw Error(BBO_ERR_KEY, "unknown state key '" + k + "'");
}

int CcpsoEngine::set(const std::string &k, int p, const double *in, int count)
{
    if (!inited_) throw Error(BBO_ERR_STATE, "set() before initialize()");
    BBO_REQUIRE(p >= 0 && p < c_.npop, "population index out of range");
    if (k == "profile") {
        timer_.enable(in[0] != 0., K_COUNT, K_NAMES);
        return 1;
    }
    // what a host-side local search (ccpso.cpp:371-435; the Python class drives it) hands back:
    // the re-weighted context vector, its value, the evaluations it spent, the `improved` flag
    BBO_HIP(hipSetDevice(params_.device));
    BBO_HIP(hipStreamSynchronize(stream_));      // (uploads below are not stream-ordered)
    if (k == "yhat") {
        BBO_REQUIRE(count == c_.n, "set yhat: wrong element count");
        std::vector<double> v(c_.ld, 0.);
        std::copy(in, in + c_.n, v.begin());
        yhat_.upload(v.data(), c_.ld, (size_t) p * c_.ld);
        return 1;
    }
    if (k == "fyhat" || k == "fev" || k == "improved") {
        BBO_REQUIRE(count == 1, "set: wrong element count");
        CcpScal s;
        scal_.download(&s, 1, p);
        if (k == "fyhat") s.fyhat = in[0];
        else if (k == "fev") s.fev = (int) in[0];
        else s.improved = in[0] != 0. ? 1 : 0;
        scal_.upload(&s, 1, p);
        return 1;
    }
    throw Error(BBO_ERR_KEY, "unknown or read-only state key '" + k + "'");
}

Optimizer* make_ccpso_engine(const bbo_params &p)
{
    return new CcpsoEngine(p);
}

} // namespace bbo
