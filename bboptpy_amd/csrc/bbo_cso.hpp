// bbo_cso.hpp -- device-resident competitive swarm optimizer.
//
// Reference: CSOSearch (src/multivariate/pso/cso.cpp:46-276; Cheng & Jin 2015 with the
// multi-loser / ring variants).  A generation = neighbourhood or swarm mean, a random shuffle,
// a sort inside every group of `pcompete` slots, the winners' mean, and the learning step of
// every loser (from the next better particle of its group).  Unlike APSO there is no O(np^2)
// term: everything is a streaming pass, priced in HBM bytes.  Particles never move in memory:
// row r is the particle born in slot r (which is also what the reference's stored neighbour
// pointers keep referring to), the swarm ORDER is the permutation occ[slot] = row.
#pragma once

#include "bbo_common.hpp"

namespace bbo {

struct CsoScal {
    double fbest, m2;
    int fev, gen, stop, conv;
    int ibest;               // ROW of the incumbent
    int pad_;
};

struct CsoConst {
    int n, ld, np, pc, ngroup, ring, correct, obj, mfev, honor_stop, npop, parts, fparts;
    int nwg;                  // workgroups of the fused cso_compete per population (rows of wgpart)
    double stol, vmax, phil, phih;
    uint64_t seed;
};

struct CsoDev {
    double *X, *V, *PM;      // [P][np][ld] positions, velocities, ring means (PM only if ring)
    double *f, *radius;      // [P][np] by row
    int *occ, *occ2;         // [P][np] slot -> row: current order, and scratch for the shuffle
    double *mean, *meanw;    // [P][ld] swarm mean, winners' mean
    double *colpart;         // [P][parts][ld]
    double *wgpart;          // [P][nwg][ld] column sums per cso_compete workgroup (fused swarm mean)
    double *fpart;           // [P][fparts][5] slab results of cso_finish_part
    const double *lower, *upper, *aux;
    CsoScal *scal;
};

class CsoEngine: public Optimizer {
public:
    explicit CsoEngine(const bbo_params &p);
    ~CsoEngine() override;
    void init(int n, const double *lower, const double *upper, const double *guess,
            const ObjectiveSpec &obj) override;
    void iterate() override;
    void solution(int population, double *x_out, int *n_evals, int *converged) override;
    void optimize(int n, const double *lower, const double *upper, const double *guess,
            const ObjectiveSpec &obj, double *x_out, int *n_evals, int *converged) override;
    int run(int max_generations) override;
    int get(const std::string &key, int population, double *out, int cap) override;
    int set(const std::string &key, int population, const double *in, int count) override;
    int dimension() const override { return c_.n; }

private:
    void generation(bool honor_stop);
    void host_evaluate(bool losers_only);
    bool all_stopped();

    bbo_params params_;
    ObjectiveSpec obj_;
    CsoConst c_ {};
    CsoDev d_ {};
    hipStream_t stream_ = nullptr;
    bool inited_ = false;
    std::vector<double> aux_h_;
    DevBuf<double> X_, V_, PM_, f_, radius_, mean_, meanw_, colpart_, wgpart_, fpart_, lower_, upper_, aux_;
    int fuse_g_ = 0;            // > 0: lanes per group of cso_compete, which then maintains the swarm mean's sums
    DevBuf<int> occ_, occ2_;
    DevBuf<CsoScal> scal_;
    KernelTimer timer_;
};

Optimizer* make_cso_engine(const bbo_params &p);

} // namespace bbo
