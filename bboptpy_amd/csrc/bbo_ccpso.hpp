// bbo_ccpso.hpp -- device-resident CCPSO2 (cooperatively coevolving particle swarms).
//
// Reference: CCPSOSearch (src/multivariate/pso/ccpso.cpp:51-454; Li & Yao 2012), without the
// optional local optimizer (`local = nullptr`, the binding's default).  The n coordinates are
// regrouped at random every generation into n/s swarms of s coordinates; every particle of
// every swarm is scored by plugging its s coordinates into the context vector yhat, which is
// 2 (n/s) np full objective evaluations per generation -- the throughput driver, and the part
// that is embarrassingly parallel (yhat does not move while they run).
#pragma once

#include "bbo_common.hpp"

namespace bbo {

struct CcpScal {
    double fyhat, fyhat0, phat, m2;
    int fev, gen, is, improved, nswarm, cpswarm, stop, conv;
    int yupd;                // a swarm moved yhat this generation (ccpso.cpp:262-268)
    int pad_;
};

struct CcpConst {
    int n, ld, np, npps, correct, adaptp, obj, mfev, honor_stop, npop;
    int pps[16];
    double stol, phat0;
    uint64_t seed;
    // swarm groups sharded over ranks: this handle evaluates the swarms
    // [nswarm * shard_rank / shard_world, nswarm * (shard_rank + 1) / shard_world)
    int shard_rank, shard_world;
};

struct CcpDev {
    double *X, *Y;           // [P][np][ld] positions, personal bests
    double *yhat, *ysave;    // [P][ld] context vector and its copy at generation start
    double *fX, *fY;         // [P][n * np] swarm-major: [j][i]
    int *ibest, *strat;      // [P][n * np]
    int *range, *grp_of;     // [P][n]: position -> coordinate, coordinate -> swarm
    double *radius;          // [P][np]
    double *rpart;           // [P][np][ceil(ld / 512)] partial sums of squares (ccp_position)
    const double *lower, *upper, *aux;
    CcpScal *scal;
};

class CcpsoEngine: public Optimizer {
public:
    explicit CcpsoEngine(const bbo_params &p);
    ~CcpsoEngine() override;
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

    // ---- swarm groups sharded over GPUs (one population; bbo_ccpso_* of the C ABI) ----------
    void set_shard(int rank, int world);
    void phase(int which);                 // 0: regroup + evaluate this rank's swarms, 1: the rest
    int table_record() const;              // doubles in one rank's record: fX | fY blocks of its swarms
    void export_tables(double *dst, bool device_memory);
    void merge_tables(const double *gathered, int world, bool device_memory);

    // ---- the optional local optimizer (ccpso.cpp:116-118, 371-435; bbo_ccpso_set_local) ------
    void set_local(Optimizer *local, int localfreq);
    double eval_full(const double *x);     // the objective at one n-vector, on the host

private:
    void launch_regroup_eval();
    void launch_rest();
    void generation(bool honor_stop);
    void host_eval_candidates();
    void host_eval_yhat();
    bool all_stopped();
    int shard_stride() const;
    void require_unsharded(const char *what) const;

    bbo_params params_;
    ObjectiveSpec obj_;
    CcpConst c_ {};
    CcpDev d_ {};
    hipStream_t stream_ = nullptr;
    bool inited_ = false;
    int shard_rank_ = 0, shard_world_ = 1;     // survive init() (c_ is rebuilt there)
    void local_search();
    void after_generation(int gen_before);
    Optimizer *local_ = nullptr;               // borrowed, like the reference's `local` pointer
    int localfreq_ = 10, nlocal_ = 0;
    uint64_t local_seed0_ = 0;
    std::vector<double> lower_h_, upper_h_;
    std::vector<double> aux_h_;
    DevBuf<double> X_, Y_, yhat_, ysave_, fX_, fY_, radius_, rpart_, lower_, upper_, aux_, gather_, stage_;
    DevBuf<int> ibest_, strat_, range_, grp_of_;
    DevBuf<CcpScal> scal_;
    KernelTimer timer_;
};

Optimizer* make_ccpso_engine(const bbo_params &p);

} // namespace bbo
